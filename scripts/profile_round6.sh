# Round-6 profile set (run on the GPU box through gpurun): rocprofv3 kernel trace of the bench command (MFM_KL_EF B=32 fp32,
# MFM_KL / MFM, bf16 B=2048), the separate PMC passes (HBM traffic, SQ counters; never together with --sys-trace / hip / hsa
# trace domains) incl. the bf16 lines' traffic files, un-profiled bench lines, the launch clock, the unchanged-loop timings,
# the recurrence table, P2P on one device.
# usage: bash scripts/profile_round6.sh [out-subdir]      (scripts/launch_timeline.sh must have been run: its .so travels)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_r06}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NB="--no-cpu-baseline"
PMC_SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
prof() {   # dir, bench args...
  d=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$d -o ktrace -- python $R/bench.py "$@" $NB > $O/bench_under_rocprof_$d.json 2>/dev/null
}
pmc() {    # dir, bench args...
  d=$1; shift
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/$d -o pmc_fetch -- python $R/bench.py "$@" $NB > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/$d -o pmc_write -- python $R/bench.py "$@" $NB > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $PMC_SQ -d $O/$d -o pmc_sq -- python $R/bench.py "$@" $NB > /dev/null 2>&1
}
prof h32 --steps 200 --warmup 10
pmc h32 --steps 30 --warmup 5
pmc h32_bf16 --dtype bf16 --steps 30 --warmup 5
for m in kl mmd; do prof m_$m --model $m --steps 200 --warmup 20; done
L="--dtype bf16 --batch 2048 --steps 30 --warmup 5"
prof l_bf16 $L
pmc l_bf16 $L
cd $R
for d in h32 m_kl m_mmd l_bf16; do
  f=$(ls $O/$d/ktrace*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/kernel_stats_$d.txt
done
python scripts/make_traffic_json.py $(ls $O/h32/pmc_fetch*.db | head -1) $(ls $O/h32/pmc_write*.db | head -1) "mosi B=32 T=20 fp32" > $O/traffic_B32.json
python scripts/make_traffic_json.py $(ls $O/h32_bf16/pmc_fetch*.db | head -1) $(ls $O/h32_bf16/pmc_write*.db | head -1) "mosi B=32 T=20 bf16 plan" > $O/traffic_B32_bf16.json
python scripts/make_traffic_json.py $(ls $O/l_bf16/pmc_fetch*.db | head -1) $(ls $O/l_bf16/pmc_write*.db | head -1) "mosi B=2048 T=20 bf16-resident plan" > $O/traffic_B2048_bf16.json
python scripts/roofline_table.py $O/h32 > $O/roofline_table_h32.txt 2>&1
python scripts/roofline_table.py $O/l_bf16 > $O/roofline_table_l_bf16.txt 2>&1
python bench.py > $O/bench_B32.json 2> $O/bench_B32.err
python bench.py --steps 400 --warmup 40 $NB > $O/bench_B32_400.json 2>/dev/null
python bench.py --dtype bf16 --steps 400 --warmup 40 $NB > $O/bench_B32_bf16.json 2>/dev/null
for m in kl mmd; do python bench.py --model $m --steps 400 --warmup 40 $NB > $O/bench_B32_$m.json 2>/dev/null; done
python bench.py --steps 400 --warmup 40 --breakdown $NB 2> $O/breakdown_B32.txt > /dev/null
python bench.py --shape you --seq 50 --steps 200 --warmup 20 $NB > $O/bench_you_B32_T50.json 2>/dev/null
python bench.py --shape mosei --seq 50 --steps 200 --warmup 20 $NB > $O/bench_mosei_B32_T50.json 2>/dev/null
python bench.py --shape mosei --seq 50 --batch 1024 --dtype bf16 --steps 30 --warmup 5 $NB > $O/bench_mosei_B1024_T50_bf16.json 2>/dev/null
python bench.py --shape you --seq 50 --batch 2048 --dtype bf16 --steps 20 --warmup 5 $NB > $O/bench_you_B2048_T50_bf16.json 2>/dev/null
python bench.py --dtype bf16 --batch 2048 --steps 50 --warmup 10 $NB > $O/bench_B2048_bf16.json 2>/dev/null
for B in 8 16 32 48 64 192 512 2048; do python bench.py --batch $B --dtype $([ $B -ge 192 ] && echo bf16 || echo fp32) --steps $([ $B -ge 192 ] && echo 40 || echo 200) --warmup 10 $NB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d %s  %.4f ms  %.0f samples/s' % ($B, d['dtype'], d['ms_per_step'], d['value']))"; done > $O/batch_sweep.txt
python scripts/bench_dropin.py > $O/graphed_steps.txt 2>&1
MFM_DROPIN_SECTIONS=1 python scripts/bench_dropin.py 2>&1 | tail -1 >> $O/graphed_steps.txt
python scripts/bench_seq_group.py > $O/seq_group.txt 2>&1
python scripts/launch_timeline.py --samples 9 > $O/launch_timeline.txt 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 scripts/bench_p2p.py --one-device 2>/dev/null | tail -1 > $O/p2p_one_device.txt
MFM_BENCH_ONE_DEVICE=1 MFM_P2P_TIMEOUT_MS=20000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 200 --warmup 20 $NB 2>/dev/null | grep "^{" > $O/bench_dp2_one_device.json
rm -rf $O/*/*.db $O/*/*.db.tmp $O/*/*.csv
ls -la $O; head -14 $O/kernel_stats_h32.txt; head -30 $O/roofline_table_h32.txt; cut -c1-260 $O/bench_B32_400.json; cat $O/batch_sweep.txt; cat $O/graphed_steps.txt | tail -14
