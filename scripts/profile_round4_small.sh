# Round-4 profile set for the headline batch size after the role workgroups (run on the GPU box through gpurun): rocprofv3
# kernel trace of the bench command (MFM_KL_EF, B=32, fp32 and bf16), the separate PMC passes (HBM traffic, SQ counters; never
# together with --sys-trace / hip / hsa trace domains), un-profiled bench lines with the role workgroups on and off.
# usage: bash scripts/profile_round4_small.sh [out-subdir]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_r04s}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NB="--no-cpu-baseline"
PMC_SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
d=h32
timeout 300 rocprofv3 --kernel-trace --stats -d $O/$d -o ktrace -- python $R/bench.py --steps 200 --warmup 10 $NB > $O/bench_under_rocprof_$d.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/$d -o pmc_fetch -- python $R/bench.py --steps 30 --warmup 5 $NB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/$d -o pmc_write -- python $R/bench.py --steps 30 --warmup 5 $NB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc $PMC_SQ -d $O/$d -o pmc_sq -- python $R/bench.py --steps 30 --warmup 5 $NB > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/b32 -o ktrace -- python $R/bench.py --dtype bf16 --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_b32.json 2>/dev/null
for m in kl mmd; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/m_$m -o ktrace -- python $R/bench.py --model $m --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_m_$m.json 2>/dev/null
done
cd $R
for m in kl mmd; do python bench.py --model $m --steps 400 --warmup 40 $NB > $O/bench_B32_$m.json 2>/dev/null; done
for d in h32 b32 m_kl m_mmd; do
  f=$(ls $O/$d/ktrace*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/kernel_stats_$d.txt
done
python scripts/make_traffic_json.py $(ls $O/h32/pmc_fetch*.db | head -1) $(ls $O/h32/pmc_write*.db | head -1) > $O/traffic_B32.json
python scripts/roofline_table.py $O/h32 > $O/roofline_table_h32.txt 2>&1
python bench.py > $O/bench_B32.json 2> $O/bench_B32.err
python bench.py --steps 400 --warmup 40 $NB > $O/bench_B32_400.json 2>/dev/null
python bench.py --dtype bf16 --steps 400 --warmup 40 $NB > $O/bench_B32_bf16.json 2>/dev/null
MFM_PROJ_FOLD=0 MFM_DW_FOLD=0 python bench.py --steps 400 --warmup 40 $NB > $O/bench_B32_roles_off.json 2>/dev/null
MFM_PROJ_FOLD=0 MFM_DW_FOLD=0 python bench.py --dtype bf16 --steps 400 --warmup 40 $NB > $O/bench_B32_bf16_roles_off.json 2>/dev/null
python bench.py --steps 400 --warmup 40 --breakdown $NB 2> $O/breakdown_B32.txt > /dev/null
for B in 8 16 24 32 33 48 64; do python bench.py --batch $B --steps 200 --warmup 20 $NB 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=%d  %.4f ms  %.0f samples/s' % ($B, d['ms_per_step'], d['value']))"; done > $O/batch_sweep_small.txt
MFM_DW_FOLD_ATOMICS=1 python bench.py --steps 400 --warmup 40 $NB > $O/bench_B32_dw_atomics.json 2>/dev/null
# two ranks on this one device (what the exchange adds to a step when its peers are local: exposed_collective_us)
MFM_BENCH_ONE_DEVICE=1 MFM_P2P_TIMEOUT_MS=20000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 200 --warmup 20 $NB > $O/bench_dp2_one_device.json 2>/dev/null
rm -rf $O/*/*.db $O/*/*.db.tmp $O/*/*.csv
ls -la $O; cat $O/kernel_stats_h32.txt | head -14; cat $O/roofline_table_h32.txt | head -30; cat $O/bench_B32_400.json $O/bench_B32_roles_off.json $O/bench_B32_dw_atomics.json | cut -c1-260; cat $O/bench_dp2_one_device.json | cut -c1-900; cat $O/traffic_B32.json | head -30; cat $O/batch_sweep_small.txt
