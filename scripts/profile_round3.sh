# Round-3 profile set (run on the GPU box through gpurun): rocprofv3 kernel traces of the bench command for the headline
# workload (MFM_KL_EF, B=32, fp32), the bf16-resident large batch (B=2048) and its fp32 counterpart, and the MFN plans, plus
# the separate PMC passes (HBM traffic, SQ counters).  Counters are collected in their own runs, never together with
# --sys-trace / hip / hsa trace domains.   usage: bash scripts/profile_round3.sh [out-subdir]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NB="--no-cpu-baseline"
PMC_SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
prof() {   # prof <dir> <steps> <bench args...>: kernel trace, then FETCH / WRITE / SQ counter passes
  d=$1; st=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$d -o ktrace -- python $R/bench.py --steps $st --warmup 10 $NB "$@" > $O/bench_under_rocprof_$d.json 2>/dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/$d -o pmc_fetch -- python $R/bench.py --steps 30 --warmup 5 $NB "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/$d -o pmc_write -- python $R/bench.py --steps 30 --warmup 5 $NB "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $PMC_SQ -d $O/$d -o pmc_sq -- python $R/bench.py --steps 30 --warmup 5 $NB "$@" > /dev/null 2>&1
}
prof h32 200
prof l_bf16 30 --dtype bf16 --batch 2048
prof l_fp32 30 --batch 2048
for m in kl mmd; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/m_$m -o ktrace -- python $R/bench.py --model $m --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_m_$m.json 2>/dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/b32 -o ktrace -- python $R/bench.py --dtype bf16 --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_b32.json 2>/dev/null
cd $R
for d in h32 b32 l_fp32 l_bf16 m_kl m_mmd; do
  f=$(ls $O/$d/ktrace*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/kernel_stats_$d.txt
done
python scripts/make_traffic_json.py $(ls $O/h32/pmc_fetch*.db | head -1) $(ls $O/h32/pmc_write*.db | head -1) > $O/traffic_B32.json
for d in h32 l_fp32 l_bf16; do python scripts/roofline_table.py $O/$d > $O/roofline_table_$d.txt 2>&1; done
# un-profiled runs of the same commands
python bench.py > $O/bench_B32.json 2> $O/bench_B32.err
python bench.py --dtype bf16 $NB > $O/bench_B32_bf16.json 2>/dev/null
for B in 1024 2048 4096; do python bench.py --dtype bf16 --batch $B --steps 50 --warmup 10 $NB > $O/bench_B${B}_bf16.json 2>/dev/null; done
python bench.py --batch 2048 --steps 30 --warmup 5 $NB > $O/bench_B2048_fp32.json 2>/dev/null
for m in kl mmd; do python bench.py --model $m $NB > $O/bench_B32_$m.json 2>/dev/null; done
python bench.py --steps 400 --warmup 40 --breakdown $NB 2> $O/breakdown_B32.txt > /dev/null
python bench.py --dtype bf16 --batch 2048 --steps 50 --warmup 10 --breakdown $NB 2> $O/breakdown_B2048_bf16.txt > /dev/null
rm -rf $O/*/*.db $O/*/*.db.tmp $O/*/*.csv
ls -la $O; du -sh $O
cat $O/roofline_table_l_bf16.txt $O/bench_B32.json
