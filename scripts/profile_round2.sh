# Round-2 profile set (run on the GPU box through gpurun): rocprofv3 kernel traces of the bench command for the headline
# workload and the new paths, plus the separate PMC passes (HBM traffic, SQ counters) for the headline and the large batch.
# Counters are collected in their own runs, never together with --sys-trace / hip / hsa trace domains.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NB="--no-cpu-baseline"
# ---- headline: MFM_KL_EF, MOSI canonical, B=32, fp32
timeout 300 rocprofv3 --kernel-trace --stats -d $O/h32 -o ktrace -- python $R/bench.py --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_B32.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/h32 -o pmc_fetch -- python $R/bench.py --steps 50 --warmup 10 $NB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/h32 -o pmc_write -- python $R/bench.py --steps 50 --warmup 10 $NB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $O/h32 -o pmc_sq -- python $R/bench.py --steps 50 --warmup 10 $NB > /dev/null 2>&1
# ---- bf16 at B=32 and the large batch in both precisions
timeout 300 rocprofv3 --kernel-trace --stats -d $O/b32 -o ktrace -- python $R/bench.py --dtype bf16 --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_B32_bf16.json 2>/dev/null
for dt in fp32 bf16; do
  A="--dtype $dt --batch 2048 --steps 30 --warmup 5 $NB"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/l_$dt -o ktrace -- python $R/bench.py $A > $O/bench_under_rocprof_B2048_$dt.json 2>/dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/l_$dt -o pmc_fetch -- python $R/bench.py $A > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/l_$dt -o pmc_write -- python $R/bench.py $A > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $O/l_$dt -o pmc_sq -- python $R/bench.py $A > /dev/null 2>&1
done
# ---- MFM_KL / MFM on the fused plan
for m in kl mmd; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/m_$m -o ktrace -- python $R/bench.py --model $m --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_B32_$m.json 2>/dev/null
done
cd $R
for d in h32 b32 l_fp32 l_bf16 m_kl m_mmd; do
  f=$(ls $O/$d/ktrace*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/kernel_stats_$d.txt
done
python scripts/make_traffic_json.py $(ls $O/h32/pmc_fetch*.db | head -1) $(ls $O/h32/pmc_write*.db | head -1) > $O/traffic_B32.json
for d in h32 l_fp32 l_bf16; do python scripts/roofline_table.py $O/$d > $O/roofline_table_$d.txt 2>&1; done
# un-profiled reference runs of the same commands
python bench.py > $O/bench_B32.json 2> $O/bench_B32.err
MFM_BENCH_ALL_CORES=1 python bench.py --steps 100 --warmup 20 > $O/bench_B32_allcores.json 2>/dev/null
python bench.py --dtype bf16 $NB > $O/bench_B32_bf16.json 2>/dev/null
for m in kl mmd; do python bench.py --model $m > $O/bench_B32_$m.json 2>/dev/null; done
rm -rf $O/*/*.db.tmp
ls -la $O
du -sh $O
