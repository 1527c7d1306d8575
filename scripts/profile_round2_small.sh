# Round-2 profile refresh for the small-batch paths only (headline MFM_KL_EF B=32 and the MFN plans); the large-batch and
# bf16 parts of scripts/profile_round2.sh are unaffected by the kernels that changed last (gemm_tn, lin_rows, mmd).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
NB="--no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/h32 -o ktrace -- python $R/bench.py --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_B32.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/h32 -o pmc_fetch -- python $R/bench.py --steps 50 --warmup 10 $NB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/h32 -o pmc_write -- python $R/bench.py --steps 50 --warmup 10 $NB > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $O/h32 -o pmc_sq -- python $R/bench.py --steps 50 --warmup 10 $NB > /dev/null 2>&1
for m in kl mmd; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/m_$m -o ktrace -- python $R/bench.py --model $m --steps 200 --warmup 20 $NB > $O/bench_under_rocprof_B32_$m.json 2>/dev/null
done
cd $R
for d in h32 m_kl m_mmd; do
  f=$(ls $O/$d/ktrace*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/rocprof_summary.py $f > $O/kernel_stats_$d.txt
done
python scripts/make_traffic_json.py $(ls $O/h32/pmc_fetch*.db | head -1) $(ls $O/h32/pmc_write*.db | head -1) > $O/traffic_B32.json
python scripts/roofline_table.py $O/h32 > $O/roofline_table_h32.txt 2>&1
cp $O/traffic_B32.json $R/profiles/r02_traffic.json       # bench.py reads the committed file for roofline.traffic
python bench.py > $O/bench_B32.json 2> $O/bench_B32.err
MFM_BENCH_ALL_CORES=1 python bench.py --steps 100 --warmup 20 > $O/bench_B32_allcores.json 2>/dev/null
for m in kl mmd; do python bench.py --model $m > $O/bench_B32_$m.json 2>/dev/null; done
python bench.py --steps 400 --warmup 40 --breakdown $NB 2> $O/breakdown_B32.txt > /dev/null
rm -rf $O/*/*.db $O/*/*.db.tmp
ls -la $O
