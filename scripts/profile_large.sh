set -x
B=${1:-2048}
mkdir -p gpurun_out/prof_large
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--batch $B --steps 30 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_large -o ktrace -- python $R/bench.py $A > $R/gpurun_out/prof_large/bench_ktrace.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_large -o pmc_fetch -- python $R/bench.py $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_large -o pmc_write -- python $R/bench.py $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $R/gpurun_out/prof_large -o pmc_sq -- python $R/bench.py $A > /dev/null 2>&1
cd $R
python scripts/roofline_table.py gpurun_out/prof_large > gpurun_out/prof_large/roofline_table.txt 2>&1
cat gpurun_out/prof_large/roofline_table.txt
tail -1 gpurun_out/prof_large/bench_ktrace.json | cut -c1-300
