set -x
mkdir -p gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2 -o ktrace -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof2/bench_ktrace.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof2 -o pmc_fetch -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof2 -o pmc_write -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $R/gpurun_out/prof2 -o pmc_sq -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
ls -la $R/gpurun_out/prof2
