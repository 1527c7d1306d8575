# Round-4 profile of the large-batch bf16 (bf16-resident) step: rocprofv3 kernel trace + separate PMC passes -> roofline table.
# usage (on the GPU box through gpurun): bash scripts/profile_round4_large.sh <tag> [extra bench args]
set -x
R=$GRAFT_REPO_ROOT
TAG=${1:-l_bf16}; shift
O=$R/gpurun_out/prof_r04/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--dtype bf16 --batch 2048 --steps 30 --warmup 5 --no-cpu-baseline $@"
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o ktrace -- python $R/bench.py $A > $O/bench_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O -o pmc_fetch -- python $R/bench.py $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O -o pmc_write -- python $R/bench.py $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $O -o pmc_sq -- python $R/bench.py $A > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $O -o pmc_lds -- python $R/bench.py $A > /dev/null 2>&1
cd $R
python scripts/roofline_table.py $O > $O/roofline_table.txt 2>&1
python scripts/rocprof_pmc_summary.py $(ls $O/pmc_lds*.db | head -1) > $O/pmc_lds.txt 2>&1
python scripts/rocprof_summary.py $(ls $O/ktrace*.db | head -1) > $O/kernel_stats.txt
rm -f $O/*.db $O/*/*.db 2>/dev/null
cat $O/roofline_table.txt; grep -i "dw_bf16\|gemm_panel" $O/pmc_lds.txt
