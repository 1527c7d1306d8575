cd $GRAFT_REPO_ROOT
for B in 256 512 1024 2048; do for M in 0 1; do echo "fp32 B=$B f32onepass=$M $(MFM_DW_F32_MINROWS=$M python bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --breakdown 2>&1 < /dev/null | grep -i 'dw\|ms_per_step' | cut -c1-150 | sed 's/{"metric.*"ms_per_step"/ms_per_step/' | tr '\n' ' ')"; done; done
