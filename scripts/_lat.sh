cd $GRAFT_REPO_ROOT
for B in 2048 4096; do for S in 0 1; do
echo "== B=$B split=$S"; MFM_LATENT_SPLIT0=$S python bench.py --dtype bf16 --batch $B --steps 40 --warmup 10 --no-cpu-baseline --breakdown 2>&1 | grep -i "latent\|ms_per_step" | cut -c1-220
done; done
echo "== fp32 B=2048"; for S in 0 1; do MFM_LATENT_SPLIT0=$S python bench.py --batch 2048 --steps 30 --warmup 5 --no-cpu-baseline --breakdown 2>&1 | grep -i "latent\|ms_per_step" | cut -c1-200; done
timeout 900 python -m pytest tests/test_gpu_large_batch.py tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -3
