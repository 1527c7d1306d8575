cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_large_batch.py tests/test_gpu_staged.py -m gpu -x -q 2>&1 | tail -4
python bench.py --dtype bf16 --batch 2048 --steps 40 --warmup 10 --no-cpu-baseline --breakdown 2>&1 | grep -i "latent\|ms_per_step" | cut -c1-160
python scripts/latent_phases.py 2048 7
for B in 512 1024; do python bench.py --batch $B --steps 40 --warmup 10 --no-cpu-baseline --breakdown 2>&1 | grep -i "latent\|ms_per_step" | cut -c1-160; done
