import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
from bench_gemm import run
print("FR", os.environ.get("MFM_GEMM_FR"))
for (M, N, K) in [(40960, 960, 325), (4096, 4096, 4096), (8192, 8192, 512), (40960, 240, 325)]:
    us = run(M, N, K, iters=20)
    print("%6d %6d %6d  %9.1f us  %6.1f TF/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
