"""Throughput of the grouped fp32 MFMA GEMM on a few large single problems (TF/s against the 157.3 TF/s fp32
matrix peak); MFM_GEMM_FR=1|2 forces 32x32 / 64x64 tiles.  Results: profiles/r01l_gemm_large.txt."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_gemm import run
print("FR", os.environ.get("MFM_GEMM_FR"))
for (M, N, K) in [(40960, 960, 325), (4096, 4096, 4096), (8192, 8192, 512), (40960, 240, 325)]:
    us = run(M, N, K, iters=20)
    print("%6d %6d %6d  %9.1f us  %6.1f TF/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
