"""Throughput of the grouped MFMA GEMMs on a few large single problems: fp32 operands (TF/s against the 157.3 TF/s
fp32 matrix peak) and bf16 operands (fp32 buffers rounded on the way into LDS; HBM-bound, GB/s of algorithmic traffic),
per tile size MFM_GEMM_FR=1|2|4 (32 / 64 / 128 square tiles); NT = forward layout, TN = weight-gradient layout."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_gemm import run
for fr in ("1", "2", "4"):
    os.environ["MFM_GEMM_FR"] = fr
    for bf16 in (False, True):
        print("FR", fr, "bf16-operand" if bf16 else "fp32")
        for (M, N, K, tn) in [(40960, 960, 325, False), (40960, 512, 325, False), (4096, 4096, 4096, False), (8192, 8192, 512, False),
                              (480, 325, 40960, True), (480, 120, 40960, True)]:
            us = run(M, N, K, iters=20, bf16=bf16, tn=tn)
            gb = 4.0 * (M * K + N * K + M * N) / us / 1e3
            print("%2s %6d %6d %6d  %9.1f us  %6.1f TF/s  %7.1f GB/s" % ("TN" if tn else "NT", M, N, K, us, 2.0 * M * N * K / us / 1e6, gb))
