#!/usr/bin/env python3
"""Micro-benchmark of the grouped fp32 MFMA GEMM: us per launch vs K, to separate fixed cost from
per-K-tile cost."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import engine as E  # noqa: E402


def run(M, N, K, batch=1, iters=100, nprob=1, bf16=False, tn=False):
    """tn: the weight-gradient layout C[M,N] = A^T B with A [K,M], B [K,N] (both contiguous along m / n), split-K"""
    descs, keep = [], []
    for _ in range(nprob):
        if tn:
            a = torch.randn(K, M, device="cuda"); b = torch.randn(K, N, device="cuda")
            c = torch.zeros(M, N, device="cuda")
            descs.append(E.make_gemm(a, b, c, M, N, K, a_sm=1, a_sk=M, b_sk=N, b_sn=1, ldc=N, accumulate=1, split_k=0))
        else:
            a = torch.randn(M, K, device="cuda"); b = torch.randn(batch * N, K, device="cuda")
            c = torch.empty(M, batch * N, device="cuda")
            descs.append(E.make_gemm(a, b, c, M, N, K, a_sm=K, a_sk=1, b_sk=1, b_sn=K, ldc=batch * N, batch=batch,
                                     b_sz=N * K, c_sz=N))
        keep.append((a, b, c))
    if bf16:
        from factorized_amd import _lib
        arr = (_lib.GemmDesc * len(descs))(*descs)
        call = lambda: _lib.check(_lib.lib().mfm_gemm_grouped_bf16(arr, len(descs), None), "gemm bf16")
    else:
        call = lambda: E.gemm_grouped(descs)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        call()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / iters


if __name__ == "__main__":
    print("%6s %6s %6s %6s %6s %10s" % ("M", "N", "K", "batch", "nprob", "us"))
    for (M, N, K, batch, nprob) in [(640, 128, 32, 1, 1), (640, 128, 128, 1, 1), (640, 128, 325, 1, 1), (640, 128, 1300, 1, 1),
                                    (640, 128, 325, 4, 1), (640, 128, 325, 4, 4), (32, 32, 32, 1, 1), (32, 32, 32, 1, 20),
                                    (4096, 512, 325, 1, 1)]:
        print("%6d %6d %6d %6d %6d %10.1f" % (M, N, K, batch, nprob, run(M, N, K, batch, nprob=nprob)))
