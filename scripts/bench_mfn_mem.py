#!/usr/bin/env python3
"""Microbenchmark of the MFN memory recurrence launches (mfn_mem.hip) alone: forward / backward time against T, to split
the per-launch fixed part (weights into registers, heads) from the per-step part.  usage: python scripts/bench_mfn_mem.py"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from factorized_amd import _lib, engine as E
from factorized_amd.mfm_model import _MemFn


def run(T, B=32, M=64, H1=128, H2=128, iters=200, bwd=False):
    dev = "cuda"
    g = lambda *s: torch.randn(*s, device=dev) * 0.3
    a1, a2, chat = g(T, B, H1), g(T, B, H2), torch.tanh(g(T, B, M))
    # the plan's layout: the memory-column blocks of the gamma nets' first layers have row stride ldw = M here
    w1m, w2m, w1b, b1b, w2b, b2b = g(H1, M), g(H2, M), g(M, H1), g(M), g(M, H2), g(M)
    gam1, gam2, mems = (torch.empty(T, B, M, device=dev) for _ in range(3))
    mem_out = torch.empty(B, M, device=dev)
    d = _MemFn._desc(T, B, M, H1, H2, a1, a2, chat, w1m, w2m, w1b, b1b, w2b, b2b, gam1, gam2, mems, 0.0, 0.0, True, 1, mem_out=mem_out)
    L = _lib.lib()
    fn = L.mfm_mfn_mem_fwd
    if bwd:
        _lib.check(fn(C.byref(d), E._stream()))
        dmem = g(B, M)
        du1, du2, dchat = torch.empty(T, B, H1, device=dev), torch.empty(T, B, H2, device=dev), torch.empty(T, B, M, device=dev)
        d = _MemFn._desc(T, B, M, H1, H2, a1, a2, chat, w1m, w2m, w1b, b1b, w2b, b2b, gam1.clone(), gam2.clone(), mems, 0.0, 0.0, True, 1,
                         dmem=dmem, du1=du1, du2=du2, dchat=dchat)
        fn = L.mfm_mfn_mem_bwd
    s = E._stream()
    for _ in range(20):
        _lib.check(fn(C.byref(d), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn(C.byref(d), s)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    for bwd in (False, True):
        print("backward" if bwd else "forward", " ".join("T=%d: %.1f us" % (T, run(T, bwd=bwd)) for T in (1, 5, 10, 20, 40)))
