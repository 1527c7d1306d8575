#!/usr/bin/env python3
"""Micro-benchmark of the LSTM sequence kernels (both families) on one GPU:
prints us per launch for (h, B, T) so per-step and prologue costs can be separated."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import engine as E  # noqa: E402


def run(h, B, T, path, bwd, dec=False, iters=50):
    bf16 = path.startswith("bf16")
    if not bf16:
        os.environ["MFM_SEQ_PATH"] = path
    Hp = (h + 15) // 16 * 16
    g = torch.randn(T, B, 4, Hp, device="cuda") * 0.5
    hs = torch.zeros(T, B, Hp, device="cuda"); cs = torch.zeros(T, B, Hp, device="cuda")
    k = 1.0 / np.sqrt(h)
    w = (torch.rand(4 * h, h, device="cuda") * 2 - 1) * k
    wi = (torch.rand(4 * h, h, device="cuda") * 2 - 1) * k
    bi = torch.zeros(4 * h, device="cuda"); bh = torch.zeros(4 * h, device="cuda")
    init = torch.randn(B, h, device="cuda")
    dh = torch.randn(T, B, Hp, device="cuda") if dec else torch.randn(B, h, device="cuda")
    dinit = torch.zeros(B, h, device="cuda")
    if dec:
        d = E.make_seq(g, hs, cs, w, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init, is_dec=True,
                       dh_ext=dh, ld_dh=Hp, d_h_init=dinit)
    else:
        d = E.make_seq(g, hs, cs, w, h, dh_ext=dh, ld_dh=h)
    call = lambda: E.lstm_seq([d], T, B, backward=bwd)
    if bf16:
        from factorized_amd import _lib
        L = _lib.lib()
        if path == "bf16":           # "bf16-nopack": the kernels gather their weight fragments themselves
            pack = torch.zeros(L.mfm_lstm_pack_bytes(h, int(dec)), dtype=torch.uint8, device="cuda")
            d.w_pack = pack.data_ptr()
        arr = (_lib.SeqDesc * 1)(d)
        if path == "bf16":
            _lib.check(L.mfm_lstm_pack_bf16(arr, 1, None), "pack")
        fn = L.mfm_lstm_seq_bwd_bf16 if bwd else L.mfm_lstm_seq_fwd_bf16
        call = lambda: _lib.check(fn(arr, 1, T, B, None), "seq bf16")
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


if __name__ == "__main__":
    hs_ = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["120", "32"])]
    Bs = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32"])]
    print("%5s %5s %4s %6s %4s %4s %10s" % ("h", "B", "T", "path", "bwd", "dec", "us"))
    for h in hs_:
        for B in Bs:
            for path in os.environ.get("BENCH_SEQ_PATHS", "mfma,small").split(","):
                for bwd in [bool(int(v)) for v in os.environ.get("BENCH_SEQ_BWD", "0,1").split(",")]:
                    for dec in (False, True):
                        for T in (1, 20, 40):
                            us = run(h, B, T, path, bwd, dec)
                            print("%5d %5d %4d %6s %4d %4d %10.1f" % (h, B, T, path, bwd, dec, us))
