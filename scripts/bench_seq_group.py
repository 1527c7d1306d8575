#!/usr/bin/env python3
"""Per-time-step cost of the one-row recurrences, alone and in the groups the plan launches them in (decoders 104 / 24 / 24,
encoders 32 / 8 / 80 / 120): us per launch at T = 1, 20, 40 and the slope, forward and backward.  Tells a slow multi-body kernel
instance (the grouped launch) from a slow recurrence."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import engine as E  # noqa: E402

os.environ["MFM_SEQ_PATH"] = "small"
keep = []


def desc(h, B, T, dec, bwd):
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
    keep.extend([g, hs, cs, w, wi, bi, bh, init, dh, dinit])
    if dec:
        return E.make_seq(g, hs, cs, w, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init, is_dec=True, dh_ext=dh, ld_dh=Hp, d_h_init=dinit)
    return E.make_seq(g, hs, cs, w, h, dh_ext=dh, ld_dh=h)


def time_group(hs_, B, T, dec, bwd, iters=100):
    ds = [desc(h, B, T, dec, bwd) for h in hs_]
    if bwd:          # the backward overwrites the saved gates in place: run a forward first so that they are activations
        E.lstm_seq(ds, T, B)
    call = lambda: E.lstm_seq(ds, T, B, backward=bwd)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print("%-28s %4s %8s %8s %8s %10s" % ("recurrences", "dir", "T=1", "T=20", "T=40", "us/step"))
for name, hs_, dec in (("dec 104", [104], True), ("dec 24", [24], True), ("dec 104+24+24 (plan)", [104, 24, 24], True),
                       ("enc 120", [120], False), ("enc 80", [80], False), ("enc 32", [32], False), ("enc 8", [8], False),
                       ("enc 32+8+80+120 (plan)", [32, 8, 80, 120], False)):
    for bwd in (False, True):
        t = [time_group(hs_, B, T, dec, bwd) for T in (1, 20, 40)]
        print("%-28s %4s %8.2f %8.2f %8.2f %10.3f" % (name, "bwd" if bwd else "fwd", t[0], t[1], t[2], (t[2] - t[1]) / 20))
