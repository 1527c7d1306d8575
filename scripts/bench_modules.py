#!/usr/bin/env python3
"""Module-path (drop-in) step times of the three model classes on one GPU: reference-style driver loop
(model.forward, torch losses, loss.backward(), torch.optim.Adam.step()), B=32, T=20, canonical sizes.
MFM_KL_EF is also timed on the fused one-call path for comparison.  `MFM_MFN_LOOP=1` forces the
step-by-step MFN memory loop (the pre-kernel implementation) for an A/B."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, synth  # noqa: E402
from factorized_amd.mfm_model import MFM, MFM_KL, MFM_KL_EF  # noqa: E402


def run(cls, steps=30):
    cfgs = C.canonical_configs(dropout=True)
    cfg = cfgs[0]
    m = cls(*cfgs).cuda()
    m.train()
    opt = torch.optim.Adam(m.parameters())
    xn, yn = synth.make_batch(cfg["input_dims"], 32, 20, seed=3)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    l1, mse = torch.nn.L1Loss(), torch.nn.MSELoss()

    def step():
        opt.zero_grad()
        (xl, xa, xv, yh), reg, miss = m.forward(x)
        d = cfg["input_dims"]
        loss = l1(yh.squeeze(1), y) + cfg["lda_xl"] * mse(xl, x[:, :, :d[0]]) + cfg["lda_xa"] * mse(xa, x[:, :, d[0]:d[0] + d[1]]) \
            + cfg["lda_xv"] * mse(xv, x[:, :, d[0] + d[1]:]) + cfg["lda_mmd"] * reg + miss
        loss.backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def run_graphed(cls, steps=100):
    from factorized_amd import train
    cfgs = C.canonical_configs(dropout=True)
    cfg = cfgs[0]
    m = cls(*cfgs).cuda()
    m.train()
    xn, yn = synth.make_batch(cfg["input_dims"], 32, 20, seed=3)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    gs = train.GraphedModuleStep(m, cfg, 32, 20)
    for _ in range(5):
        gs.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gs.step(x, y)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


if __name__ == "__main__":
    print("%-12s %10s %12s %18s" % ("class", "ms/step", "samples/s", "hipGraph ms/step"))
    for cls in (MFM_KL_EF, MFM_KL, MFM):
        ms = run(cls)
        gms = run_graphed(cls) if cls is not MFM_KL_EF else float("nan")
        print("%-12s %10.3f %12.0f %18.3f" % (cls.__name__, ms, 32 / ms * 1e3, gms))
