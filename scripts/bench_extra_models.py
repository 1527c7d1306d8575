#!/usr/bin/env python3
"""Step times of the ablation / missing-modality classes (M_A-M_D, MFM_missing, seq2seq, basic_missing; reference
mfm_model.py:201-467, 766-1017) on the module path, canonical MOSI sizes, B=32, T=20: forward + a scalar objective over every
output + backward + Adam (torch.optim.Adam and factorized_amd.optim.Adam -- these classes have no flat gradient buffer, so both
run the stock per-tensor update), and forward only.  The reference has no driver for the ablations (train_ablation /
train_missing of mfm_mosi.py:505-1288 are loop variants on private data); the objective here touches every output once."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, synth  # noqa: E402
from factorized_amd import mfm_model as M  # noqa: E402

NAMES = ["M_A", "M_B", "M_C", "M_D", "MFM_missing", "seq2seq", "basic_missing"]


def flat(o, out):
    if torch.is_tensor(o):
        out.append(o)
    elif isinstance(o, (list, tuple)):
        for q in o:
            flat(q, out)
    return out


def objective(out):
    ts = [t for t in flat(out, []) if t.dtype.is_floating_point]
    return sum((t * t).mean() if t.dim() else t for t in ts)


def run(name, steps=40):
    cfgs = C.canonical_configs(dropout=True)
    cfg = cfgs[0]
    m = getattr(M, name)(*cfgs).cuda()
    m.train()
    opt = torch.optim.Adam(m.parameters())
    xn, _ = synth.make_batch(cfg["input_dims"], 32, 20, seed=3)
    x = torch.from_numpy(xn).cuda()

    def step():
        opt.zero_grad()
        objective(m.forward(x)).backward()
        opt.step()

    def fwd():
        with torch.no_grad():
            m.forward(x)
    res = []
    for fn in (step, fwd):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        res.append(1e3 * (time.perf_counter() - t0) / steps)
    # the same step captured once into a hipGraph and replayed (train.GraphedStep)
    from factorized_amd import train
    gs = train.GraphedStep(m, lambda mod, xx: objective(mod.forward(xx)), [x])
    for _ in range(5):
        gs.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gs.step(x)
    torch.cuda.synchronize()
    res.append(1e3 * (time.perf_counter() - t0) / steps)
    assert torch.isfinite(gs.loss).item()
    return res, sum(p.numel() for p in m.parameters())


if __name__ == "__main__":
    print("%-14s %10s %14s %12s %12s %16s %10s" % ("class", "params", "train ms/step", "samples/s", "forward ms", "graphed ms/step", "graphed/eager"))
    for n in NAMES:
        (tr, fw, gr), np_ = run(n)
        print("%-14s %10d %14.3f %12.0f %12.3f %16.3f %10.2f" % (n, np_, tr, 32 / tr * 1e3, fw, gr, gr / tr))
