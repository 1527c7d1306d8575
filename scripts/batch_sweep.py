#!/usr/bin/env python3
"""Batch sweep of the fused MFM_KL_EF training step on one MI355X: samples/s, step TFLOP/s and the
fraction of the fp32 matrix/vector peak (157.3 TF) versus per-GPU batch size, for both recurrent
kernel families.  At the reference's B=32 the step is latency-bound (80 serial LSTM steps); the
roofline fraction only becomes meaningful as B grows."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, engine, synth  # noqa: E402

PEAK = 157.3          # fp32 matrix == fp32 vector peak, TFLOP/s (MI355X_MICROARCH.md)
PEAK_BF16 = 2500.0    # dense bf16 MFMA peak


def run(B, path, steps, shape="mosi", T=20):
    if path in ("bf16", "auto"):          # the plan's own kernel selection (fp32 "auto", bf16 plan)
        os.environ.pop("MFM_SEQ_PATH", None)
    else:
        os.environ["MFM_SEQ_PATH"] = path
    cfgs = {"mosi": C.canonical_configs, "you": C.you_configs, "mosei": C.mosei_configs}[shape](dropout=True)
    e = engine.MFMEngine(cfgs, precision="bf16" if path == "bf16" else "fp32")
    e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
    ce = cfgs[0].get("loss", "l1") == "ce"
    xn, yn = synth.make_batch(cfgs[0]["input_dims"], B, T, seed=3, output_dim=cfgs[0]["output_dim"],
                              classes=cfgs[0]["output_dim"] if ce else 0)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    for _ in range(5):
        e.train_step(x, y, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e.train_step(x, y, check=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    w = e.work_per_step(T, B)
    peak = PEAK_BF16 if path == "bf16" else PEAK
    return dict(B=B, path=path, ms=1e3 * dt, samples_per_s=B / dt, tflops=w["flops"] / dt / 1e12,
                frac=w["flops"] / dt / 1e12 / peak, hbm_gbs=w["bytes"] / dt / 1e9)


if __name__ == "__main__":
    Bs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 128, 512, 2048, 8192]
    shape = sys.argv[2] if len(sys.argv) > 2 else "mosi"
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    print("# shape %s, T=%d; frac = fraction of the dtype's dense matrix peak: fp32 rows (auto) of 157.3 TF, bf16 rows of 2500 TF; alg GB/s = "
          "algorithmic bytes per step / time (peak ~8000)" % (shape, T))
    print("%6s %6s %9s %12s %8s %8s %9s" % ("B", "path", "ms/step", "samples/s", "TFLOP/s", "frac", "alg GB/s"))
    for B in Bs:
        for path in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("auto", "bf16")):
            r = run(B, path, steps=50 if B <= 2048 else 10, shape=shape, T=T)
            print("%6d %6s %9.3f %12.0f %8.2f %8.4f %9.1f" % (r["B"], r["path"], r["ms"], r["samples_per_s"],
                                                              r["tflops"], r["frac"], r["hbm_gbs"]))
            sys.stdout.flush()
