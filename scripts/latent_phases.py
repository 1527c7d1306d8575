#!/usr/bin/env python3
"""Print the phase timeline (shader-clock cycles of workgroup 0) of the latent kernels."""
import os
import sys

os.environ["MFM_LATENT_DBG"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import _lib, configs, engine, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfgs = configs.canonical_configs()
e = engine.MFMEngine(cfgs, precision=os.environ.get("MFM_PHASES_PRECISION", "fp32"))
e.load_weights(synth.make_weights(e.layout.shapes))
xn, yn = synth.make_batch(cfgs[0]["input_dims"], B, 20)
x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
for _ in range(3):
    e.forward(x, y, train=True, want_xhat=False)
    e.backward(x, y)
torch.cuda.synchronize()
p = e.plan(20, B)
off = _lib.lib().mfm_plan_debug_offset(p.handle)
ts = p.workspace[off:off + 64 * 8].view(torch.int64).cpu().numpy()
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = ts[:24]
print("latent_fwd (cycles, workgroup 0):")
print("  prologue(load inputs+ops) %d" % (f[1] - f[0]))
prev = f[1]
for s in range(NS):
    print("  stage %d: issue/copy %6d   compute %6d" % (s, f[2 + 2 * s] - prev, f[3 + 2 * s] - f[2 + 2 * s]))
    prev = f[3 + 2 * s]
print("  epilogue %d   total %d" % (f[20] - prev, f[20] - f[0]))
b = ts[24:24 + 1 + 3 * NS]
print("latent_bwd:")
prev = b[0]
for s in range(NS - 1, -1, -1):
    m1, m2, m3 = b[1 + 3 * s], b[2 + 3 * s], b[3 + 3 * s]
    print("  stage %d: copy+pass1 %6d   pass2a %6d   pass2b %6d" % (s, m1 - prev, m2 - m1, m3 - m2))
    prev = m3
print("  total after seeds %d" % (prev - b[0]))
