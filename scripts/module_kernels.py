"""Run N eager module-path steps of one class (for `rocprofv3 --kernel-trace --stats`): which kernels make up
the step of MFM / MFM_KL."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, synth  # noqa: E402
from factorized_amd import mfm_model as M  # noqa: E402

cls = getattr(M, sys.argv[1] if len(sys.argv) > 1 else "MFM_KL")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfgs = C.canonical_configs(dropout=True)
cfg = cfgs[0]
m = cls(*cfgs).cuda()
m.train()
opt = torch.optim.Adam(m.parameters())
xn, yn = synth.make_batch(cfg["input_dims"], 32, 20, seed=3)
x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
l1, mse = torch.nn.L1Loss(), torch.nn.MSELoss()
d = cfg["input_dims"]
for _ in range(N):
    opt.zero_grad()
    (xl, xa, xv, yh), reg, miss = m.forward(x)
    loss = l1(yh.squeeze(1), y) + cfg["lda_xl"] * mse(xl, x[:, :, :d[0]]) + cfg["lda_xa"] * mse(xa, x[:, :, d[0]:d[0] + d[1]]) \
        + cfg["lda_xv"] * mse(xv, x[:, :, d[0] + d[1]:]) + cfg["lda_mmd"] * reg + miss
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done", N)
