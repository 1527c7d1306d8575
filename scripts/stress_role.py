"""Long-run check of the in-launch hand-overs: N fused training steps with the role workgroups and N with the separate launches
(MFM_PROJ_FOLD=0 MFM_DW_FOLD=0 MFM_WT_IMG=0) on the same batches; losses must stay finite (a wait that gave up poisons them) and
the two parameter trajectories must agree.  usage: python scripts/stress_role.py [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, engine, synth      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfgs = C.canonical_configs(dropout=False)
B, T = 32, 20
batches = [synth.make_batch(cfgs[0]["input_dims"], B, T, seed=100 + i) for i in range(8)]
dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in batches]
out = []
for off in (False, True, True):      # the second `off` run measures what the atomics' summation order alone does to a trajectory
    for k in ("MFM_PROJ_FOLD", "MFM_DW_FOLD", "MFM_WT_IMG"):
        if off:
            os.environ[k] = "0"
        else:
            os.environ.pop(k, None)
    e = engine.MFMEngine(cfgs)
    e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
    tr = []
    for i in range(steps):
        x, y = dev[i % len(dev)]
        l = e.train_step(x, y, lr=1e-4)
        if i % 250 == 0 or i == steps - 1:
            tr.append(e.loss_dict(l)["loss"])
    torch.cuda.synchronize()
    out.append((np.array(tr), e.params.cpu().numpy().copy()))
    print("role workgroups %s: loss trace %s" % ("off" if off else "on", np.round(out[-1][0], 4)))
assert np.isfinite(out[0][0]).all() and np.isfinite(out[0][1]).all(), "NaN with the role workgroups (a wait gave up?)"
dl = np.abs(out[0][0] - out[1][0]).max() / np.abs(out[1][0]).max()
dp = np.abs(out[0][1] - out[1][1]).max()
nl = np.abs(out[2][0] - out[1][0]).max() / np.abs(out[1][0]).max()
npar = np.abs(out[2][1] - out[1][1]).max()
print("on vs off: max relative loss difference %.2e, max parameter difference %.2e over %d steps" % (dl, dp, steps))
print("off vs off (run-to-run, atomics order): %.2e, %.2e" % (nl, npar))
assert dl < max(10 * nl, 2e-2) and dp < max(10 * npar, 5e-2)
print("ok")
