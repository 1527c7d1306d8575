"""Counting loop for wrong hand-overs: grad_step with the role workgroups against the separate-launch gradients of the same
batch, N steps per shape; prints how many steps had a tensor off by more than 1e-4 of its largest gradient and which tensors
(profiles/r04_handover_safety.txt).  usage: python scripts/handover_flake.py [steps per shape]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from factorized_amd import configs as C, engine, synth
cfgs = C.canonical_configs(dropout=False)
cfg = cfgs[0]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
tot = 0
for (B, T) in [(32, 1), (16, 1), (32, 2), (32, 20)]:
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    ref = engine.MFMEngine(cfgs); ref.handover = False
    w = synth.make_weights(ref.layout.shapes, seed=1234); ref.load_weights(w)
    ref.grad_step(x, y); torch.cuda.synchronize()
    rg = ref.grads.clone()
    scale = {n: max(float(v.abs().max()), 1e-12) for n, v in ref.grad_views().items()}
    e = engine.MFMEngine(cfgs); e.load_weights(w)
    bad = 0; names = {}
    for r in range(reps):
        e.grad_step(x, y)
        d = (e.grads - rg).abs()
        if float(d.max()) > 1e-5:
            gv = e.layout.views(d)
            nb = [n for n, v in gv.items() if float(v.max()) > 1e-4 * scale[n] + 1e-7]
            if nb:
                bad += 1
                for n in nb: names[n] = names.get(n, 0) + 1
    tot += bad
    print("B=%d T=%d: %d bad of %d %s" % (B, T, bad, reps, names), flush=True)
print("TOTAL bad", tot)
