"""Fused single-GPU step against the data-parallel split (mfm_plan_grad_step + mfm_adam_flat as separate
calls) on one GPU: the split itself must cost nothing, so that the exchange is the only data-parallel overhead."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import configs as C, engine, synth, train
cfgs = C.canonical_configs(dropout=True); cfg = cfgs[0]
e = engine.MFMEngine(cfgs, device="cuda:0")
e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
data = train.DeviceDataset(cfg, 1280, 20, 32, e.device, seed=11)
def fused(i):
    x, y = data.batch(i % data.nb); e.train_step(x, y, lr=1e-3, check=False)
def split(i):
    x, y = data.batch(i % data.nb); e.grad_step(x, y, check=False); e.adam(lr=1e-3, grad_scale=1.0)
for name, fn in (("fused train_step", fused), ("grad_step + adam", split), ("fused train_step", fused), ("grad_step + adam", split)):
    for i in range(30): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400): fn(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-18s %.4f ms/step" % (name, 1e3 * dt / 400))
