import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from factorized_amd import configs as C, engine, synth, train
cfgs = C.canonical_configs(dropout=True); cfg = cfgs[0]
e = engine.MFMEngine(cfgs, device="cuda:0")
e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
data = train.DeviceDataset(cfg, 1280, 20, 32, e.device, seed=11)
x, y = data.batch(0)
xs, ys = x.clone(), y.clone()
for i in range(20): e.train_step(xs, ys, lr=1e-3, check=False)
torch.cuda.synchronize()
def timeit(fn, n=500):
    for i in range(30): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
def eager(i):
    xb, yb = data.batch(i % data.nb); e.train_step(xb, yb, lr=1e-3, check=False)
print("eager fused step   %.4f ms" % timeit(eager))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3): e.train_step(xs, ys, lr=1e-3, check=False)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    e.train_step(xs, ys, lr=1e-3, check=False)
def replay(i):
    xb, yb = data.batch(i % data.nb); xs.copy_(xb); ys.copy_(yb); g.replay()
print("graph replay (+copy) %.4f ms" % timeit(replay))
def replay_only(i): g.replay()
print("graph replay only  %.4f ms" % timeit(replay_only))
