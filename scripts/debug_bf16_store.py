"""per-tensor gradient error of a bf16 plan against the fp32 oracle: bf16-resident storage vs fp32 storage (MFM_BF16_STORE=0)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from oracle import mfm_oracle as O
from factorized_amd import engine, synth
name = sys.argv[1] if len(sys.argv) > 1 else "klef_b5_t1"
os.environ["MFM_BF16_SEQ_MINB"] = "1"
cs = cases.load_case(name)
x, y = torch.from_numpy(cs["x"]), torch.from_numpy(cs["y"])
m = O.build("kl_ef", cs["cfgs"]); w = synth.make_weights(O.state_shapes(m), seed=1234); O.load_numpy_weights(m, w); m.train()
O.loss_terms(m, x, y, cs["cfg"], cs["loss_kind"])["loss"].backward()
res = {}
for store in ("1", "0"):
    os.environ["MFM_BF16_STORE"] = store
    e = engine.MFMEngine(cs["cfgs"], precision="bf16"); e.load_weights(w)
    e.forward(x.cuda(), y.cuda(), train=True); e.backward(x.cuda(), y.cuda(), stage=0)
    res[store] = {n: v.cpu().numpy().astype(np.float64) for n, v in e.grad_views().items()}
for n, p in m.named_parameters():
    r = p.grad.numpy().astype(np.float64); nr = np.linalg.norm(r)
    if nr < 1e-12: continue
    a, b = np.linalg.norm(res["1"][n] - r) / nr, np.linalg.norm(res["0"][n] - r) / nr
    if a > 0.02 or b > 0.02: print("%-32s resident %.4f  fp32-stored %.4f  |ref| %.3e" % (n, a, b, nr))

# intermediate buffers of decoder a (index n_enc + 1) in both storage modes
bufs = {}
for store in ("1", "0"):
    os.environ["MFM_BF16_STORE"] = store
    e = engine.MFMEngine(cs["cfgs"], precision="bf16"); e.load_weights(w)
    e.forward(x.cuda(), y.cuda(), train=True)
    T, B = cs["T"], cs["B"]
    f = {k: (v.float().cpu().numpy().copy() if torch.is_tensor(v) else v) for k, v in e.seq_buffers(T, B, 5).items()}
    e.backward(x.cuda(), y.cuda(), stage=0)
    b = {k: (v.float().cpu().numpy().copy() if torch.is_tensor(v) else v) for k, v in e.seq_buffers(T, B, 5).items()}
    bufs[store] = (f, b)
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
f1, b1 = bufs["1"]; f0, b0 = bufs["0"]
d = f1["d"]
print("fwd gates", rel(f1["gates"], f0["gates"]), "hs", rel(f1["hs"], f0["hs"]), "cs", rel(f1["cs"], f0["cs"]))
print("dxhat", rel(b1["dxhat"][:, :d], b0["dxhat"][:, :d]), "pad", np.abs(b1["dxhat"][:, d:]).max() if b1["dxhat"].shape[1] > d else 0)
print("dhs", rel(b1["dhs"], b0["dhs"]), "dA", rel(b1["gates"], b0["gates"]))
print("dhs resident", b1["dhs"][0, 0, :8], "\ndhs fp32    ", b0["dhs"][0, 0, :8])
