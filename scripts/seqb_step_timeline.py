#!/usr/bin/env python3
"""A clock on the links of the bf16 MFMA recurrence's forward step (csrc/lstm_seq_bf16.hip, seqb_fwd_body): runs the instrumented
builds of scripts/seqb_step_timeline.sh (one stamp point each) on the four encoder LSTMs of the MOSI shapes in one launch and prints,
for the waves of batch tile 0 of the h = 120 LSTM, the average shader-clock distance of every point from the top of the step.

    bash scripts/seqb_step_timeline.sh && python scripts/seqb_step_timeline.py [B=1024] [T=20] [only_ef=0]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POINTS = {1: "LDS reads returned (x-proj, record, h)", 2: "first barrier passed", 3: "global loads + stores issued",
          4: "recurrent product issued", 5: "gates, c, h computed", 6: "LDS writes done (+ prefetch waited for)", 7: "second barrier passed"}

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from factorized_amd import _lib
from factorized_amd import engine as E
B, T, only = %(B)d, %(T)d, %(only)d
ENC = [120] if only else [32, 8, 80, 120]
torch.manual_seed(0)
keep, descs = [], []
for h in ENC:
    Hp = (h + 15) // 16 * 16
    k = 1.0 / np.sqrt(h)
    gates = (torch.randn(T, B, 4, Hp, device="cuda") * 0.5).to(torch.bfloat16)
    hs = torch.zeros(T, B, Hp, device="cuda", dtype=torch.bfloat16)
    cs = torch.zeros(T, B, Hp, device="cuda")
    w = (torch.rand(4 * h, h, device="cuda") * 2 - 1) * k
    pack = torch.zeros(_lib.lib().mfm_lstm_pack_bytes(h, 0), dtype=torch.uint8, device="cuda")
    hl = torch.zeros(B, Hp, device="cuda")
    keep.append((gates, hs, cs, w, pack, hl))
    descs.append(E.make_seq(gates, hs, cs, w, h, w_pack=pack, store_bf16=True, h_last=hl))
L = _lib.lib()
arr = (_lib.SeqDesc * len(descs))(*descs)
_lib.check(L.mfm_lstm_pack_bf16(arr, len(descs), None), "pack")
for _ in range(3): _lib.check(L.mfm_lstm_seq_fwd_bf16(arr, len(descs), T, B, None), "fwd")
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): L.mfm_lstm_seq_fwd_bf16(arr, len(descs), T, B, None)
b.record(); torch.cuda.synchronize()
cs = keep[-1][2]
print(json.dumps(dict(us=1e3 * a.elapsed_time(b) / 20, stamps=[float(v) for v in cs[T - 1, 0, :8].cpu()])))
"""


def run(lib, B, T, only):
    env = dict(os.environ)
    if lib:
        env["MFM_LIB_PATH"] = lib
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, B=B, T=T, only=only)], env=env, capture_output=True, text=True)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    only = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    base = run(None, B, T, only)
    print("%s, B=%d T=%d: %.2f us per launch = %.3f us per step (uninstrumented build)"
          % ("h = 120 LSTM alone" if only else "encoders l / a / v / ef in one launch", B, T, base["us"], base["us"] / T))
    print("average shader clocks from the top of the step, waves 0-7 of tile 0 of the h = 120 LSTM; launch time of the instrumented build")
    prev = 0.0
    for k in range(1, 8):
        lib = os.path.join(ROOT, "scripts", "tmp", "stampb", "libmfm_hip_stampb%d.so" % k)
        if not os.path.exists(lib):
            print("missing", lib)
            continue
        r = run(lib, B, T, only)
        st = sorted(r["stamps"])
        med = st[len(st) // 2]
        print("P%d %-40s median %6.0f (+%5.0f)  min %6.0f max %6.0f   launch %7.2f us" % (k, POINTS[k], med, med - prev, st[0], st[-1], r["us"]))
        prev = med


if __name__ == "__main__":
    main()
