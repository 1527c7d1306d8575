#!/usr/bin/env python3
"""A clock on the links of the one-row LSTM forward step (csrc/lstm_seq_small.hip, small_fwd_body<KQ, 1>): runs the instrumented
builds of scripts/seq_step_timeline.sh (one stamp point each) on one recurrence and prints, per wave of workgroup 0, the average
shader-clock distance of every point from the top of the step, the segment lengths, and the un-instrumented launch time next to
the instrumented ones (how much the stamps perturb).

    bash scripts/seq_step_timeline.sh && python scripts/seq_step_timeline.py [h=104] [dec=1] [B=32] [T=40]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POINTS = {1: "h_{t-1} read from LDS", 2: "recurrent FMAs issued", 3: "gates, c, h computed", 4: "LDS writes acknowledged",
          5: "barrier released", 6: "next step's top (period)"}

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from factorized_amd import engine as E
h, dec, B, T = %(h)d, %(dec)d, %(B)d, %(T)d
Hp = (h + 15) // 16 * 16
torch.manual_seed(0)
g = torch.randn(T, B, 4, Hp, device="cuda") * 0.5
hs = torch.zeros(T, B, Hp, device="cuda"); cs = torch.zeros(T, B, Hp, device="cuda")
k = 1.0 / np.sqrt(h)
w = (torch.rand(4 * h, h, device="cuda") * 2 - 1) * k
wi = (torch.rand(4 * h, h, device="cuda") * 2 - 1) * k
bi = torch.zeros(4 * h, device="cuda"); bh = torch.zeros(4 * h, device="cuda")
init = torch.randn(B, h, device="cuda")
d = E.make_seq(g, hs, cs, w, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init, is_dec=True) if dec else E.make_seq(g, hs, cs, w, h)
os.environ["MFM_SEQ_PATH"] = "small"
call = lambda: E.lstm_seq([d], T, B)
for _ in range(5): call()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): call()
b.record(); torch.cuda.synchronize()
nw = 8 * Hp // 64
print(json.dumps(dict(us=1e3 * a.elapsed_time(b) / 100, waves=nw, stamps=[float(v) for v in cs[T - 1, 0, :nw].cpu()])))
"""


def run(lib, h, dec, B, T):
    env = dict(os.environ)
    if lib:
        env["MFM_LIB_PATH"] = lib
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, h=h, dec=dec, B=B, T=T)], env=env, capture_output=True, text=True)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 104
    dec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    base = run(None, h, dec, B, T)
    base1 = run(None, h, dec, B, 1)
    per_step_us = (base["us"] - base1["us"]) / (T - 1)
    print("%s h=%d B=%d T=%d: %.2f us per launch, T=1 %.2f us -> %.3f us per time step (uninstrumented build)"
          % ("decoder" if dec else "encoder", h, B, T, base["us"], base1["us"], per_step_us))
    res = {}
    for k in range(1, 7):
        lib = os.path.join(ROOT, "scripts", "tmp", "stamp", "libmfm_hip_stamp%d.so" % k)
        if not os.path.exists(lib):
            print("missing", lib)
            continue
        res[k] = run(lib, h, dec, B, T)
    nw = base["waves"]
    print("\naverage shader clocks from the top of the step (P0), per wave of workgroup 0; launch time of the instrumented build")
    print("%-28s %8s %8s %8s %8s   %s" % ("point", "min", "median", "max", "launch", "per wave"))
    med = {}
    for k, r in res.items():
        st = sorted(r["stamps"])
        med[k] = st[len(st) // 2]
        print("%-28s %8.0f %8.0f %8.0f %7.2fus   %s" % ("P%d %s" % (k, POINTS[k]), st[0], med[k], st[-1], r["us"],
                                                        " ".join("%.0f" % v for v in r["stamps"])))
    if len(med) == 6:
        period = med[6]
        clk = period / per_step_us / 1e3          # GHz
        print("\nperiod %.0f clocks = %.3f us per step -> shader clock %.2f GHz" % (period, per_step_us, clk))
        segs = [("LDS hand-over: barrier release -> h_{t-1} in registers", med[1]), ("recurrent product (FMAs)", med[2] - med[1]),
                ("quad reduce + gates + c + h (DPP, transcendentals)", med[3] - med[2]),
                ("LDS writes of the step acknowledged", med[4] - med[3]), ("barrier (slowest wave arrives)", med[5] - med[4]),
                ("record written out, next step's top", period - med[5])]
        for name, v in segs:
            print("  %-58s %6.0f clocks  %5.1f %%  %.3f us" % (name, v, 100 * v / period, v / clk / 1e3))


if __name__ == "__main__":
    main()
