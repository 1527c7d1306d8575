#!/usr/bin/env python3
"""A clock on the LAUNCHES of the headline step (MFM_KL_EF, B <= 32: six launches).  Runs the stamp build of
scripts/launch_timeline.sh (csrc/lstamp.h: thread 0 of every workgroup stores the device-wide 100 MHz clock at a few points of its
life) on bench.py's workload, steps back to back, and prints for the LAST step of each of N samples -- medians over the samples --

  * per launch: first entry, last exit, the gap to the next launch's first entry, and per workgroup ROLE (projection producers,
    the rows of each encoder / decoder, weight-gradient producers) the median / min / max time of every stamped point, in us from
    the launch's first entry;
  * per recurrence: prologue, in-situ cost per time step, epilogue, so that prologue + T x step + tail adds up to the launch;
  * the step time of the stamp build next to the product build's (what the instrumentation costs).

    bash scripts/launch_timeline.sh && python scripts/launch_timeline.py [--batch 32] [--seq 20] [--samples 9] > profiles/r06_launch_timeline.txt
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "scripts", "tmp", "lstamp", "libmfm_hip_lstamp.so")
KNAMES = ["enc fwd (foldproj)", "dec fwd", "dec fc1 + MSE + dH", "dec BPTT", "enc BPTT (folddw)", "adam"]
PT = {
    0: {0: "entry", 1: "weights in registers", 2: "top of the time loop (x-proj of t=0,1 arrived)", 5: "top of step 1", 6: "top of step T-1",
        7: "time loop done (last record flushed)", 9: "all encoders' t=0 flags seen", 8: "stores acknowledged (sync_stores)",
        15: "latent fwd chain done = exit"},
    1: {0: "entry", 1: "W_ih in registers", 2: "top of step 0", 3: "step 0 done, record flushed", 4: "W_ih+W_hh in registers",
        5: "top of step 2", 6: "top of step T-1", 7: "time loop done", 15: "exit"},
    2: {0: "entry", 3: "first fragment's weights + targets requested", 4: "H rows parked (in front of the barrier)", 1: "operands arrived, H tile in LDS",
        5: "first fragment's operands arrived", 6: "first fragment: products + epilogue done", 7: "all fragments done",
        2: "product 1 + MSE done (loss reduced)", 8: "product 2: first weights arrived", 9: "product 2: first fragment multiplied", 15: "exit (dH added)"},
    3: {0: "entry", 1: "W^T in registers", 2: "top of the time loop (saved state of T-1 arrived)", 5: "top of step T-2", 6: "top of step 1",
        3: "top of step 0", 7: "done (d h_init written)", 15: "exit"},
    4: {0: "entry", 9: "latent bwd chain done", 8: "stores acknowledged, row stamped", 1: "W^T in registers",
        2: "top of the time loop", 5: "top of step T-2", 6: "top of step 1", 3: "top of step 0", 7: "BPTT done, t=0 stamped", 15: "exit"},
    5: {0: "entry", 15: "exit"},
}
for _s in range(8):
    PT[0][17 + _s] = "  latent fwd: stage %d done" % _s
    PT[4][18 + _s] = "  latent bwd: stage %d done" % _s
    PT[1][17 + _s] = "  latent fwd (tail block): stage %d done" % _s
    PT[3][18 + _s] = "  latent bwd (head block): stage %d done" % _s
PT[1][16] = "  latent fwd (tail block): tables + record in LDS"
PT[1][25] = "  latent fwd (tail block): losses reduced"
PT[3][8] = "  step 0: W_ih rows requested"
PT[3][9] = "  step 0: pointwise part done (dA_0 in LDS)"
PT[3][10] = "  step 0: dA_0 W_ih partial sums written (in front of the barrier)"
PT[3][11] = "  step 0: barrier passed"
PT[3][16] = "  latent bwd (head block): tables + records in LDS"
PT[3][17] = "  latent bwd (head block): seeds done"
PT[0][26] = "    stage 2: top"
PT[0][27] = "    stage 2: input segment read from LDS"
PT[0][28] = "    stage 2: next stage's weights requested"
PT[0][29] = "    stage 2: this stage's weights have arrived"
PT[0][30] = "    stage 2: product, reduce, output written (in front of the barrier)"
PT[0][16] = "  latent fwd: tables + h_T in LDS"
PT[0][25] = "  latent fwd: losses reduced"
PT[4][16] = "  latent bwd: tables + records in LDS"
PT[4][17] = "  latent bwd: seeds done"
SUB = ["A issued (stamps waited for if not prefetched)", "next block's stamps asked", "operands parked in LDS + barrier", "next operands requested",
       "product + epilogue done", "closing barrier"]
ORDER = {0: [0, 1, 2, 5, 6, 7, 9, 8, 16, 17, 18, 26, 27, 28, 29, 30, 19, 20, 21, 22, 23, 24, 25, 15], 1: [0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 15], 2: [0, 3, 4, 1, 5, 6, 7, 2, 8, 9, 15], 3: [0, 1, 2, 5, 6, 3, 8, 9, 10, 11, 7, 16, 17, 25, 24, 23, 22, 21, 20, 19, 18, 15],
         4: [0, 16, 17, 25, 24, 23, 22, 21, 20, 19, 18, 9, 8, 1, 2, 5, 6, 3, 7, 15], 5: [0, 15]}

CHILD_STEP = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import torch
from factorized_amd import configs as C, engine, synth, train
cfgs = C.canonical_configs(dropout=True); cfg = cfgs[0]
B, T = %(B)d, %(T)d
e = engine.MFMEngine(cfgs, device="cuda:0")
e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
data = train.DeviceDataset(cfg, 1280, T, B, e.device, seed=11)
st = train.DataParallelStep(e, 1, lr=1e-3, allreduce=None, rank=0)
def run(n):
    for i in range(n):
        x, y = data.batch(i %% data.nb); st.step(x, y)
run(100); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 400)
print(json.dumps({"ms": 1e3 * best}))
"""


def step_ms(lib, B, T):
    env = dict(os.environ)
    if lib:
        env["MFM_LIB_PATH"] = lib
    out = subprocess.run([sys.executable, "-c", CHILD_STEP % dict(root=ROOT, B=B, T=T)], env=env, capture_output=True, text=True)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-3000:])
    return json.loads(out.stdout.strip().splitlines()[-1])["ms"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=20)
    ap.add_argument("--samples", type=int, default=9)
    ap.add_argument("--steps", type=int, default=60, help="back-to-back steps per sample (the last one is read)")
    args = ap.parse_args()
    B, T = args.batch, args.seq
    if not os.path.exists(LIB):
        raise SystemExit("missing %s: run scripts/launch_timeline.sh" % LIB)
    prod_ms = step_ms(None, B, T)
    stamp_ms = step_ms(LIB, B, T)

    os.environ["MFM_LIB_PATH"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    from factorized_amd import configs as C, engine, synth, train
    dbg = ctypes.CDLL(LIB)
    nk, nb, npt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    dbg.mfm_debug_lstamp_dims(ctypes.byref(nk), ctypes.byref(nb), ctypes.byref(npt))
    nk, nb, npt = nk.value, nb.value, npt.value
    cfgs = C.canonical_configs(dropout=True)
    cfg = cfgs[0]
    e = engine.MFMEngine(cfgs, device="cuda:0")
    e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
    data = train.DeviceDataset(cfg, 1280, T, B, e.device, seed=11)
    st = train.DataParallelStep(e, 1, lr=1e-3, allreduce=None, rank=0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def run(n, first=0):
        for i in range(n):
            x, y = data.batch((first + i) % data.nb)
            st.step(x, y)

    run(100)
    torch.cuda.synchronize()
    buf = np.zeros((nk, nb, npt), dtype=np.uint64)
    dbg.mfm_debug_lstamp_read.argtypes = [ctypes.c_void_p]
    dbg.mfm_debug_lstamp_read(buf.ctypes.data)
    samples = []
    for s in range(args.samples):
        run(args.steps, s * args.steps)
        assert dbg.mfm_debug_lstamp_read(buf.ctypes.data) == 0
        a = buf.astype(np.float64)
        a[buf == 0] = np.nan
        # stamps of the last step only: a point written by an earlier step but not by the last (cannot happen: same control flow
        # every step) would show as an outlier far in the past
        t0 = np.nanmin(a[0, :, 0])
        samples.append((a - t0) / 100.0)          # us from the first workgroup entry of the step
    S = np.stack(samples)                          # [sample, kernel, block, point]
    med = np.nanmedian(S, axis=0)

    n_role_p = (cus - 4 * B) // 32 * 32
    enc_names = ["enc l (h=32)", "enc a (h=8)", "enc v (h=80)", "enc ef (h=120)"]
    dec_names = ["dec l (h=104)", "dec a (h=24)", "dec v (h=24)"]
    # dec_fc1.hip: 16-row tiles x column groups of 8 fragments, decoder after decoder (canonical MOSI widths)
    fc1_classes, _rt, _t0 = [], -(-T * B // 16), 0
    for _nm, _d in zip(("l (300 columns)", "a (5 columns)", "v (20 columns)"), (300, 5, 20)):
        _nf = -(-_d // 16)
        _fpg = min(8, _nf)
        _cg = -(-_nf // _fpg)
        fc1_classes.append(("fc1 of decoder %s: %d row tiles x %d column groups%s" % (_nm, _rt, _cg, " (dH added with atomics)" if _cg > 1 else " (dH stored)"),
                            _t0, min(_t0 + _rt * _cg, nb)))
        _t0 += _rt * _cg
    classes = {
        0: [("projection role workgroups", 0, n_role_p)] + [(enc_names[i], n_role_p + i * B, n_role_p + (i + 1) * B) for i in range(4)],
        1: [(dec_names[i], i * B, (i + 1) * B) for i in range(3)] + [("latent forward tail blocks (classifier, logvar heads, losses)", 3 * B, nb)],
        2: fc1_classes,
        3: [(dec_names[i], i * B, (i + 1) * B) for i in range(3)] + [("latent backward head blocks (disc / KLD seeds, classifier, logvar stages)", 3 * B, nb)],
        4: [(enc_names[i], i * B, (i + 1) * B) for i in range(4)] + [("weight-gradient role workgroups", 4 * B, nb)],
        5: [("all blocks", 0, nb)],
    }
    print("launch timeline of the MFM_KL_EF step, B=%d T=%d fp32 (scripts/launch_timeline.py; csrc/lstamp.h)" % (B, T))
    print("step time: product build %.4f ms, stamp build %.4f ms (+%.1f us of instrumentation); %d samples, medians; clock: 100 MHz "
          "device-wide counter (10 ns)" % (prod_ms, stamp_ms, 1e3 * (stamp_ms - prod_ms), args.samples))
    print()
    first = [np.nanmin(med[k, :, 0]) for k in range(nk)]
    last = [np.nanmax(med[k]) for k in range(nk)]
    print("%-22s %9s %9s %9s %9s" % ("launch", "first in", "last out", "span", "gap->next"))
    for k in range(nk):
        gap = first[k + 1] - last[k] if k + 1 < nk else float("nan")
        print("%-22s %9.2f %9.2f %9.2f %9.2f" % (KNAMES[k], first[k], last[k], last[k] - first[k], gap))
    print("(us from the step's first workgroup entry; the step's last exit -> next step's first entry is not seen by one step's stamps:\n"
          " step time - last out = %.2f us)" % (1e3 * stamp_ms - last[nk - 1]))
    for k in range(nk):
        print("\n== %s: us from this launch's first entry; median [min .. max] over the role's workgroups" % KNAMES[k])
        for name, lo, hi in classes[k]:
            blk = med[k, lo:hi]
            if blk.size == 0 or np.all(np.isnan(blk)):
                continue
            nwg = int(np.sum(~np.isnan(blk[:, 0])))
            print("  -- %s (%d workgroups)" % (name, nwg))
            pts = list(ORDER[k])
            if (k == 0 and name.startswith("projection")):
                pts = [0, 1] + list(range(16, 32)) + [12, 13, 15]
                labels = {0: "entry", 1: "W_ih block + x_t slice in LDS", 12: "items done", 13: "W^T images written", 15: "exit (zero spans cleared)"}
                labels.update({16 + i: "flag of item %d raised" % i for i in range(16)})
            elif (k == 4 and name.startswith("weight")):
                pts = [0] + list(range(16, 32)) + [15] + list(range(7, 13)) + list(range(1, 7))
                labels = {0: "entry", 15: "exit"}
                labels.update({16 + i: "top of table iteration %d" % i for i in range(16)})
                labels.update({7 + i: "  iteration 1: " + SUB[i] for i in range(6)})
                labels.update({1 + i: "  iteration 5: " + SUB[i] for i in range(6)})
            else:
                labels = PT[k]
            for p in pts:
                col = blk[:, p] - first[k]
                if np.all(np.isnan(col)):
                    continue
                print("     %-52s %8.2f  [%7.2f .. %7.2f]" % (labels.get(p, "point %d" % p), np.nanmedian(col), np.nanmin(col), np.nanmax(col)))
            # prologue + T x step + tail for the recurrences
            if k in (0, 1, 3, 4) and not name.startswith(("projection", "weight", "image", "latent")):
                g = lambda p: np.nanmedian(blk[:, p] - first[k])
                if k in (0, 1):
                    t_a = 1 if k == 0 else 2
                    n_between = (T - 1) - t_a
                    per = (g(6) - g(5)) / n_between if n_between > 0 else float("nan")
                    loop = g(7) - g(2)
                    print("     => prologue %.2f | time loop %.2f (in-situ %.3f us per step x %d = %.2f) | tail %.2f  (exit at %.2f)"
                          % (g(2), loop, per, T, per * T, g(15) - g(7), g(15)))
                else:
                    n_between = (T - 2) - 1
                    per = (g(6) - g(5)) / n_between if n_between > 0 else float("nan")
                    loop = g(7) - g(2)
                    print("     => prologue %.2f | time loop %.2f (in-situ %.3f us per step x %d = %.2f) | tail %.2f  (exit at %.2f)"
                          % (g(2), loop, per, T, per * T, g(15) - g(7), g(15)))


if __name__ == "__main__":
    main()
