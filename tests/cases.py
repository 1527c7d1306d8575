"""Golden-case table shared by the CPU (oracle) and GPU (HIP) parity tests.
Mirrors CASES in tests/golden/make_golden.py."""
import os

import numpy as np

from factorized_amd import configs as C
from factorized_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ODD = dict(input_dims=[37, 3, 11], zl_size=20, za_size=12, zv_size=36, zy_size=24,
           fy_size=12, fl_size=28, fa_size=4, fv_size=20)

CASES = {
    "klef_b32_t20": ("kl_ef", C.canonical_configs, {}, 32, 20, 20),
    "klef_b1_t20": ("kl_ef", C.canonical_configs, {}, 1, 20, 3),
    "klef_b33_t7": ("kl_ef", C.canonical_configs, {}, 33, 7, 3),
    "klef_b5_t1": ("kl_ef", C.canonical_configs, {}, 5, 1, 3),
    "klef_b229_t20": ("kl_ef", C.canonical_configs, {}, 229, 20, 2),
    "klef_you_b32_t50": ("kl_ef", C.you_configs, {}, 32, 50, 3),
    "klef_mosei_b64_t20": ("kl_ef", C.mosei_configs, {}, 64, 20, 3),
    "klef_odd_b19_t9": ("kl_ef", C.canonical_configs, ODD, 19, 9, 3),
    "kl_b32_t20": ("kl", C.canonical_configs, {}, 32, 20, 5),
    "mmd_b32_t20": ("mmd", C.canonical_configs, {}, 32, 20, 5),
}
KLEF_CASES = [k for k, v in CASES.items() if v[0] == "kl_ef"]


def load_case(name):
    variant, fn, over, B, T, steps = CASES[name]
    cfgs = fn(dropout=False, **over)
    cfg = cfgs[0]
    loss_kind = cfg.get("loss", "l1")
    classes = cfg["output_dim"] if loss_kind == "ce" else 0
    x, y = synth.make_batch(cfg["input_dims"], B, T, seed=7, output_dim=cfg["output_dim"],
                            classes=classes)
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    return dict(name=name, variant=variant, cfgs=cfgs, cfg=cfg, B=B, T=T, steps=steps,
                loss_kind=loss_kind, x=x, y=y, gold=gold)


def summarize(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[:8], np.zeros(max(0, 8 - a.size))])


def grad_err(a, b, abs_slack=1e-7):
    """rel_err for parameter gradients: max|a-b| beyond `abs_slack`, over max|b|.  A gradient that is ~0 by
    construction (e.g. a bias behind cancelling L1 signs) carries only the rounding noise of the summation
    order (atomics: ~4e-9), which a purely relative measure cannot bound."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not b.size:
        return 0.0
    return float(max(np.abs(a - b).max() - abs_slack, 0.0) / max(np.abs(b).max(), 1e-12))


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'relative fp32 tolerance' used throughout
    (BASELINE.json north_star: 1e-4)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-12) if b.size else 1.0
    return float(np.abs(a - b).max() / den) if b.size else 0.0


def report(key, value):
    """Record a measured worst-case error (DESIGN.md quotes these): appended to gpurun_out/parity_worst.jsonl when
    that directory exists (the GPU runs create it), otherwise dropped."""
    import json
    d = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "gpurun_out")
    d = os.path.normpath(d)
    if os.path.isdir(d):
        try:
            with open(os.path.join(d, "parity_worst.jsonl"), "a") as f:
                f.write(json.dumps({"key": key, "value": float(value)}) + "\n")
        except OSError:
            pass
