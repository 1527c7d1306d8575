"""Shared set-up of the ablation / missing-modality parity cases (mirrors make_golden.py::run_extra)."""
import os

import numpy as np
import torch

from factorized_amd import configs as C
from factorized_amd import synth
from tests.cases import GOLDEN

EXTRA = ["M_A", "M_B", "M_C", "M_D", "MFM_missing", "seq2seq", "basic_missing"]
EXTRA_SIZES = dict(input_dims=[37, 3, 11], h_dims=[40, 12, 20], memsize=24, zl_size=20, za_size=12, zv_size=36, zy_size=24,
                   fy_size=12, fl_size=28, fa_size=4, fv_size=20)
GAUSS_KEYS = {"M_A": ["zl", "zy"], "M_B": ["zl", "za", "zv"], "M_C": ["zy"], "M_D": [], "MFM_missing": ["zl", "za", "zv", "zy"],
              "seq2seq": ["zv", "za", "zl"], "basic_missing": ["zy", "zy", "zy"]}


def extra_configs():
    cfgs = C.canonical_configs(dropout=False, **EXTRA_SIZES)
    cfgs[1]["shapes"], cfgs[2]["shapes"], cfgs[3]["shapes"], cfgs[4]["shapes"] = 36, 20, 28, 44
    return cfgs


SIZES = ["small", "canonical"]        # extra_<name>.npz (odd sizes, B=12, T=6) / extra2_<name>.npz (canonical MOSI dims, B=33, T=20)


def load_extra(name, size="small"):
    """-> (six configs, golden npz, x [T,B,D] float32, [gauss tensors in loss_MMD call order])"""
    cfgs = extra_configs() if size == "small" else C.canonical_configs(dropout=False)
    cfg = cfgs[0]
    gold = np.load(os.path.join(GOLDEN, "%s_%s.npz" % ("extra" if size == "small" else "extra2", name)))
    B, T = (int(v) for v in gold["meta"])
    x, _ = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    gauss = []
    if "gauss" in gold.files:
        zs = {"zl": cfg["zl_size"], "za": cfg["za_size"], "zv": cfg["zv_size"], "zy": cfg["zy_size"]}
        sizes = [zs[k] for k in GAUSS_KEYS[name]]
        gauss = list(torch.split(torch.from_numpy(np.ascontiguousarray(gold["gauss"])), sizes, dim=1))
    return cfgs, gold, x, gauss
