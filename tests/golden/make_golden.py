#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference (pliang279/factorized) has no tests or golden vectors, so parity is
pinned on its own outputs: this script imports /root/reference/mfm_model.py
read-only (no bytecode written, `.cuda()` turned into a no-op because the file
hard-codes it: mfm_model.py:29,51-52,76-77,147-153), loads the deterministic
weights/inputs of factorized_amd/synth.py, and runs forward / joint loss
(mfm_mosi.py:430-439 restated below because the Python-2 driver cannot be
imported) / backward / torch.optim.Adam.  Only numbers are written out -- inputs
are regenerated from seeds, outputs are stored as small summaries.  Nothing of the
reference's source travels.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn.functional as F

torch.Tensor.cuda = lambda self, *a, **k: self
import mfm_model as REF  # noqa: E402  (the reference)

from factorized_amd import configs as C  # noqa: E402
from factorized_amd import synth  # noqa: E402

torch.set_num_threads(1)  # deterministic summation order for the fixtures

CASES = [
    # name, variant, configs-fn, overrides, B, T, steps
    ("klef_b32_t20", "kl_ef", C.canonical_configs, {}, 32, 20, 20),
    ("klef_b1_t20", "kl_ef", C.canonical_configs, {}, 1, 20, 3),
    ("klef_b33_t7", "kl_ef", C.canonical_configs, {}, 33, 7, 3),
    ("klef_b5_t1", "kl_ef", C.canonical_configs, {}, 5, 1, 3),
    ("klef_b229_t20", "kl_ef", C.canonical_configs, {}, 229, 20, 2),
    ("klef_you_b32_t50", "kl_ef", C.you_configs, {}, 32, 50, 3),
    ("klef_mosei_b64_t20", "kl_ef", C.mosei_configs, {}, 64, 20, 3),
    ("klef_odd_b19_t9", "kl_ef", C.canonical_configs,
     dict(input_dims=[37, 3, 11], zl_size=20, za_size=12, zv_size=36, zy_size=24,
          fy_size=12, fl_size=28, fa_size=4, fv_size=20), 19, 9, 3),
    ("kl_b32_t20", "kl", C.canonical_configs, {}, 32, 20, 5),
    ("mmd_b32_t20", "mmd", C.canonical_configs, {}, 32, 20, 5),
]

REF_CLASS = {"kl_ef": REF.MFM_KL_EF, "kl": REF.MFM_KL, "mmd": REF.MFM}


def summarize(t):
    a = t.detach().cpu().numpy().astype(np.float64).ravel()
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[:8], np.zeros(max(0, 8 - a.size))])


def ref_losses(model, x, y, cfg, loss_kind):
    # restates mfm_mosi.py:430-439 / mfm_you.py:484 around the REFERENCE model
    d_l, d_a, _ = cfg["input_dims"]
    decoded, reg, missing = model.forward(x)
    x_l_hat, x_a_hat, x_v_hat, y_hat = decoded
    gl = F.mse_loss(x_l_hat, x[:, :, :d_l])
    ga = F.mse_loss(x_a_hat, x[:, :, d_l:d_l + d_a])
    gv = F.mse_loss(x_v_hat, x[:, :, d_l + d_a:])
    gen = cfg["lda_xl"] * gl + cfg["lda_xa"] * ga + cfg["lda_xv"] * gv
    if loss_kind == "ce":
        disc = F.cross_entropy(y_hat, y)
    elif y_hat.shape[1] == 1:
        disc = F.l1_loss(y_hat.squeeze(1), y)
    else:
        disc = F.l1_loss(y_hat, y)
    loss = disc + gen + cfg["lda_mmd"] * reg + missing
    return dict(disc=disc, gen=gen, gen_l=gl, gen_a=ga, gen_v=gv, reg=reg, loss=loss), decoded


def run_case(name, variant, cfg_fn, overrides, B, T, steps, light=False):
    cfgs = cfg_fn(dropout=False, **overrides)
    cfg = cfgs[0]
    loss_kind = cfg.get("loss", "l1")
    model = REF_CLASS[variant](*cfgs)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.make_weights(shapes, seed=1234)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    classes = cfg["output_dim"] if loss_kind == "ce" else 0
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7, output_dim=cfg["output_dim"],
                              classes=classes)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    model.train()
    out = {}
    gauss = None
    if variant == "mmd":
        rs = np.random.RandomState(99)
        sizes = [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]]
        gauss = [torch.from_numpy(rs.normal(size=(B, s)).astype(np.float32)) for s in sizes]
        out["mmd_gauss"] = np.concatenate([g.numpy() for g in gauss], axis=1)

    def forward_with_gauss():
        if gauss is None:
            return ref_losses(model, x, y, cfg, loss_kind)
        it = iter(gauss)
        orig = torch.randn
        torch.randn = lambda *a, **k: next(it)     # loss_MMD's sample (mfm_model.py:26)
        try:
            return ref_losses(model, x, y, cfg, loss_kind)
        finally:
            torch.randn = orig

    opt = torch.optim.Adam(model.parameters())     # mfm_mosi.py:403 (defaults)
    trace = []
    for s in range(steps):
        opt.zero_grad()
        terms, decoded = forward_with_gauss()
        terms["loss"].backward()
        if s == 0:
            for k in ("disc", "gen", "gen_l", "gen_a", "gen_v", "reg", "loss"):
                out["fwd_" + k] = np.float64(terms[k].item())
            x_l_hat, x_a_hat, x_v_hat, y_hat = decoded
            if not light:          # light cases (large batches) keep summaries only
                out["y_hat"] = y_hat.detach().numpy().copy()
                out["x_a_hat"] = x_a_hat.detach().numpy().copy()
            else:
                out["y_hat_sum"] = summarize(y_hat)
                out["x_a_hat_sum"] = summarize(x_a_hat)
            for tag, xh in (("x_l_hat", x_l_hat), ("x_v_hat", x_v_hat)):
                if not light:
                    out[tag + "_first"] = xh[0].detach().numpy().copy()
                    out[tag + "_last"] = xh[-1].detach().numpy().copy()
                out[tag + "_sum"] = summarize(xh)
            names = [n for n, _ in model.named_parameters()]
            out["grad_summary"] = np.stack([
                summarize(p.grad) if p.grad is not None else np.full(10, np.nan)
                for _, p in model.named_parameters()])
            out["param_names"] = np.array(names)
        trace.append([terms[k].item() for k in ("loss", "disc", "gen", "reg")])
        opt.step()
        if s == 0:
            out["param_after1"] = np.stack([summarize(p) for p in model.parameters()])
    out["param_after_last"] = np.stack([summarize(p) for p in model.parameters()])
    out["trace"] = np.array(trace, dtype=np.float64)
    out["meta"] = np.array([B, T, steps])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss0=%.6f" % trace[0][0], "lossN=%.6f" % trace[-1][0],
          "bytes=%d" % os.path.getsize(os.path.join(HERE, name + ".npz")))


def run_staged(name, B, T, n1, n2, variant="kl_ef"):
    """train_beta_vae's schedule (mfm_mosi.py:238-239, 278-284, 346-358): ONE Adam optimizer, `n1` steps on
    gen + reg (stage 1), then `n2` steps on disc + reg (stage 2), on the reference MFM_KL_EF (or, round 3, on
    MFM_KL / MFM: the stage masks then also cover the Memory Fusion Network's tensors; MFM's loss_MMD sample is a
    stored seeded sequence as in run_case).  Also stored: the gradient summaries of both stage losses at the initial weights
    (NaN rows = parameters the stage loss does not reach).  Two traces:
    'frozen' with this torch's zero_grad (sets .grad to None -> Adam skips parameters the stage loss does not
    reach) and 'legacy' with zero_grad(set_to_none=False), which is what the reference's PyTorch 0.4 did (a
    zero gradient keeps the parameter moving on its decaying first moment)."""
    cfgs = C.canonical_configs(dropout=False)
    cfg = cfgs[0]
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    out = {}
    gauss = None
    if variant == "mmd":
        rs = np.random.RandomState(99)
        sizes = [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]]
        gauss = [torch.from_numpy(rs.normal(size=(B, s)).astype(np.float32)) for s in sizes]
        out["mmd_gauss"] = np.concatenate([g.numpy() for g in gauss], axis=1)

    def losses_of(model):
        if gauss is None:
            return ref_losses(model, x, y, cfg, "l1")
        it = iter(gauss)
        orig = torch.randn
        torch.randn = lambda *a, **k: next(it)     # loss_MMD's sample (mfm_model.py:26)
        try:
            return ref_losses(model, x, y, cfg, "l1")
        finally:
            torch.randn = orig

    for mode in ("frozen", "legacy"):
        model = REF_CLASS[variant](*cfgs)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        w = synth.make_weights(shapes, seed=1234)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
        model.train()
        opt = torch.optim.Adam(model.parameters())
        if mode == "frozen":      # gradients of both stage losses at the INITIAL weights (NaN rows: .grad is None)
            for stage in (1, 2):
                opt.zero_grad(set_to_none=True)
                terms, _ = losses_of(model)
                reg = cfg["lda_mmd"] * terms["reg"]
                (terms["gen"] + reg if stage == 1 else terms["disc"] + reg).backward()
                out["grad_summary_stage%d" % stage] = np.stack([
                    summarize(p.grad) if p.grad is not None else np.full(10, np.nan)
                    for _, p in model.named_parameters()])
        trace = []
        for s in range(n1 + n2):
            stage = 1 if s < n1 else 2
            opt.zero_grad(set_to_none=(mode == "frozen"))
            terms, _ = losses_of(model)
            reg = cfg["lda_mmd"] * terms["reg"]
            loss = terms["gen"] + reg if stage == 1 else terms["disc"] + reg      # mfm_mosi.py:278-281
            loss.backward()
            opt.step()
            trace.append([loss.item(), terms["disc"].item(), terms["gen"].item(), terms["reg"].item()])
            if s == n1 - 1:
                out[mode + "_param_after_stage1"] = np.stack([summarize(p) for p in model.parameters()])
        out[mode + "_param_after_stage2"] = np.stack([summarize(p) for p in model.parameters()])
        out[mode + "_trace"] = np.array(trace, dtype=np.float64)
    out["param_names"] = np.array([n for n, _ in model.named_parameters()])
    out["meta"] = np.array([B, T, n1, n2])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "frozen lossN=%.6f legacy lossN=%.6f" % (out["frozen_trace"][-1][0], out["legacy_trace"][-1][0]),
          "bytes=%d" % os.path.getsize(os.path.join(HERE, name + ".npz")))


EXTRA_SIZES = dict(input_dims=[37, 3, 11], h_dims=[40, 12, 20], memsize=24, zl_size=20, za_size=12, zv_size=36, zy_size=24,
                   fy_size=12, fl_size=28, fa_size=4, fv_size=20)
EXTRA = ["M_A", "M_B", "M_C", "M_D", "MFM_missing", "seq2seq", "basic_missing"]


def run_extra(name, B=12, T=6, canonical=False):
    """(canonical=True, round 3: a second size per class -- the canonical MOSI dims at B=33, T=20 -> extra2_<name>.npz)
    ablations M_A..M_D (mfm_model.py:201-467) and the missing-modality family (:766-1017): the REFERENCE classes'
    forward outputs (summaries of every returned tensor, in order) and the gradient summaries of a scalar objective
    that touches every output (oracle/mfm_oracle_extra.py::test_objective).  loss_MMD's torch.randn samples are
    replaced by a seeded sequence that is stored next to the results."""
    from oracle import mfm_oracle_extra as X
    if canonical:
        cfgs = C.canonical_configs(dropout=False)
    else:
        cfgs = C.canonical_configs(dropout=False, **EXTRA_SIZES)
        cfgs[1]["shapes"], cfgs[2]["shapes"], cfgs[3]["shapes"], cfgs[4]["shapes"] = 36, 20, 28, 44
    cfg = cfgs[0]
    model = getattr(REF, name)(*cfgs)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.make_weights(shapes, seed=1234)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    model.train()
    xn, _ = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    x = torch.from_numpy(xn)
    rs = np.random.RandomState(99)
    zs = {"zl": cfg["zl_size"], "za": cfg["za_size"], "zv": cfg["zv_size"], "zy": cfg["zy_size"]}
    gauss = [torch.from_numpy(rs.normal(size=(B, zs[k])).astype(np.float32)) for k in X.N_GAUSS[name]]
    it = iter(gauss)
    orig = torch.randn
    torch.randn = lambda *a, **k: next(it)
    try:
        out = model.forward(x)
    finally:
        torch.randn = orig
    flat = X.flatten_outputs(out)
    X.test_objective(out).backward()
    res = {"out_summary": np.stack([summarize(o) for o in flat]),
           "out_shapes": np.array([list(o.shape) + [0] * (3 - o.dim()) for o in flat]),
           "objective": np.float64(X.test_objective(out).item()),
           "param_names": np.array([n for n, _ in model.named_parameters()]),
           "grad_summary": np.stack([summarize(p.grad) if p.grad is not None else np.full(10, np.nan)
                                     for _, p in model.named_parameters()]),
           "meta": np.array([B, T])}
    if gauss:
        res["gauss"] = np.concatenate([g.numpy() for g in gauss], axis=1)
    tag = "extra2_" if canonical else "extra_"
    np.savez_compressed(os.path.join(HERE, "%s%s.npz" % (tag, name)), **res)
    print(tag + name, "objective=%.6f" % res["objective"], "outputs=%d" % len(flat),
          "bytes=%d" % os.path.getsize(os.path.join(HERE, "%s%s.npz" % (tag, name))))


STAGED = [("klef_staged_b32_t20", 32, 20, 4, 4, "kl_ef"), ("kl_staged_b32_t20", 32, 20, 4, 4, "kl"),
          ("mmd_staged_b32_t20", 32, 20, 4, 4, "mmd")]
# BASELINE config 4 (MOSEI shape, large batch): summaries + loss trace only
LIGHT = [("klef_mosei_b1024_t20", "kl_ef", C.mosei_configs, {}, 1024, 20, 8),
         # SURVEY section 8d config 4 says T = 50: the same shape at the sequence length the config names (round 5)
         ("klef_mosei_b256_t50", "kl_ef", C.mosei_configs, {}, 256, 50, 4),
         # BASELINE config 3 (YouTube shape: cross-entropy head, D = 410, T = 50) at a batch where a bf16 plan is bf16-resident
         # (12,800 rows): the one combination round 5 timed but compared with nothing (round 6)
         ("klef_you_b256_t50", "kl_ef", C.you_configs, {}, 256, 50, 4)]


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for case in CASES:
        if only and case[0] not in only:
            continue
        run_case(*case)
    for case in STAGED:
        if only and case[0] not in only:
            continue
        run_staged(*case)
    for name in EXTRA:
        if only and ("extra_" + name) not in only:
            continue
        run_extra(name)
    for name in EXTRA:
        if only and ("extra2_" + name) not in only:
            continue
        run_extra(name, B=33, T=20, canonical=True)
    for case in LIGHT:
        if only and case[0] not in only:
            continue
        torch.set_num_threads(8)
        run_case(*case, light=True)
