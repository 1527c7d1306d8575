"""MFM_KL and MFM (the classes train_mfm instantiates, reference mfm_mosi.py:398-401; mfm_model.py:662-764, 469-555,
with the Memory Fusion Network encoder :93-199) on the FUSED plan: one C call per forward / step, no autograd, no
torch ops -- against the reference's golden outputs (kl_b32_t20, mmd_b32_t20), the CPU oracle's gradients for every
tensor of the state_dict, and the reference's own Adam trajectories.  Tolerance 1e-4 relative fp32."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle as O
from factorized_amd import configs, synth
from tests import cases
from tests.cases import grad_err, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _engine(cfgs, variant, **kw):
    from factorized_amd import engine
    e = engine.MFMEngine(cfgs, variant=variant, **kw)
    w = synth.make_weights(e.layout.shapes, seed=1234)
    e.load_weights(w)
    return e, w


def _gauss_from_gold(gold, cfg):
    return torch.from_numpy(np.ascontiguousarray(gold["mmd_gauss"]))


def _oracle(variant, cfgs, w, gauss=None):
    m = O.build(variant, cfgs)
    O.load_numpy_weights(m, w)
    m.train()
    if gauss is not None:
        cfg = cfgs[0]
        sizes = [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]]
        m.mmd_gauss = list(torch.split(gauss, sizes, dim=1))
    return m


@pytest.fixture(params=["att-rows", "att-gemm", "heads-gemm"], autouse=True)
def att_path(request, monkeypatch):
    """Every test of this file runs on every form of the MFN attention block: the row-block forward launches (lin_rows_kernel,
    the fp32 default up to T*B = 5120, forced here for any row count: "att-rows"), the grouped GEMMs + row kernels
    ("att-gemm": MFM_LIN_ROWS=0).  (A one-launch-per-direction form with every intermediate in LDS was built in round 2,
    measured slower and removed in round 5: profiles/r02_mfn_att_fused.txt.)"""
    if request.param == "att-rows":
        monkeypatch.setenv("MFM_LIN_ROWS", "1")
        monkeypatch.setenv("MFM_LIN_ROWS_MAXROWS", "100000000")
    else:
        monkeypatch.setenv("MFM_LIN_ROWS", "0")
        monkeypatch.delenv("MFM_LIN_ROWS_MAXROWS", raising=False)
    # "heads-gemm": the heads on [h_T | mem_T] as grouped GEMMs (MFM_MFN_HEADS_FOLD=0) instead of inside the memory
    # recurrence launches (the fp32 default)
    monkeypatch.delenv("MFM_MFN_HEADS_FOLD", raising=False)
    if request.param == "heads-gemm":
        monkeypatch.setenv("MFM_MFN_HEADS_FOLD", "0")
    return request.param


@pytest.mark.parametrize("name", ["kl_b32_t20", "mmd_b32_t20"])
def test_fused_mfn_plan_matches_reference_golden_and_oracle(name):
    _need_gpu()
    cs = cases.load_case(name)
    cfg, gold, variant = cs["cfg"], cs["gold"], cs["variant"]
    e, w = _engine(cs["cfgs"], variant)
    assert list(e.layout.shapes) == list(gold["param_names"])
    gauss = _gauss_from_gold(gold, cfg) if variant == "mmd" else None
    if gauss is not None:
        e.gauss = gauss.cuda()
    x, y = torch.from_numpy(cs["x"]), torch.from_numpy(cs["y"])
    xd, yd = x.cuda(), y.cuda()
    out = e.forward(xd, yd, train=True)
    ld = e.loss_dict(out["losses"])
    for k in ("disc", "gen_l", "gen_a", "gen_v", "gen", "reg", "loss"):
        ref = float(gold["fwd_" + k])
        assert abs(ld[k] - ref) <= TOL * max(abs(ref), 1e-3), (k, ld[k], ref)
    assert rel_err(out["y_hat"].cpu().numpy(), gold["y_hat"]) < TOL
    assert rel_err(out["x_a_hat"].cpu().numpy(), gold["x_a_hat"]) < TOL
    xl = out["x_l_hat"].cpu().numpy()
    assert rel_err(xl[0], gold["x_l_hat_first"]) < TOL and rel_err(xl[-1], gold["x_l_hat_last"]) < TOL
    # ---- every gradient against the oracle, and the reference's own summaries
    torch.set_num_threads(4)
    m = _oracle(variant, cs["cfgs"], w, gauss)
    O.loss_terms(m, x, y, cfg)["loss"].backward()
    e.backward(xd, yd, stage=0)
    gv = e.grad_views()
    rows, worst = [], ("", 0.0)
    for n, p in m.named_parameters():
        g = gv[n].cpu().numpy()
        if p.grad is None:                       # mfn_encoder.out_fc1 / out_fc2: in the state_dict, never used
            assert np.all(g == 0.0), n
            rows.append(np.full(10, np.nan))
            continue
        rows.append(cases.summarize(g))
        err = grad_err(g, p.grad.numpy())
        if err > worst[1]:
            worst = (n, err)
    cases.report("mfn_plan_grad_%s" % name, worst[1])
    assert worst[1] < TOL, "worst gradient mismatch %s: %.3e" % worst
    gs, got = gold["grad_summary"], np.stack(rows)
    ok = ~np.isnan(gs[:, 0])
    scale = np.maximum(np.abs(gs[ok][:, :1]), 1e-6)
    assert np.max(np.abs(got[ok] - gs[ok]) / scale) < 5 * TOL


@pytest.mark.parametrize("name", ["kl_b32_t20", "mmd_b32_t20"])
def test_fused_mfn_plan_training_trajectory_matches_reference(name):
    _need_gpu()
    cs = cases.load_case(name)
    cfg, gold, variant = cs["cfg"], cs["gold"], cs["variant"]
    e, _ = _engine(cs["cfgs"], variant)
    if variant == "mmd":
        e.gauss = _gauss_from_gold(gold, cfg).cuda()
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    trace = []
    for s in range(cs["steps"]):
        ld = e.loss_dict(e.train_step(x, y, lr=1e-3))
        trace.append([ld["loss"], ld["disc"], ld["gen"], ld["reg"]])
        if s == 0:
            p1 = np.stack([cases.summarize(v.cpu().numpy()) for v in e.param_views().values()])
    trace, ref = np.array(trace), gold["trace"]
    terr = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("mfn_plan_trace_rel_%s" % name, terr)
    assert terr < 0.5 * TOL, (trace[-1], ref[-1])        # measured worst 1.3e-5 (MMD variant), profiles/r02_parity_worst.jsonl
    ok = ~np.isnan(gold["grad_summary"][:, 0])          # parameters that receive a gradient
    scale1 = np.maximum(np.abs(gold["param_after1"][:, :1]), 1e-3)
    e1 = float(np.max((np.abs(p1 - gold["param_after1"]) / scale1)[ok]))
    pl = np.stack([cases.summarize(v.cpu().numpy()) for v in e.param_views().values()])
    scale = np.maximum(np.abs(gold["param_after_last"][:, :1]), 1e-3)
    el = float(np.max((np.abs(pl - gold["param_after_last"]) / scale)[ok]))
    cases.report("mfn_plan_param_after1_rel_%s" % name, e1)
    cases.report("mfn_plan_param_after_last_rel_%s" % name, el)
    assert e1 < 0.25 * TOL and el < 0.25 * TOL, (e1, el)      # measured worst 6.7e-6 / 6.8e-6
    # the unused MFN output layers never move (torch.optim.Adam skips parameters without a gradient)
    pv = e.param_views()
    w0 = synth.make_weights(e.layout.shapes, seed=1234)
    for n in ("mfn_encoder.out_fc1.weight", "mfn_encoder.out_fc2.bias"):
        assert np.array_equal(pv[n].cpu().numpy(), w0[n]), n


ODD_MFN = dict(input_dims=[37, 3, 11], h_dims=[40, 12, 20], memsize=24, zl_size=20, za_size=12, zv_size=36, zy_size=24,
               fy_size=12, fl_size=28, fa_size=4, fv_size=20)


@pytest.mark.parametrize("panel", [False, True])
@pytest.mark.parametrize("variant,B,T,od", [("kl", 19, 9, 1), ("mmd", 19, 9, 1), ("kl", 300, 6, 7), ("mmd", 5, 1, 1)])
def test_fused_mfn_plan_odd_sizes_and_large_batch_vs_oracle(variant, B, T, od, panel, monkeypatch):
    """ragged sizes (nothing a multiple of 16), 7-output head, T = 1, and a batch that takes the staged latent kernels
    and multi-row recurrence tiles; forward losses, every gradient, 3 Adam steps"""
    _need_gpu()
    if panel:
        monkeypatch.setenv("MFM_PANEL_MINROWS", "1")      # six LSTMs' projections (24 column groups) on the row-panel GEMM
        monkeypatch.setenv("MFM_DW_ONEPASS_MINROWS", "1")     # and their weight gradients on the one-pass kernel
    else:
        monkeypatch.delenv("MFM_PANEL_MINROWS", raising=False)
        monkeypatch.delenv("MFM_DW_ONEPASS_MINROWS", raising=False)
    cfgs = configs.canonical_configs(dropout=False, output_dim=od, **ODD_MFN)
    cfgs[1]["shapes"], cfgs[2]["shapes"], cfgs[3]["shapes"], cfgs[4]["shapes"] = 36, 20, 28, 44
    cfg = cfgs[0]
    e, w = _engine(cfgs, variant)
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=23, output_dim=od)
    gl = cfg["zl_size"] + cfg["za_size"] + cfg["zv_size"] + cfg["zy_size"]
    gauss = torch.from_numpy(np.random.RandomState(5).normal(size=(B, gl)).astype(np.float32)) if variant == "mmd" else None
    if gauss is not None:
        e.gauss = gauss.cuda()
    torch.set_num_threads(4)
    m = _oracle(variant, cfgs, w, gauss)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    terms = O.loss_terms(m, x, y, cfg)
    terms["loss"].backward()
    xd, yd = x.cuda(), y.cuda()
    out = e.forward(xd, yd, train=True, want_xhat=False)
    ld = e.loss_dict(out["losses"])
    for k in ("disc", "gen", "reg", "loss"):
        ref = float(terms[k].detach())
        assert abs(ld[k] - ref) <= TOL * max(abs(ref), 1e-3), (k, ld[k], ref)
    e.backward(xd, yd, stage=0)
    gv = e.grad_views()
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        err = grad_err(gv[n].cpu().numpy(), p.grad.numpy())
        if err > worst[1]:
            worst = (n, err)
    assert worst[1] < TOL, "worst gradient mismatch %s: %.3e" % worst
    g0 = {n: p.grad.detach().clone().numpy() for n, p in m.named_parameters() if p.grad is not None}
    opt = torch.optim.Adam(m.parameters())
    for _ in range(3):
        opt.zero_grad()
        O.loss_terms(m, x, y, cfg)["loss"].backward()
        opt.step()
        e.train_step(xd, yd)
    pv = e.param_views()
    for n, p in m.named_parameters():
        d = np.abs(pv[n].cpu().numpy() - p.detach().numpy())
        if n in g0:
            sig = np.abs(g0[n]) > 1e-3 * np.abs(g0[n]).max()
            if sig.any():
                assert d[sig].max() < 2e-5, n
        assert d.max() < 1.01 * 3e-3, n


@pytest.mark.parametrize("B,T,force", [(32, 20, True), (33, 3, True), (270, 4, False)])
def test_mmd_gemm_form_matches_oracle(B, T, force, monkeypatch):
    """MMD of the non-KL MFM as Gram-matrix GEMMs (mmd.hip, the default from B = 48; forced onto smaller batches here):
    regulariser value, every gradient (the MMD gradient enters through the latent backward's seed record) against the
    oracle at canonical sizes; B = 33 / 270 leave ragged last rows in the B x B matrices."""
    _need_gpu()
    if force:
        monkeypatch.setenv("MFM_MMD_GEMM_MINB", "1")
    else:
        monkeypatch.delenv("MFM_MMD_GEMM_MINB", raising=False)
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    e, w = _engine(cfgs, "mmd")
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=29)
    gl = cfg["zl_size"] + cfg["za_size"] + cfg["zv_size"] + cfg["zy_size"]
    gauss = torch.from_numpy(np.random.RandomState(6).normal(size=(B, gl)).astype(np.float32))
    e.gauss = gauss.cuda()
    torch.set_num_threads(4)
    m = _oracle("mmd", cfgs, w, gauss)
    x, y = torch.from_numpy(xn), torch.from_numpy(yn)
    terms = O.loss_terms(m, x, y, cfg)
    terms["loss"].backward()
    xd, yd = x.cuda(), y.cuda()
    out = e.forward(xd, yd, train=True, want_xhat=False)
    ld = e.loss_dict(out["losses"])
    for k in ("disc", "gen", "reg", "loss"):
        ref = float(terms[k].detach())
        assert abs(ld[k] - ref) <= TOL * max(abs(ref), 1e-3), (k, ld[k], ref)
    e.backward(xd, yd, stage=0)
    gv = e.grad_views()
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        err = grad_err(gv[n].cpu().numpy(), p.grad.numpy())
        if err > worst[1]:
            worst = (n, err)
    cases.report("mmd_gemm_form_B%d" % B, worst[1])
    assert worst[1] < TOL, "worst gradient mismatch %s: %.3e" % worst


class _StepMask(torch.nn.Module):
    """nn.Dropout stand-in for the oracle's per-timestep MFN dropouts: multiplies call t by mask[t]."""

    def __init__(self, mask):
        super().__init__()
        self.mask, self.t = mask, 0

    def forward(self, x):
        out = x * self.mask[self.t]
        self.t += 1
        return out


def test_fused_mfn_plan_dropout_masks_and_gradients():
    """train mode: the relu/dropout masks the attention GEMMs' epilogues drew are read back (0 or 1/(1-p), dropped
    fraction = p among the units the relu lets through), injected into the oracle together with the latent stack's
    masks, and forward losses + every gradient must then agree -- forward and backward use the same masks."""
    _need_gpu()
    P = dict(zl_to_fl_dropout=0.2, za_to_fa_dropout=0.5, zv_to_fv_dropout=0.7, zy_to_fy_dropout=0.3, fy_to_y_dropout=0.4)
    cfgs = configs.canonical_configs(dropout=True, **P)
    cfgs[1]["drop"], cfgs[2]["drop"], cfgs[3]["drop"], cfgs[4]["drop"] = 0.5, 0.3, 0.0, 0.0   # gamma nets: mfn_mem tests
    cfg = cfgs[0]
    B, T = 48, 6
    e, w = _engine(cfgs, "kl")
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=5)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    out = e.forward(x, y, train=True, want_xhat=False)
    buf = {k: v.cpu().numpy().copy() for k, v in e.mfn_buffers(T, B).items()}
    rec, _, lay = e.latent_record(T, B)
    rec = rec.cpu().numpy().copy()
    for tag, p, wname, src in (("1", 0.5, "mfn_encoder.att1_fc1", "cstar"), ("2", 0.3, "mfn_encoder.att2_fc1", "attended")):
        mk, h = buf["m" + tag], buf["h" + tag]
        keep = 1.0 / (1.0 - p)
        assert np.all((mk == 0.0) | (np.abs(mk - keep) < 1e-6))
        pre = buf[src].astype(np.float64) @ w[wname + ".weight"].T.astype(np.float64) + w[wname + ".bias"]
        pos = pre > 1e-6
        frac = float((mk[pos] == 0.0).mean())
        sigma = np.sqrt(p * (1 - p) / pos.sum())
        assert abs(frac - p) < 5 * sigma, (tag, frac, p)
        assert np.all(mk[pre < -1e-6] == 0.0)
        assert np.max(np.abs(h - np.maximum(pre, 0.0) * mk)) < 1e-5 * max(1.0, np.abs(pre).max())
    e.backward(x, y, stage=0)
    m = _oracle("kl", cfgs, w)
    # the reference's dropout acts AFTER the relu: a unit the relu zeroed needs no mask; use scale / 0 from the kernel
    # where pre > 0 and the scale value elsewhere (multiplying an exact zero)
    def step_mask(mk, p, n):
        full = np.where(mk == 0.0, 0.0, 1.0 / (1.0 - p)).astype(np.float32)
        return torch.from_numpy(full.reshape(T, B, n))
    pre1 = buf["cstar"].astype(np.float64) @ w["mfn_encoder.att1_fc1.weight"].T + w["mfn_encoder.att1_fc1.bias"]
    pre2 = buf["attended"].astype(np.float64) @ w["mfn_encoder.att2_fc1.weight"].T + w["mfn_encoder.att2_fc1.bias"]
    mk1 = np.where(pre1 > 0, buf["m1"], 2.0).astype(np.float32)      # relu-dead units: any value
    mk2 = np.where(pre2 > 0, buf["m2"], 1.0 / 0.7).astype(np.float32)
    m.mfn_encoder.att1_dropout = _StepMask(torch.from_numpy(mk1.reshape(T, B, -1)))
    m.mfn_encoder.att2_dropout = _StepMask(torch.from_numpy(mk2.reshape(T, B, -1)))
    sites = {"zl_to_fl": "zl_to_fl_dropout", "za_to_fa": "za_to_fa_dropout", "zv_to_fv": "zv_to_fv_dropout",
             "zy_to_fy": "zy_to_fy_dropout", "fy_to_y": "fy_to_y_dropout"}
    class _Fixed(torch.nn.Module):
        def __init__(self, mask):
            super().__init__()
            self.mask = mask
        def forward(self, v):
            return v * self.mask
    for site, key in sites.items():
        o_, n_ = lay["mask"][site], lay["width"][site]
        setattr(m, key, _Fixed(torch.from_numpy(rec[:, o_:o_ + n_].copy())))
    torch.set_num_threads(4)
    terms = O.loss_terms(m, torch.from_numpy(xn), torch.from_numpy(yn), cfg)
    terms["loss"].backward()
    ld = e.loss_dict(out["losses"])
    for k in ("disc", "gen", "reg", "loss"):
        ref = float(terms[k].detach())
        assert abs(ld[k] - ref) <= TOL * max(abs(ref), 1e-3), (k, ld[k], ref)
    gv = e.grad_views()
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert grad_err(gv[n].cpu().numpy(), p.grad.numpy()) < TOL, n
    # fresh masks on the next call
    e.forward(x, y, train=True, want_xhat=False)
    assert not np.array_equal(e.mfn_buffers(T, B)["m1"].cpu().numpy(), buf["m1"])
