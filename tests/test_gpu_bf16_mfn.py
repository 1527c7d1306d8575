"""bf16 plans of MFM_KL / MFM (round-2 review: `MFMEngine(cfgs, precision="bf16", variant="kl" | "mmd")` was a supported
plan with no numeric evidence).  On a bf16 plan the Memory Fusion Network's attention products run on the bf16-operand
grouped GEMM with their relu+dropout-mask / tanh / mask / accumulate epilogues, the heads on [h_T | mem_T] as accumulating
bf16 GEMM segments, and -- from B = 128, or always under MFM_BF16_SEQ_MINB=1 -- all six encoder recurrences on the bf16
MFMA kernels (the three MFN LSTMs with their per-step cell-state gradient input `dc_ext`).

Bounds are those of tests/test_gpu_bf16.py ("matched loss curve vs fp32, not 1e-4"): loss terms
within 1e-3 relative of the reference's fp32 golden, every gradient within 8e-2 relative L2 / cosine > 0.995 of the fp32
oracle, N-step loss traces within 2e-3 of the reference's own fp32 trace."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle as O
from factorized_amd import configs, synth
from tests import cases

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _engine(cfgs, variant):
    from factorized_amd import engine
    e = engine.MFMEngine(cfgs, variant=variant, precision="bf16")
    w = synth.make_weights(e.layout.shapes, seed=1234)
    e.load_weights(w)
    return e, w


def _oracle(variant, cfgs, w, gauss=None):
    m = O.build(variant, cfgs)
    O.load_numpy_weights(m, w)
    m.train()
    if gauss is not None:
        cfg = cfgs[0]
        m.mmd_gauss = list(torch.split(gauss, [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]], dim=1))
    return m


@pytest.fixture(params=["all-bf16", "default"])
def seq_policy(request, monkeypatch):
    """'all-bf16': the six encoder and three decoder recurrences on the bf16 MFMA kernels at every batch size
    (lstm_seq.hip::bf16_seq_pays); 'default': the plan's own selection"""
    if request.param == "all-bf16":
        monkeypatch.setenv("MFM_BF16_SEQ_MINB", "1")
        monkeypatch.setenv("MFM_BF16_STORE", "1")       # and the bf16-RESIDENT saved activations (default from T*B = 3840)
    else:
        monkeypatch.delenv("MFM_BF16_SEQ_MINB", raising=False)
        monkeypatch.delenv("MFM_BF16_STORE", raising=False)
    return request.param


def _grad_bounds(e, m, tag):
    gv = e.grad_views()
    worst_g, worst_c = ("", 0.0), ("", 1.0)
    for n, p in m.named_parameters():
        g = gv[n].cpu().numpy().astype(np.float64).ravel()
        if p.grad is None:
            assert np.all(g == 0.0), n
            continue
        r = p.grad.numpy().astype(np.float64).ravel()
        nr = np.linalg.norm(r)
        if nr < 1e-9:
            continue
        rel = np.linalg.norm(g - r) / nr
        cos = float(g @ r / (np.linalg.norm(g) * nr + 1e-300))
        if rel > worst_g[1]:
            worst_g = (n, rel)
        if cos < worst_c[1]:
            worst_c = (n, cos)
    cases.report("bf16_mfn_grad_relL2_%s" % tag, worst_g[1])
    cases.report("bf16_mfn_grad_one_minus_cos_%s" % tag, 1.0 - worst_c[1])
    assert worst_g[1] < 8e-2, worst_g
    assert worst_c[1] > 0.995, worst_c
    assert worst_g[1] > 1e-5          # not accidentally the fp32 path


@pytest.mark.parametrize("name", ["kl_b32_t20", "mmd_b32_t20"])
def test_bf16_mfn_forward_and_gradients_near_fp32_reference(name, seq_policy):
    _need_gpu()
    cs = cases.load_case(name)
    cfg, gold, variant = cs["cfg"], cs["gold"], cs["variant"]
    e, w = _engine(cs["cfgs"], variant)
    gauss = torch.from_numpy(np.ascontiguousarray(gold["mmd_gauss"])) if variant == "mmd" else None
    if gauss is not None:
        e.gauss = gauss.cuda()
    x, y = torch.from_numpy(cs["x"]), torch.from_numpy(cs["y"])
    xd, yd = x.cuda(), y.cuda()
    out = e.forward(xd, yd, train=True)
    ld = e.loss_dict(out["losses"])
    worst_l = 0.0
    for k in ("disc", "gen_l", "gen_a", "gen_v", "reg", "loss"):
        ref = float(gold["fwd_" + k])
        worst_l = max(worst_l, abs(ld[k] - ref) / max(abs(ref), 1e-3))
    cases.report("bf16_mfn_loss_terms_rel_%s_%s" % (name, seq_policy), worst_l)
    assert worst_l < 1e-3, (ld, worst_l)
    assert cases.rel_err(out["y_hat"].cpu().numpy(), gold["y_hat"]) < 5e-2
    assert cases.rel_err(out["x_a_hat"].cpu().numpy(), gold["x_a_hat"]) < 5e-2
    torch.set_num_threads(4)
    m = _oracle(variant, cs["cfgs"], w, gauss)
    O.loss_terms(m, x, y, cfg)["loss"].backward()
    e.backward(xd, yd, stage=0)
    _grad_bounds(e, m, "%s_%s" % (name, seq_policy))


@pytest.mark.parametrize("name", ["kl_b32_t20", "mmd_b32_t20"])
def test_bf16_mfn_loss_curve_tracks_fp32_reference(name, seq_policy):
    _need_gpu()
    cs = cases.load_case(name)
    e, _ = _engine(cs["cfgs"], cs["variant"])
    if cs["variant"] == "mmd":
        e.gauss = torch.from_numpy(np.ascontiguousarray(cs["gold"]["mmd_gauss"])).cuda()
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    trace = []
    for _ in range(cs["steps"]):
        ld = e.loss_dict(e.train_step(x, y, lr=1e-3))
        trace.append([ld["loss"], ld["disc"], ld["gen"], ld["reg"]])
    trace, ref = np.array(trace), cs["gold"]["trace"]
    dev_ = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("bf16_mfn_trace_rel_%s_%s" % (name, seq_policy), dev_)
    assert dev_ < 2e-3, (trace[-1], ref[-1])
    assert trace[-1, 0] < trace[0, 0]
    # the unused MFN output layers never move
    w0 = synth.make_weights(e.layout.shapes, seed=1234)
    assert np.array_equal(e.param_views()["mfn_encoder.out_fc1.weight"].cpu().numpy(), w0["mfn_encoder.out_fc1.weight"])


ODD_MFN = dict(input_dims=[37, 3, 11], h_dims=[40, 12, 20], memsize=24, zl_size=20, za_size=12, zv_size=36, zy_size=24,
               fy_size=12, fl_size=28, fa_size=4, fv_size=20)


@pytest.mark.parametrize("panel", [False, True])
@pytest.mark.parametrize("variant,B,T,od", [("kl", 19, 9, 1), ("mmd", 19, 9, 1), ("kl", 300, 6, 7), ("mmd", 5, 1, 1)])
def test_bf16_mfn_odd_sizes_and_large_batch_vs_oracle(variant, B, T, od, panel, seq_policy, monkeypatch):
    """ragged sizes, 7-output head, T = 1, and B = 300 (bf16 recurrences by default, staged latent kernels); optionally
    the row-panel projection GEMM and the one-pass weight-gradient kernel in their bf16 forms"""
    _need_gpu()
    if panel:
        monkeypatch.setenv("MFM_PANEL_MINROWS", "1")
        monkeypatch.setenv("MFM_DW_ONEPASS_MINROWS", "1")
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
        assert abs(ld[k] - ref) <= 2e-3 * max(abs(ref), 1e-2), (k, ld[k], ref)
    e.backward(xd, yd, stage=0)
    _grad_bounds(e, m, "odd_%s_B%d_%s_%s" % (variant, B, "panel" if panel else "tiled", seq_policy))


class _StepMask(torch.nn.Module):
    def __init__(self, mask):
        super().__init__()
        self.mask, self.t = mask, 0

    def forward(self, x):
        out = x * self.mask[self.t]
        self.t += 1
        return out


class _Fixed(torch.nn.Module):
    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, v):
        return v * self.mask


def _force_relu_pattern(lin, active, known):
    """forward hook on an oracle Linear that feeds a relu: make the relu take the PLAN's on/off decision wherever it is known
    (a unit whose pre-activation sits within bf16 rounding of zero flips between the bf16 plan and the fp32 oracle -- on an
    8 x 8 layer one flipped unit is a quarter of the gradient, and which units sit that close depends on the mask stream, i.e.
    on the seed).  On where the plan is on: the value moves by < 2 |pre| (rounding-sized), the gradient passes; off: neither."""
    active, known = torch.from_numpy(active), torch.from_numpy(known)

    def hook(mod, inp, out):
        on = known & active & (out <= 0)
        off = known & (~active) & (out > 0)
        out = out + on.float() * (2.0 * out.abs() + 1e-30).detach()
        return torch.where(off, -torch.ones_like(out), out)
    lin.register_forward_hook(hook)
    return lin


@pytest.mark.parametrize("seed", [3, 1234, 7, 11, 99, 2024, 31337, 5])
def test_bf16_mfn_dropout_epilogues(seed):
    """train mode on a bf16 plan: the relu + dropout masks drawn in the bf16 GEMM epilogues are 0 | 1/(1-p), keep the
    right fraction, are applied to the stored activation, and -- injected into the fp32 oracle together with the latent
    stack's masks AND its relu on/off pattern (both are in the plan's record) -- give losses / gradients within the bf16
    rounding bound for EVERY seed of the mask stream: forward and backward use the same masks."""
    _need_gpu()
    P = dict(zl_to_fl_dropout=0.2, za_to_fa_dropout=0.5, zv_to_fv_dropout=0.7, zy_to_fy_dropout=0.3, fy_to_y_dropout=0.4)
    cfgs = configs.canonical_configs(dropout=True, **P)
    cfgs[1]["drop"], cfgs[2]["drop"], cfgs[3]["drop"], cfgs[4]["drop"] = 0.5, 0.3, 0.0, 0.0
    cfg = cfgs[0]
    B, T = 48, 6
    e, w = _engine(cfgs, "kl")
    e.seed = seed
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=5)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    out = e.forward(x, y, train=True, want_xhat=False)
    buf = {k: v.cpu().numpy().copy() for k, v in e.mfn_buffers(T, B).items()}
    rec, _, lay = e.latent_record(T, B)
    rec = rec.cpu().numpy().copy()

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy().astype(np.float64)
    pres = {}
    for tag, p, wname, src in (("1", 0.5, "mfn_encoder.att1_fc1", "cstar"), ("2", 0.3, "mfn_encoder.att2_fc1", "attended")):
        mk, h = buf["m" + tag], buf["h" + tag]
        keep = 1.0 / (1.0 - p)
        assert np.all((mk == 0.0) | (np.abs(mk - keep) < 1e-6))
        pre = bf(buf[src]) @ bf(w[wname + ".weight"]).T + w[wname + ".bias"]        # what the bf16 GEMM computes
        pres[tag] = pre
        pos = pre > 1e-4
        frac = float((mk[pos] == 0.0).mean())
        sigma = np.sqrt(p * (1 - p) / pos.sum())
        assert abs(frac - p) < 5 * sigma, (tag, frac, p)
        assert np.all(mk[pre < -1e-4] == 0.0)
        assert np.max(np.abs(h - np.maximum(pre, 0.0) * mk)) < 2e-4 * max(1.0, np.abs(pre).max())
    e.backward(x, y, stage=0)
    m = _oracle("kl", cfgs, w)
    mk1 = np.where(pres["1"] > 0, buf["m1"], 2.0).astype(np.float32)
    mk2 = np.where(pres["2"] > 0, buf["m2"], 1.0 / 0.7).astype(np.float32)
    m.mfn_encoder.att1_dropout = _StepMask(torch.from_numpy(mk1.reshape(T, B, -1)))
    m.mfn_encoder.att2_dropout = _StepMask(torch.from_numpy(mk2.reshape(T, B, -1)))
    sites = {"zl_to_fl": "zl_to_fl_dropout", "za_to_fa": "za_to_fa_dropout", "zv_to_fv": "zv_to_fv_dropout",
             "zy_to_fy": "zy_to_fy_dropout", "fy_to_y": "fy_to_y_dropout"}
    for site, key in sites.items():
        o_, n_ = lay["mask"][site], lay["width"][site]
        setattr(m, key, _Fixed(torch.from_numpy(rec[:, o_:o_ + n_].copy())))
        # relu behind fc1: the plan's decision is visible where the unit was not dropped (post-dropout activation > 0 <=> on)
        a_ = lay["act"][site]
        _force_relu_pattern(getattr(m, site + "_fc1"), rec[:, a_:a_ + n_] > 0, rec[:, o_:o_ + n_] > 0)
    for site, ch in (("zl_to_fl", "l"), ("za_to_fa", "a"), ("zv_to_fv", "v"), ("zy_to_fy", "y")):
        f_, n_ = lay["f"][ch], lay["width"][site]            # relu behind fc2 (no dropout): known everywhere
        _force_relu_pattern(getattr(m, site + "_fc2"), rec[:, f_:f_ + n_] > 0, np.ones((B, n_), dtype=bool))
    torch.set_num_threads(4)
    terms = O.loss_terms(m, torch.from_numpy(xn), torch.from_numpy(yn), cfg)
    terms["loss"].backward()
    ld = e.loss_dict(out["losses"])
    for k in ("disc", "gen", "reg", "loss"):
        ref = float(terms[k].detach())
        assert abs(ld[k] - ref) <= 5e-3 * max(abs(ref), 1e-2), (k, ld[k], ref)
    # a relu unit whose pre-activation sits within bf16 rounding of zero can flip between the bf16 plan and the fp32
    # oracle: gradients are bounded in relative L2 like every other bf16 gradient, looser for that reason
    gv = e.grad_views()
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        g, r = gv[n].cpu().numpy().astype(np.float64).ravel(), p.grad.numpy().astype(np.float64).ravel()
        nr = np.linalg.norm(r)
        if nr < 1e-9:
            continue
        rel = np.linalg.norm(g - r) / nr
        if rel > worst[1]:
            worst = (n, rel)
    cases.report("bf16_mfn_dropout_grad_relL2_seed%d" % seed, worst[1])
    assert worst[1] < 0.08, worst
    e.forward(x, y, train=True, want_xhat=False)
    assert not np.array_equal(e.mfn_buffers(T, B)["m1"].cpu().numpy(), buf["m1"])
