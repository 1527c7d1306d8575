"""GPU parity of the fused MFM_KL_EF plan (forward, joint loss, backward, Adam) against
(a) the golden fixtures produced by the reference itself and (b) the CPU oracle run on the same
seeded inputs.  Tolerance: 1e-4 relative fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle as O
from factorized_amd import synth
from tests import cases
from tests.cases import grad_err, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _engine(cs):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    e = engine.MFMEngine(cs["cfgs"])
    w = synth.make_weights(e.layout.shapes, seed=1234)
    e.load_weights(w)
    return e, w


def _oracle(cs, w):
    torch.set_num_threads(4)
    m = O.build("kl_ef", cs["cfgs"])
    O.load_numpy_weights(m, w)
    m.train()
    return m


@pytest.mark.parametrize("name", cases.KLEF_CASES)
def test_forward_matches_reference_golden(name):
    cs = cases.load_case(name)
    e, _ = _engine(cs)
    gold = cs["gold"]
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    out = e.forward(x, y, train=True)          # dropout p=0 in the parity configs
    ld = e.loss_dict(out["losses"])
    for k in ("disc", "gen_l", "gen_a", "gen_v", "gen", "reg", "loss"):
        ref = float(gold["fwd_" + k])
        assert abs(ld[k] - ref) <= TOL * max(abs(ref), 1e-3), (k, ld[k], ref)
    assert rel_err(out["y_hat"].cpu().numpy(), gold["y_hat"]) < TOL
    assert rel_err(out["x_a_hat"].cpu().numpy(), gold["x_a_hat"]) < TOL
    xl, xv = out["x_l_hat"].cpu().numpy(), out["x_v_hat"].cpu().numpy()
    assert rel_err(xl[0], gold["x_l_hat_first"]) < TOL and rel_err(xl[-1], gold["x_l_hat_last"]) < TOL
    assert rel_err(xv[0], gold["x_v_hat_first"]) < TOL and rel_err(xv[-1], gold["x_v_hat_last"]) < TOL
    assert np.allclose(cases.summarize(xl)[:2], gold["x_l_hat_sum"][:2], rtol=TOL, atol=1e-4)


@pytest.fixture(params=["mfma", "small", "small:1", "small:2", "small:4", "stepwise"])
def seq_path(request, monkeypatch):
    path, _, rows = request.param.partition(":")
    if path == "stepwise":                            # recurrent GEMM + cell kernel per step (h > 128 path)
        monkeypatch.setenv("MFM_SEQ_STEPWISE", "1")
        path = "small"
    else:
        monkeypatch.delenv("MFM_SEQ_STEPWISE", raising=False)
    monkeypatch.setenv("MFM_SEQ_PATH", path)
    if rows:
        monkeypatch.setenv("MFM_SEQ_ROWS", rows)      # force the row-tile size of the VALU kernels
    else:
        monkeypatch.delenv("MFM_SEQ_ROWS", raising=False)
    return request.param


@pytest.mark.parametrize("name", cases.KLEF_CASES)
def test_gradients_match_oracle_and_golden(name, seq_path):
    cs = cases.load_case(name)
    e, w = _engine(cs)
    cfg = cs["cfg"]
    x, y = torch.from_numpy(cs["x"]), torch.from_numpy(cs["y"])
    m = _oracle(cs, w)
    terms = O.loss_terms(m, x, y, cfg, cs["loss_kind"])
    terms["loss"].backward()
    xd, yd = x.cuda(), y.cuda()
    e.forward(xd, yd, train=True, want_xhat=False)
    e.backward(xd, yd, stage=0)
    gv = e.grad_views()
    worst = ("", 0.0)
    rows = []
    for n, p in m.named_parameters():
        g = gv[n].cpu().numpy()
        err = grad_err(g, p.grad.numpy())
        rows.append(cases.summarize(g))
        if err > worst[1]:
            worst = (n, err)
    assert worst[1] < TOL, "worst gradient mismatch %s: %.3e" % worst
    gold = cs["gold"]["grad_summary"]
    got = np.stack(rows)
    scale = np.maximum(np.abs(gold[:, :1]), 1e-6)           # per-tensor L2 norm
    assert np.max(np.abs(got - gold) / scale) < 5 * TOL      # norm / sum / first-8 vs the reference


def test_wide_hidden_sizes_match_oracle():
    """Sizes from the reference's hyper-parameter search (mfm_mosi.py:1305-1320: zl, fl up to 256) make
    encoder_l 156, ef_encoder 244 and decoder_l 272 wide: beyond the weight-resident LSTM kernels (128) and
    the latent row kernels; forward losses, every gradient and a 3-step Adam trajectory against the oracle."""
    from factorized_amd import configs, engine
    cfgs = configs.canonical_configs(dropout=False, zl_size=156, fl_size=256, za_size=16, fa_size=32)
    cfg = cfgs[0]
    B, T = 9, 6
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=21)
    e = engine.MFMEngine(cfgs)
    w = synth.make_weights(e.layout.shapes, seed=77)
    e.load_weights(w)
    m = O.build("kl_ef", cfgs)
    O.load_numpy_weights(m, w)
    m.train()
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
        err = grad_err(gv[n].cpu().numpy(), p.grad.numpy())
        if err > worst[1]:
            worst = (n, err)
    assert worst[1] < TOL, "worst gradient mismatch %s: %.3e" % worst
    opt = torch.optim.Adam(m.parameters())
    for _ in range(3):
        opt.zero_grad()
        O.loss_terms(m, x, y, cfg)["loss"].backward()
        opt.step()
        e.train_step(xd, yd)
    pv = e.param_views()
    for n, p in m.named_parameters():
        # Adam normalises by sqrt(v): an element whose gradient is ~0 moves by up to lr per step whatever the
        # rounding says, so the trajectory bound is a fraction of the 3*lr the parameters can move at all
        assert np.max(np.abs(pv[n].cpu().numpy() - p.detach().numpy())) < 0.1 * 3e-3, n


def test_grad_step_equals_forward_plus_backward():
    """mfm_plan_grad_step (the data-parallel step's one-enqueue fwd+bwd, gradient buffer cleared inside the
    first launch) must leave exactly what forward + backward leave, also on a dirty gradient buffer."""
    cs = cases.load_case("klef_b32_t20")
    e, _ = _engine(cs)
    xd, yd = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    out = e.forward(xd, yd, train=True, want_xhat=False)
    e.backward(xd, yd, stage=0)
    g_ref = e.grads.clone()
    l_ref = out["losses"].clone()
    e.grads.fill_(3.0)                      # stale contents must not leak into the result
    l = e.grad_step(xd, yd)
    torch.cuda.synchronize()
    assert rel_err(e.grads.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6   # atomics reorder sums: not bitwise
    assert rel_err(l.cpu().numpy()[:5], l_ref.cpu().numpy()[:5]) < 1e-6
    # and a standalone backward afterwards still clears the buffer itself
    e.grads.fill_(5.0)
    e.forward(xd, yd, train=True, want_xhat=False)
    e.backward(xd, yd, stage=0)
    assert rel_err(e.grads.cpu().numpy(), g_ref.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("stage", [1, 2])
def test_staged_losses(stage):
    """train_beta_vae's stage 1 (gen+reg) and stage 2 (disc+reg) gradients (mfm_mosi.py:278-281)."""
    cs = cases.load_case("klef_b33_t7")
    e, w = _engine(cs)
    cfg = cs["cfg"]
    x, y = torch.from_numpy(cs["x"]), torch.from_numpy(cs["y"])
    m = _oracle(cs, w)
    terms = O.loss_terms(m, x, y, cfg, cs["loss_kind"])
    O.stage_loss(terms, cfg, stage).backward()
    xd, yd = x.cuda(), y.cuda()
    e.forward(xd, yd, train=True, want_xhat=False)
    e.backward(xd, yd, stage=stage)
    gv = e.grad_views()
    for n, p in m.named_parameters():
        g = gv[n].cpu().numpy()
        if p.grad is None:
            assert np.all(g == 0.0), n      # torch leaves .grad None; the flat buffer holds zeros
        else:
            assert grad_err(g, p.grad.numpy()) < TOL, n


@pytest.mark.parametrize("name", ["klef_b32_t20", "klef_b33_t7", "klef_odd_b19_t9", "klef_you_b32_t50"])
def test_training_trajectory_matches_reference(name):
    """N fused steps (fwd+bwd+Adam, one C call each) vs the reference's own loss trace and final
    parameters (golden), i.e. 'matched loss curves'."""
    cs = cases.load_case(name)
    e, _ = _engine(cs)
    gold = cs["gold"]
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    trace = []
    for s in range(cs["steps"]):
        losses = e.train_step(x, y, lr=1e-3)
        ld = e.loss_dict(losses)
        trace.append([ld["loss"], ld["disc"], ld["gen"], ld["reg"]])
        if s == 0:
            p1 = np.stack([cases.summarize(v.cpu().numpy()) for v in e.param_views().values()])
    trace = np.array(trace)
    ref = gold["trace"]
    cases.report("klef_trace_rel_%s" % name, np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    # measured worst over the four cases and all runs of round 2: 7.2e-7 (profiles/r02_parity_worst.jsonl).  Bounds: >= 10x
    # the measured worst cases (the weight gradients are summed with atomics in a run-dependent order and Adam amplifies
    # the last-bit differences over the steps: a 3x margin would flake on another clock state / CU count)
    assert np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)) < 0.1 * TOL, (trace[-1], ref[-1])
    assert np.max(np.abs(trace[0] - ref[0]) / np.maximum(np.abs(ref[0]), 1e-2)) < TOL
    scale1 = np.maximum(np.abs(gold["param_after1"][:, :1]), 1e-3)
    assert np.max(np.abs(p1 - gold["param_after1"]) / scale1) < 0.5 * TOL        # measured worst 3.2e-6
    pl = np.stack([cases.summarize(v.cpu().numpy()) for v in e.param_views().values()])
    scale = np.maximum(np.abs(gold["param_after_last"][:, :1]), 1e-3)
    cases.report("klef_param_after1_rel_%s" % name, np.max(np.abs(p1 - gold["param_after1"]) / scale1))
    cases.report("klef_param_after_last_rel_%s" % name, np.max(np.abs(pl - gold["param_after_last"]) / scale))
    assert np.max(np.abs(pl - gold["param_after_last"]) / scale) < 0.5 * TOL      # measured worst 3.3e-6


def test_full_size_properties():
    """Size-independent checks at a large batch (B=2048): linearity of the gradient in the loss
    weights and batch-permutation equivariance -- no CPU reference needed."""
    cs = cases.load_case("klef_b32_t20")
    e, _ = _engine(cs)
    B, T = 2048, 20
    xn, yn = synth.make_batch(cs["cfg"]["input_dims"], B, T, seed=3)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    out = e.forward(x, y, train=False)
    yh = out["y_hat"].clone()
    xa = out["x_a_hat"].clone()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).cuda()
    out2 = e.forward(x[:, perm].contiguous(), y[perm].contiguous(), train=False)
    assert torch.allclose(out2["y_hat"], yh[perm], rtol=1e-5, atol=1e-6)
    assert torch.allclose(out2["x_a_hat"], xa[:, perm], rtol=1e-5, atol=1e-6)
    l1, l2 = e.loss_dict(out["losses"]), e.loss_dict(out2["losses"])
    assert abs(l1["loss"] - l2["loss"]) < 1e-4 * abs(l1["loss"])
    # KLD is a batch SUM, the other terms batch MEANS
    half = e.forward(x[:, :B // 2].contiguous(), y[:B // 2].contiguous(), train=False)
    lh = e.loss_dict(half["losses"])
    other = e.forward(x[:, B // 2:].contiguous(), y[B // 2:].contiguous(), train=False)
    lo = e.loss_dict(other["losses"])
    assert abs((lh["reg"] + lo["reg"]) - l1["reg"]) < 1e-4 * abs(l1["reg"])
    assert abs(0.5 * (lh["disc"] + lo["disc"]) - l1["disc"]) < 1e-4 * abs(l1["disc"])
    assert abs(0.5 * (lh["gen"] + lo["gen"]) - l1["gen"]) < 1e-4 * abs(l1["gen"])


def test_eval_mode_is_deterministic_and_train_mode_is_not():
    """Dropout statistics, mask/scale values and forward-backward mask consistency: tests/test_gpu_dropout.py.
    Here only: eval mode is bit-reproducible, train mode (canonical dropouts) is not."""
    from factorized_amd import configs
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    cfgs = configs.canonical_configs(dropout=True)
    e = engine.MFMEngine(cfgs)
    e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
    xn, yn = synth.make_batch(cfgs[0]["input_dims"], 512, 20, seed=5)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    a = e.forward(x, y, train=False)["y_hat"].clone()
    b = e.forward(x, y, train=False)["y_hat"].clone()
    assert torch.equal(a, b)
    c = e.forward(x, y, train=True)["x_v_hat"].clone()
    d = e.forward(x, y, train=True)["x_v_hat"].clone()
    assert not torch.equal(c, d)            # different masks on successive calls
    assert torch.isfinite(c).all() and torch.isfinite(d).all()
