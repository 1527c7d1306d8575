"""The reference's training loop UNCHANGED (train() of train_mfm, reference mfm_mosi.py:419-443, and the stage losses of
train_beta_vae, :255-285) with only the two import lines switched:

    from factorized_amd.mfm_model import MFM_KL_EF          # was: from mfm_model import MFM_KL_EF
    import factorized_amd.optim as optim                    # was: import torch.optim as optim

must follow the reference's own trajectories (goldens klef_b32_t20, klef_staged_b32_t20) -- with factorized_amd.optim.Adam
(one fused Adam launch over the model's flat buffers) and with the stock torch.optim.Adam alike."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from factorized_amd import configs, synth
from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(cfgs, fast=True, cls="MFM_KL_EF"):
    from factorized_amd import mfm_model as M
    model = getattr(M, cls)(*cfgs)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.make_weights(shapes, seed=1234)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    model.fast_grads = fast
    return model


def _reference_loop(model, optimizer, X, y, config, steps, stage_of=None, zero_kw=None):
    """mfm_mosi.py:424-442 (and :278-281 for the stage losses), statement by statement"""
    criterion = nn.L1Loss()
    gen_criterion = nn.MSELoss()
    d_l, d_a, d_v = config["input_dims"]
    model.train()
    trace = []
    for step in range(steps):
        optimizer.zero_grad(**(zero_kw or {}))
        batch_X = X
        batch_y = y
        decoded, mmd_loss, missing_loss = model.forward(batch_X)
        [x_l_hat, x_a_hat, x_v_hat, y_hat] = decoded
        batch_X_l = batch_X[:, :, :d_l]
        batch_X_a = batch_X[:, :, d_l:d_l + d_a]
        batch_X_v = batch_X[:, :, d_l + d_a:]
        gen_loss = config["lda_xl"] * gen_criterion(x_l_hat, batch_X_l) + config["lda_xa"] * gen_criterion(x_a_hat, batch_X_a) \
            + config["lda_xv"] * gen_criterion(x_v_hat, batch_X_v)
        disc_loss = criterion(y_hat.squeeze(1), batch_y)
        stage = stage_of(step) if stage_of else 0
        if stage == 1:
            loss = gen_loss + config["lda_mmd"] * mmd_loss
        elif stage == 2:
            loss = disc_loss + config["lda_mmd"] * mmd_loss
        else:
            loss = disc_loss + gen_loss + config["lda_mmd"] * mmd_loss + missing_loss
        loss.backward()
        optimizer.step()
        trace.append([loss.item(), disc_loss.item(), gen_loss.item(), mmd_loss.item()])
    return np.array(trace)


@pytest.mark.parametrize("which,fast", [("ours", True), ("torch", True), ("torch", False), ("ours", False)])
def test_unchanged_reference_loop_follows_reference_trajectory(which, fast):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import factorized_amd.optim as optim
    cs = cases.load_case("klef_b32_t20")
    cfg, gold = cs["cfg"], cs["gold"]
    model = _model(cs["cfgs"], fast)
    optimizer = (optim.Adam if which == "ours" else torch.optim.Adam)(model.parameters())      # :403, BEFORE .to(device) (:414)
    model = model.to("cuda")
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    trace = _reference_loop(model, optimizer, X, y, cfg, cs["steps"])
    ref = gold["trace"]
    terr = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("dropin_trace_rel_%s_%s" % (which, "fast" if fast else "pertensor"), terr)
    assert terr < 0.1 * TOL, (trace[-1], ref[-1])
    pl = np.stack([cases.summarize(p.detach().cpu().numpy()) for p in model.parameters()])
    scale = np.maximum(np.abs(gold["param_after_last"][:, :1]), 1e-3)
    perr = float(np.max(np.abs(pl - gold["param_after_last"]) / scale))
    cases.report("dropin_param_rel_%s_%s" % (which, "fast" if fast else "pertensor"), perr)
    assert perr < 0.5 * TOL, perr
    if which == "ours" and fast:
        # the fast path really ran: gradients are views of ONE buffer, nothing went through the stock optimizer
        assert model._grad_views_attached() and optimizer._fallback is None
        g = model._grad_flat
        assert all(p.grad.data_ptr() >= g.data_ptr() and p.grad.data_ptr() < g.data_ptr() + 4 * g.numel() for p in model.parameters())


def test_unchanged_reference_loop_mfm_kl_on_the_fused_plan():
    """train_mfm instantiates MFM_KL for config['type'] == 'kl' (reference mfm_mosi.py:398-399): the same unchanged loop, the
    model's forward is one call of the fused plan (variant "kl"), gradients land in the flat buffer, one fused Adam launch"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import factorized_amd.optim as optim
    cs = cases.load_case("kl_b32_t20")
    cfg, gold = cs["cfg"], cs["gold"]
    model = _model(cs["cfgs"], True, "MFM_KL")
    optimizer = optim.Adam(model.parameters())
    model = model.to("cuda")
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    trace = _reference_loop(model, optimizer, X, y, cfg, cs["steps"])
    ref = gold["trace"]
    terr = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("dropin_trace_rel_mfm_kl", terr)
    assert terr < 0.5 * TOL, (trace[-1], ref[-1])
    assert model._grad_views_attached() and optimizer._fallback is None
    # the unused MFN output layers never received a gradient and never moved (Adam skips them like torch does)
    w0 = synth.make_weights({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1234)
    assert np.array_equal(model.mfn_encoder.out_fc1.weight.detach().cpu().numpy(), w0["mfn_encoder.out_fc1.weight"])


def test_unchanged_reference_loop_mfm_mmd_on_the_fused_plan():
    """train_mfm instantiates MFM otherwise (reference mfm_mosi.py:400-401; the class BASELINE.json's north_star names): since
    round 4 its forward is one call of the fused plan too (variant "mmd") -- the forward leaves d MMD / d z unscaled, the backward
    weighs it with the upstream gradient the loop's `lda_mmd * mmd_loss` puts on the regulariser.  Reference trajectory with the
    reference's own N(0,1) samples injected (golden mmd_b32_t20)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import factorized_amd.optim as optim
    cs = cases.load_case("mmd_b32_t20")
    cfg, gold = cs["cfg"], cs["gold"]
    model = _model(cs["cfgs"], True, "MFM")
    optimizer = optim.Adam(model.parameters())
    model = model.to("cuda")
    g = torch.from_numpy(np.ascontiguousarray(gold["mmd_gauss"]))
    sizes = [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]]
    model.mmd_gauss = [t.cuda() for t in torch.split(g, sizes, dim=1)]
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    trace = _reference_loop(model, optimizer, X, y, cfg, cs["steps"])
    ref = gold["trace"]
    terr = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("dropin_trace_rel_mfm_mmd", terr)
    assert terr < 0.5 * TOL, (trace[-1], ref[-1])
    assert model._grad_views_attached() and optimizer._fallback is None
    pl = np.stack([cases.summarize(p.detach().cpu().numpy()) for p in model.parameters()])
    scale = np.maximum(np.abs(gold["param_after_last"][:, :1]), 1e-3)
    perr = float(np.max(np.abs(pl - gold["param_after_last"]) / scale))
    assert perr < 0.5 * TOL, perr
    # a different weight on the regulariser than the plan's own lda_mmd: gradients follow the loop's loss, not the config
    model.zero_grad()
    (xl, xa, xv, yh), reg, _ = model.forward(X)
    (3.0 * reg).backward()
    g3 = model._grad_flat.clone()
    model.zero_grad()
    (xl, xa, xv, yh), reg, _ = model.forward(X)
    reg.backward()
    g1 = model._grad_flat.clone()
    assert torch.allclose(g3, 3.0 * g1, rtol=1e-5, atol=1e-9) and float(g1.abs().max()) > 0


@pytest.mark.parametrize("mode", ["frozen", "legacy"])
def test_unchanged_staged_loop_with_dropin_adam(mode):
    """train_beta_vae's two stage losses through loss.backward(): tensors a stage loss does not reach get no gradient; the
    drop-in Adam then skips them (zero_grad(), torch >= 2 semantics: 'frozen') or keeps them moving on their decaying
    momentum (zero_grad(set_to_none=False), the reference's PyTorch 0.4: 'legacy') -- both pinned to the reference's traces"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import factorized_amd.optim as optim
    gold = np.load(cases.GOLDEN + "/klef_staged_b32_t20.npz")
    B, T, n1, n2 = (int(v) for v in gold["meta"])
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    model = _model(cfgs)
    optimizer = optim.Adam(model.parameters())
    model = model.to("cuda")
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    X, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    zero_kw = {} if mode == "frozen" else {"set_to_none": False}
    trace = _reference_loop(model, optimizer, X, y, cfg, n1 + n2, stage_of=lambda s: 1 if s < n1 else 2, zero_kw=zero_kw)
    ref = gold[mode + "_trace"]
    terr = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("dropin_staged_trace_rel_%s" % mode, terr)
    assert terr < 0.5 * TOL, (trace[-1], ref[-1])
    last = np.stack([cases.summarize(p.detach().cpu().numpy()) for p in model.parameters()])
    sc = np.maximum(np.abs(gold[mode + "_param_after_stage2"][:, :1]), 1e-3)
    perr = float(np.max(np.abs(last - gold[mode + "_param_after_stage2"]) / sc))
    cases.report("dropin_staged_param_rel_%s" % mode, perr)
    assert perr < 0.5 * TOL, perr


def test_flat_gradients_accumulate_and_match_per_tensor_path():
    """two backward passes without zero_grad add up (autograd semantics); the flat path's gradients equal the per-tensor
    path's; zero_grad() of the drop-in optimizer clears them in place; set p.grad = None by hand and the next backward
    re-attaches the views"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import factorized_amd.optim as optim
    cs = cases.load_case("klef_b33_t7")
    cfg = cs["cfg"]
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()

    def loss_of(model):
        decoded, reg, miss = model.forward(X)
        return (decoded[3].squeeze(1) - y).abs().mean() + sum(((d - X[:, :, a:b]) ** 2).mean() for d, (a, b) in
                                                              zip(decoded[:3], ((0, 300), (300, 305), (305, 325)))) + reg
    fast, slow = _model(cs["cfgs"], True).cuda(), _model(cs["cfgs"], False).cuda()
    fast.train(); slow.train()
    for m in (fast, slow):
        loss_of(m).backward()
        loss_of(m).backward()
    for (n, p), q in zip(fast.named_parameters(), slow.parameters()):
        assert cases.grad_err(p.grad.cpu().numpy(), q.grad.cpu().numpy()) < 1e-5, n
    single = _model(cs["cfgs"], True).cuda()
    single.train()
    loss_of(single).backward()
    for (n, p), q in zip(fast.named_parameters(), single.parameters()):
        assert cases.grad_err(p.grad.cpu().numpy(), 2.0 * q.grad.cpu().numpy()) < 1e-5, n
    opt = optim.Adam(fast.parameters())
    # zero_grad(): every tensor "no gradient yet" (torch: .grad = None); on a lazily-training model it costs no launch -- the
    # next forward's first launch clears the buffer, every backward behind it overwrites.  set_to_none=False: zeros in place
    opt.zero_grad()
    assert fast._grad_views_attached() and not fast._grad_present.any() and fast._grad_fresh
    opt.zero_grad(set_to_none=False)
    assert fast._grad_views_attached() and float(fast._grad_flat.abs().max()) == 0.0
    opt.zero_grad()
    for p in fast.parameters():
        p.grad = None
    loss_of(fast).backward()
    assert fast._grad_views_attached() and fast._grad_present.all()
    for (n, p), q in zip(fast.named_parameters(), single.parameters()):
        assert cases.grad_err(p.grad.cpu().numpy(), q.grad.cpu().numpy()) < 1e-5, n
