"""The reference's UNCHANGED training step (train() of mfm_mosi.py:424-442: zero_grad, model.forward, torch losses, backward,
Adam) captured ONCE into a hipGraph and replayed (train.GraphedModuleStep) -- for the classes whose forward / backward is one
call of the fused plan (MFM_KL_EF, MFM_KL).  A captured launch freezes its kernel arguments, so everything that must differ
between replays lives in device words the graph itself advances (mfm_plan_state_layout): the dropout streams, the epochs of the
in-launch hand-overs (role workgroups at B <= 32), the optimizer's step count and learning rate (optim.Adam(capturable=True)).
Checked: the replayed loop follows the REFERENCE's own trajectory (golden klef_b32_t20 / kl_b32_t20), the role workgroups run
inside the graph, successive replays draw different masks, eager calls and replays can be mixed."""
import numpy as np
import pytest
import torch

from factorized_amd import configs, synth
from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(cfgs, cls="MFM_KL_EF"):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import mfm_model as M
    model = getattr(M, cls)(*cfgs)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.make_weights(shapes, seed=1234)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    return model.cuda().train()


@pytest.mark.parametrize("case,cls", [("klef_b32_t20", "MFM_KL_EF"), ("kl_b32_t20", "MFM_KL"), ("klef_b33_t7", "MFM_KL_EF")])
def test_graphed_reference_loop_follows_reference_trajectory(case, cls):
    from factorized_amd import train
    cs = cases.load_case(case)
    cfg, gold = cs["cfg"], cs["gold"]
    model = _model(cs["cfgs"], cls)
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    gs = train.GraphedModuleStep(model, cfg, cs["B"], cs["T"], lr=1e-3)
    assert gs.fused
    plan = model.engine.plan(cs["T"], cs["B"])
    if cs["B"] <= 32 and cls == "MFM_KL_EF":
        # the step inside the graph is the role-workgroup form (in-launch hand-overs with device-side epochs)
        assert plan.get_option("proj_roles_active") == 1 and plan.get_option("dw_roles_active") == 1
    trace = []
    for _ in range(cs["steps"]):
        loss, disc = gs.step(X, y)
        trace.append(float(loss))
    ref = gold["trace"][:, 0]
    terr = float(np.max(np.abs(np.array(trace) - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("graphed_loop_trace_rel_%s" % case, terr)
    assert terr < 0.1 * TOL, (trace[-1], ref[-1])
    pl = np.stack([cases.summarize(p.detach().cpu().numpy()) for p in model.parameters()])
    scale = np.maximum(np.abs(gold["param_after_last"][:, :1]), 1e-3)
    perr = float(np.max(np.abs(pl - gold["param_after_last"]) / scale))
    cases.report("graphed_loop_param_rel_%s" % case, perr)
    assert perr < 0.5 * TOL, perr
    assert model.engine.check_status() == 0
    # the graph advanced the plan's replay counters itself, once per replay
    torch.cuda.synchronize()
    assert int(plan.tick.item()) == cs["steps"]
    if plan.get_option("dw_roles_active"):
        assert int(plan.dw_tick.item()) == cs["steps"]


def test_replays_draw_new_dropout_masks_and_mix_with_eager_calls():
    """canonical dropouts on: the masks the latent kernels draw (read back from the plan's record) differ from replay to
    replay and from the eager calls in between; keep fractions are right; nothing waits for a stale hand-over flag."""
    from factorized_amd import train
    cfgs = configs.canonical_configs(dropout=True)
    cfg = cfgs[0]
    B, T = 32, 20
    model = _model(cfgs)
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    X, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    gs = train.GraphedModuleStep(model, cfg, B, T, lr=1e-4)
    eng = model.engine
    rec, _, lay = eng.latent_record(T, B)
    site = "zv_to_fv"                       # p = 0.7
    o, w = lay["mask"][site], lay["width"][site]
    masks, losses = [], []
    for i in range(6):
        if i in (2, 4):                     # an eager training step of the same plan between two replays
            eng.train_step(X, y, lr=1e-4)
            torch.cuda.synchronize()
            masks.append(rec[:, o:o + w].detach().cpu().numpy().copy())
        loss, _ = gs.step(X, y)
        torch.cuda.synchronize()
        losses.append(float(loss))
        masks.append(rec[:, o:o + w].detach().cpu().numpy().copy())
    assert np.isfinite(losses).all()
    for a in range(len(masks)):
        vals = np.unique(masks[a])
        assert all(v == 0.0 or abs(v - 1 / 0.3) < 1e-5 for v in vals.tolist()), vals
        for b in range(a + 1, len(masks)):
            assert not np.array_equal(masks[a], masks[b]), (a, b)
    keep = np.mean([np.mean(m > 0) for m in masks])
    assert abs(keep - 0.3) < 5 * np.sqrt(0.3 * 0.7 / (len(masks) * B * w))
    assert eng.check_status() == 0


def test_graphed_loop_learning_rate_is_a_device_word():
    from factorized_amd import train
    cs = cases.load_case("klef_b32_t20")
    model = _model(cs["cfgs"])
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    gs = train.GraphedModuleStep(model, cs["cfg"], cs["B"], cs["T"], lr=1e-3)
    gs.step(X, y)
    torch.cuda.synchronize()
    p0 = model.engine.params.detach().clone()
    gs.set_lr(0.0)
    gs.step(X, y)
    torch.cuda.synchronize()
    assert torch.equal(p0, model.engine.params)
    gs.set_lr(1e-3)
    gs.step(X, y)
    torch.cuda.synchronize()
    assert not torch.equal(p0, model.engine.params)


def test_capturable_optimizer_state_dict_round_trip():
    """optimizer.state_dict() carries the flat Adam state of a fused model (moments + step counts): a resumed run continues
    the trajectory instead of restarting the bias correction."""
    import factorized_amd.optim as optim
    cs = cases.load_case("klef_b32_t20")
    cfg = cs["cfg"]
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    d_l, d_a, d_v = cfg["input_dims"]
    l1, mse = torch.nn.L1Loss(), torch.nn.MSELoss()

    def step(model, opt):
        opt.zero_grad()
        (xl, xa, xv, yh), kld, miss = model.forward(X)
        loss = l1(yh.squeeze(1), y) + cfg["lda_xl"] * mse(xl, X[:, :, :d_l]) + cfg["lda_xa"] * mse(xa, X[:, :, d_l:d_l + d_a]) \
            + cfg["lda_xv"] * mse(xv, X[:, :, d_l + d_a:]) + cfg["lda_mmd"] * kld + miss
        loss.backward()
        opt.step()
        return float(loss.detach())

    a = _model(cs["cfgs"])
    oa = optim.Adam(a.parameters())
    for _ in range(5):
        step(a, oa)
    sd_m, sd_o = {k: v.clone() for k, v in a.state_dict().items()}, oa.state_dict()
    assert len(sd_o["fused"]) == 1 and sd_o["fused"][0]["steps"][0] == 5
    b = _model(cs["cfgs"])
    b.load_state_dict(sd_m)
    ob = optim.Adam(b.parameters())
    ob.load_state_dict(sd_o)
    la = [step(a, oa) for _ in range(5)]
    lb = [step(b, ob) for _ in range(5)]
    assert np.allclose(la, lb, rtol=1e-6), (la, lb)
    ref = cs["gold"]["trace"][5:10, 0]
    assert np.max(np.abs(np.array(lb) - ref) / np.abs(ref)) < 0.1 * TOL


def test_graphed_step_recaptures_after_a_failed_handover():
    """(advisor, round 4) plan options do not reach into a captured graph: after a hand-over time-out the replays would go on
    using the in-launch hand-overs under a sticky status -- every step skipped, for good.  GraphedModuleStep polls the plan's
    host-mapped status word before each replay; when it is raised it clears the status, switches the engine to separate
    launches, captures the step again and says so.  (A captured launch cannot take a fault injection -- that is a host-side
    kernel argument -- so the failure is planted where a real one leaves it: status word + host word.)"""
    from factorized_amd import train, _lib
    cs = cases.load_case("klef_b32_t20")
    model = _model(cs["cfgs"], "MFM_KL_EF")
    X, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    gs = train.GraphedModuleStep(model, cs["cfg"], cs["B"], cs["T"], lr=1e-3)
    plan = model.engine.plan(cs["T"], cs["B"])
    assert plan.get_option("dw_roles_active") == 1
    for _ in range(3):
        gs.step(X, y)
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in model.parameters()]
    plan.state.view(torch.int32)[_lib.MFM_LOSS_SLOTS] = 2          # what a weight-gradient consumer that gave up leaves behind
    torch.cuda.synchronize()
    plan.host_words[1] = 2
    # (without the poll: every replay from here on is skipped by the guarded Adam)
    with pytest.warns(RuntimeWarning, match="re-captured"):
        loss, disc = gs.step(X, y)
    assert gs.recaptures == 1 and not model.engine.handover and model.engine.handover_failures == 1
    for _ in range(3):
        loss, disc = gs.step(X, y)
    torch.cuda.synchronize()
    assert plan.get_option("dw_roles_active") == 0 and plan.get_option("proj_roles_active") == 0
    assert np.isfinite(float(loss)) and model.engine.check_status() == 0
    assert any(not torch.equal(a, p) for a, p in zip(before, model.parameters()))
