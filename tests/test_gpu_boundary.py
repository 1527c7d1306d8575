"""Drop-in boundary behaviour a reference driver depends on beyond forward/backward numerics:
whole-module checkpoints (`torch.save(model, path)` / `torch.load(path)`, reference mfm_mosi.py:342-346, 473-481),
copy.deepcopy, and loud failures where the fused plan cannot honour an autograd request."""
import copy
import io

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mfm_oracle as O
from factorized_amd import synth
from tests import cases
from tests.cases import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _ref_loss(model, x, y, cfg):
    d_l, d_a, _ = cfg["input_dims"]
    decoded, reg, missing = model.forward(x)
    x_l_hat, x_a_hat, x_v_hat, y_hat = decoded
    gen = cfg["lda_xl"] * F.mse_loss(x_l_hat, x[:, :, :d_l]) + cfg["lda_xa"] * F.mse_loss(x_a_hat, x[:, :, d_l:d_l + d_a]) \
        + cfg["lda_xv"] * F.mse_loss(x_v_hat, x[:, :, d_l + d_a:])
    return F.l1_loss(y_hat.squeeze(1), y) + gen + cfg["lda_mmd"] * reg + missing


def _roundtrip(model):
    buf = io.BytesIO()
    torch.save(model, buf)                      # the reference's checkpoint format: the whole module
    buf.seek(0)
    return torch.load(buf, weights_only=False)


def test_klef_whole_module_checkpoint_roundtrip():
    _need_gpu()
    from factorized_amd import mfm_model as M
    cs = cases.load_case("klef_b32_t20")
    cfg = cs["cfg"]
    ref = O.build("kl_ef", cs["cfgs"])
    model = M.MFM_KL_EF(*cs["cfgs"])
    model.load_state_dict(ref.state_dict())
    model = model.cuda()
    model.train()
    opt = torch.optim.Adam(model.parameters())
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    for _ in range(2):                          # the engine (native plan handles, workspaces) exists from here on
        opt.zero_grad()
        _ref_loss(model, x, y, cfg).backward()
        opt.step()
    assert model._engine is not None
    loaded = _roundtrip(model)
    assert list(loaded.state_dict().keys()) == list(ref.state_dict().keys())
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert torch.equal(a, b), k
    assert loaded._engine is None               # nothing native travelled; a fresh engine is adopted on first use
    model.eval(); loaded.eval()
    with torch.no_grad():
        d0, k0, _ = model.forward(x)
        d1, k1, _ = loaded.forward(x)
    for a, b in zip(d0, d1):
        assert torch.equal(a, b)
    # the KLD sum is accumulated with float atomics over the batch rows: equal up to the order of the additions
    assert abs(float(k0) - float(k1)) <= 1e-6 * abs(float(k0))
    # the restored module trains, and its fused engine still shares the module's storage
    loaded.train()
    before = loaded.fy_to_y_fc2.weight.detach().clone()
    loaded.engine.train_step(x, y)
    torch.cuda.synchronize()
    assert not torch.equal(before, loaded.fy_to_y_fc2.weight.detach())
    opt2 = torch.optim.Adam(loaded.parameters())
    opt2.zero_grad()
    _ref_loss(loaded, x, y, cfg).backward()
    opt2.step()
    # the original is untouched by what happened to the copy
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        if k == "fy_to_y_fc2.weight":
            assert not torch.equal(a, b)
    # deepcopy (best-model snapshots in user drivers) works the same way
    twin = copy.deepcopy(model)
    assert twin._engine is None
    twin.eval()
    with torch.no_grad():
        d2, _, _ = twin.forward(x)
    assert torch.equal(d2[3], d0[3])
    # state_dict checkpoints load into a fresh module as well
    fresh = M.MFM_KL_EF(*cs["cfgs"]).cuda()
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        d3, _, _ = fresh.forward(x)
    assert torch.equal(d3[3], d0[3])


@pytest.mark.parametrize("variant", ["kl", "mmd"])
def test_mfn_models_whole_module_checkpoint_after_graphed_steps(variant):
    _need_gpu()
    from factorized_amd import mfm_model as M
    from factorized_amd import train
    cs = cases.load_case("kl_b32_t20" if variant == "kl" else "mmd_b32_t20")
    cfg = cs["cfg"]
    model = (M.MFM_KL if variant == "kl" else M.MFM)(*cs["cfgs"]).cuda()
    model.train()
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    stepper = train.GraphedModuleStep(model, cfg, cs["B"], cs["T"])
    for _ in range(2):
        stepper.step(x, y)
    torch.cuda.synchronize()
    loaded = _roundtrip(model)
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert torch.equal(a, b), k
    model.eval(); loaded.eval()
    g = [torch.randn(cs["B"], n, generator=torch.Generator().manual_seed(i)).cuda()
         for i, n in enumerate((cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]))]
    model.mmd_gauss = loaded.mmd_gauss = g
    with torch.no_grad():
        d0, r0, _ = model.forward(x)
        d1, r1, _ = loaded.forward(x)
    for a, b in zip(d0, d1):
        assert torch.equal(a, b)
    assert torch.allclose(r0, r1, rtol=1e-6, atol=1e-7)


def test_klef_backward_refuses_stale_or_consumed_activations():
    """The plan keeps ONE set of activations per (T,B): a second forward before backward, or a second backward,
    must raise instead of returning gradients of the wrong graph."""
    _need_gpu()
    from factorized_amd import mfm_model as M
    from factorized_amd import _lib, train
    cs = cases.load_case("klef_b33_t7")
    cfg = cs["cfg"]
    model = M.MFM_KL_EF(*cs["cfgs"]).cuda()
    model.train()
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    loss_a = _ref_loss(model, x, y, cfg)
    model.eval()
    with torch.no_grad():
        model.forward(x)                        # e.g. an evaluation inside the step
    model.train()
    with pytest.raises(RuntimeError, match="another forward"):
        loss_a.backward()
    loss_b = _ref_loss(model, x, y, cfg)
    loss_b.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already back-propagated"):
        loss_b.backward()
    # a different (T,B) uses a different workspace and does not interfere
    loss_c = _ref_loss(model, x, y, cfg)
    with torch.no_grad():
        model.forward(x[:, :5].contiguous())
    loss_c.backward()
    # the input gradient is not produced by the fused plan: ask for it and it says so
    with pytest.raises(_lib.MfmError, match="requires grad"):
        model.forward(x.clone().requires_grad_(True))
    # (graph replay of the fused plan: tests/test_gpu_graph.py -- refused until round 4, when the dropout streams and the
    # hand-over epochs got device words that a captured step advances)


def test_mfn_model_input_gradient_matches_oracle():
    """MFM_KL on the grouped sequence launches: d loss / d x (ADVICE: it used to be silently None)."""
    _need_gpu()
    from factorized_amd import mfm_model as M
    cs = cases.load_case("kl_b32_t20")
    cfg = cs["cfg"]
    ref = O.build("kl", cs["cfgs"])
    w = synth.make_weights(O.state_shapes(ref), seed=1234)
    O.load_numpy_weights(ref, w)
    ref.train()
    model = M.MFM_KL(*cs["cfgs"])
    model.load_state_dict(ref.state_dict())
    model = model.cuda()
    model.train()
    x = torch.from_numpy(cs["x"][:8, :12].copy())
    y = torch.from_numpy(cs["y"][:12].copy())
    xr = x.clone().requires_grad_(True)
    xd = x.cuda().requires_grad_(True)
    _ref_loss(ref, xr, y, cfg).backward()
    _ref_loss(model, xd, y.cuda(), cfg).backward()
    assert xd.grad is not None
    err = rel_err(xd.grad.cpu().numpy(), xr.grad.numpy())
    cases.report("mfm_kl_dx_rel", err)
    assert err < TOL


def test_forked_child_does_not_tear_down_the_parents_plans():
    """A reference driver forks with live engines around (DataLoader workers, multiprocessing.Manager): the child inherits
    the Python objects but not a usable HIP runtime, and a garbage collection there must not run mfm_plan_destroy on the
    parent's handles (it used to end the child with a segmentation fault)."""
    import gc
    import os
    _need_gpu()
    from factorized_amd import configs, engine
    cfgs = configs.canonical_configs(dropout=False)
    e = engine.MFMEngine(cfgs)
    e.load_weights(synth.make_weights(e.layout.shapes))
    xn, yn = synth.make_batch(cfgs[0]["input_dims"], 8, 5)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    e.train_step(x, y)
    torch.cuda.synchronize()
    plan = e.plan(5, 8)
    assert plan.handle
    pid = os.fork()
    if pid == 0:                                   # child: drop every plan and collect; no HIP call may happen
        try:
            for p in list(getattr(e, "_plans", {}).values()):
                p.__del__()
            gc.collect()
        finally:
            os._exit(0)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    # the parent's plan is untouched
    e.train_step(x, y)
    torch.cuda.synchronize()
    assert e.check_status() == 0 and plan.handle
