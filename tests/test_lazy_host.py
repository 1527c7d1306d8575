"""Host logic of the lazy losses (factorized_amd/lazy.py) on the CPU: a torch stand-in implements the step interface, the
reference's loss expression (mfm_mosi.py:433-439) is built on the lazy outputs, and values / gradients must equal the plain
torch evaluation -- on the symbolic path and on every fallback (materialisation) path."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from factorized_amd import lazy

D = (6, 2, 3)
LDA = (1.0, 0.01, 0.5)


class FakeStep(lazy.StepBase):
    """x_hat_m = a_m * x_m + b_m, y_hat = c * mean(x) [B,1], reg = sum of squares of the parameters: differentiable stand-ins
    for the plan (same interface, torch autograd underneath)"""

    def __init__(self, params, x, kind=0):
        self.p, self.x, self.x_version = params, x, x._version
        self.dims, self.lda, self.loss_kind = D, LDA, kind
        self.fast_calls = 0
        self._out = self._forward()
        self.views = tuple(t.detach().clone() for t in self._out[:4])
        self.scalar_view = torch.zeros(())
        self.slots = torch.zeros(8)
        lo = 0
        for m in range(3):
            self.slots[1 + m] = F.mse_loss(self.views[m], x[:, :, lo:lo + D[m]])
            lo += D[m]
        self.slots[4] = self._out[4].detach()
        self.grads = None
        self.live = True

    def _forward(self):
        p, x = self.p, self.x
        outs, lo = [], 0
        for m in range(3):
            outs.append(p["a"][m] * x[:, :, lo:lo + D[m]] + p["b"][m])
            lo += D[m]
        od = 1 if self.loss_kind == 0 else 3
        y = (x.mean(dim=(0, 2)).unsqueeze(1) * p["c"][:od].unsqueeze(0))
        reg = (p["a"] ** 2).sum() + (p["b"] ** 2).sum() + (p["c"] ** 2).sum()
        return outs + [y, reg]

    def check_live(self, what):
        if not self.live:
            raise RuntimeError("stale")

    def realize(self):
        if self.real is None:
            self.real = tuple(self._forward())
        return self.real

    def backward_weighted(self, coef, labels, terms):
        self.fast_calls += 1
        outs = self._forward()
        loss, lo = 0.0, 0
        for m in range(3):
            if coef.get(1 + m, 0.0) != 0.0:
                assert abs(coef[1 + m] - LDA[m]) < 1e-9
                loss = loss + LDA[m] * F.mse_loss(outs[m], self.x[:, :, lo:lo + D[m]])
            lo += D[m]
        if labels is not None:
            d = F.l1_loss(outs[3], labels.reshape(outs[3].shape)) if self.loss_kind == 0 else F.cross_entropy(outs[3], labels)
            self.slots[0] = d.detach()
            loss = loss + coef.get(0, 0.0) * d
        loss = loss + coef.get(4, 0.0) * outs[4]
        g = torch.autograd.grad(loss, list(self.p.values()), allow_unused=True)
        for p, gi in zip(self.p.values(), g):
            gi = torch.zeros_like(p) if gi is None else gi
            p.grad = gi if p.grad is None else p.grad + gi

    def host_slots(self):
        return [float(v) for v in self.slots[:5]]

    def device_slots(self):
        return self.slots


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(3, generator=g).requires_grad_() for k in ("a", "b", "c")}


def _batch(kind=0, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(4, 5, sum(D), generator=g)
    y = torch.randn(5, generator=g) if kind == 0 else torch.randint(0, 3, (5,), generator=g)
    return x, y


def _lazy_outputs(step):
    v = step.views
    return [lazy.LazyOut(v[i], step, i) for i in range(4)], lazy.LossExpr(step, {lazy.REG: 1.0}), 0.0


def _reference_expression(decoded, reg, missing, x, y, kind=0, lda=LDA, lda_mmd=0.3, mse=None, disc_fn=None):
    """mfm_mosi.py:430-439 statement by statement"""
    mse = mse or nn.MSELoss()
    disc_fn = disc_fn or (nn.L1Loss() if kind == 0 else nn.CrossEntropyLoss())
    x_l_hat, x_a_hat, x_v_hat, y_hat = decoded
    y_hat = y_hat.squeeze(1)
    mmd_loss = lda_mmd * reg
    x_l, x_a, x_v = x[:, :, :D[0]], x[:, :, D[0]:D[0] + D[1]], x[:, :, D[0] + D[1]:]
    gen_loss = lda[0] * mse(x_l_hat, x_l) + lda[1] * mse(x_a_hat, x_a) + lda[2] * mse(x_v_hat, x_v)
    disc_loss = disc_fn(y_hat, y)
    loss = disc_loss + gen_loss + mmd_loss + missing
    return loss, disc_loss, gen_loss, mmd_loss


def _plain(kind=0, **kw):
    p = _params()
    x, y = _batch(kind)
    st = FakeStep(p, x, kind)
    outs = st._forward()
    loss, disc, gen, mmd = _reference_expression(outs[:4], outs[4], 0.0, x, y, kind, **kw)
    loss.backward()
    return p, (loss.item(), disc.item(), gen.item(), mmd.item())


def _grads_equal(p, q, tol=1e-6):
    for k in p:
        a = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        b = q[k].grad if q[k].grad is not None else torch.zeros_like(q[k])
        assert torch.allclose(a, b, rtol=tol, atol=tol), (k, a, b)


@pytest.mark.parametrize("kind", [0, 1])
def test_reference_expression_stays_symbolic_and_matches_torch(kind):
    ref_p, ref_vals = _plain(kind)
    p = _params()
    x, y = _batch(kind)
    st = FakeStep(p, x, kind)
    decoded, reg, missing = _lazy_outputs(st)
    loss, disc, gen, mmd = _reference_expression(decoded, reg, missing, x, y, kind)
    for e in (loss, disc, gen, mmd):
        assert isinstance(e, lazy.LossExpr)
    assert st.real is None
    loss.backward()
    assert st.fast_calls == 1 and st.real is None and st.disc_in_slot
    _grads_equal(p, ref_p)
    got = (loss.item(), disc.item(), gen.item(), mmd.item())
    assert np.allclose(got, ref_vals, rtol=1e-6, atol=1e-6), (got, ref_vals)
    assert float(loss) == loss.item() and "%.3f" % loss.item()


def test_item_before_backward_and_detach():
    _, ref_vals = _plain()
    p = _params()
    x, y = _batch()
    st = FakeStep(p, x)
    loss, disc, gen, mmd = _reference_expression(*_lazy_outputs(st), x, y)
    assert abs(loss.item() - ref_vals[0]) < 1e-6 and abs(disc.item() - ref_vals[1]) < 1e-6     # disc not in its slot yet
    d = loss.detach()
    assert type(d) is torch.Tensor and not d.requires_grad and abs(d.item() - ref_vals[0]) < 1e-6
    assert st.real is None and st.fast_calls == 0


def test_stage_losses_and_scaling_stay_symbolic():
    p = _params()
    x, y = _batch()
    st = FakeStep(p, x)
    loss, disc, gen, mmd = _reference_expression(*_lazy_outputs(st), x, y)
    s1, s2 = gen + mmd, disc + mmd                       # train_beta_vae, mfm_mosi.py:278-281
    assert sorted(s1._coef) == [1, 2, 3, 4] and sorted(s2._coef) == [0, 4]
    assert s1._fast_backward_ok() and s2._fast_backward_ok()
    s1.backward()                                         # stage 1 has no discriminative term: its VALUE is still reported
    assert st.fast_calls == 1 and st.disc_in_slot and abs(disc.item() - F.l1_loss(st.views[3].squeeze(1), y).item()) < 1e-6
    # ... and its gradient is not: only the regulariser reaches c
    assert torch.allclose(p["c"].grad, 2 * 0.3 * p["c"].detach())
    half = loss / 2
    assert isinstance(half, lazy.LossExpr) and not half._fast_backward_ok()      # weights no longer the plan's: autograd's job
    assert isinstance(-loss + 1.0 - 0.5, lazy.LossExpr) and isinstance(2 - loss, lazy.LossExpr)
    assert isinstance(sum([disc, gen, mmd]), lazy.LossExpr)


def _fallback_case(build, kind=0, **kw):
    """build(decoded, reg, x, y) -> loss; evaluated on lazy outputs and on plain tensors: same value, same gradients"""
    q = _params()
    x, y = _batch(kind)
    sq = FakeStep(q, x, kind)
    outs = sq._forward()
    ref = build(outs[:4], outs[4], x, y)
    ref.backward()
    p = _params()
    st = FakeStep(p, x, kind)
    decoded, reg, _ = _lazy_outputs(st)
    loss = build(decoded, reg, x, y)
    val = loss.item()
    loss.backward()
    assert abs(val - ref.item()) < 1e-5 * max(1.0, abs(ref.item())), (val, ref.item())
    _grads_equal(p, q, 1e-5)
    return st


def test_fallback_sum_reduction():
    mse = nn.MSELoss(reduction="sum")
    st = _fallback_case(lambda d, r, x, y: _reference_expression(d, r, 0.0, x, y, mse=mse)[0])
    assert st.real is not None and st.fast_calls == 0


def test_fallback_non_aliasing_target():
    def build(d, r, x, y):
        xc = x.clone()                                   # same numbers, another storage: not provably the batch slice
        return _reference_expression(d, r, 0.0, xc, y)[0]
    st = _fallback_case(build)
    assert st.real is not None and st.fast_calls == 0


def test_fallback_wrong_slice_and_modified_batch():
    def build(d, r, x, y):
        # x_a_hat against the first two LANGUAGE columns: right shape, wrong offset
        return F.mse_loss(d[1], x[:, :, :D[1]]) + r
    _fallback_case(build)
    p = _params()
    x, y = _batch()
    st = FakeStep(p, x)
    decoded, reg, _ = _lazy_outputs(st)
    x.add_(1.0)                                          # the batch changed in place after the forward
    e = F.mse_loss(decoded[0], x[:, :, :D[0]])
    assert not isinstance(e, lazy.LossExpr)


def test_fallback_other_weights_and_reused_output():
    st = _fallback_case(lambda d, r, x, y: _reference_expression(d, r, 0.0, x, y, lda=(1.0, 1.0, 1.0))[0])
    assert st.real is not None and st.fast_calls == 0

    def reuse(d, r, x, y):
        loss = _reference_expression(d, r, 0.0, x, y)[0]
        return loss + 0.1 * d[0].abs().mean() + (d[3] ** 2).mean()       # the outputs used a second way
    st = _fallback_case(reuse)
    assert st.real is not None and st.fast_calls == 0


def test_fallback_tensor_arithmetic_and_torch_functions():
    w = torch.tensor(0.7)
    _fallback_case(lambda d, r, x, y: w * _reference_expression(d, r, 0.0, x, y)[0])
    _fallback_case(lambda d, r, x, y: torch.stack([_reference_expression(d, r, 0.0, x, y)[0], r]).sum())
    _fallback_case(lambda d, r, x, y: (d[3].squeeze(1) - y).abs().mean() + sum(((dd - x[:, :, a:b]) ** 2).mean() for dd, (a, b) in
                                                                              zip(d[:3], ((0, 6), (6, 8), (8, 11)))) + r)


def test_fallback_backward_arguments_and_broadcast_labels():
    p = _params()
    x, y = _batch()
    st = FakeStep(p, x)
    loss = _reference_expression(*_lazy_outputs(st), x, y)[0]
    loss.backward(retain_graph=True)
    assert st.fast_calls == 0 and st.real is not None
    q, _ = _plain()
    _grads_equal(p, q)
    # labels of another shape than y_hat: torch broadcasts [B,1] against [B] -- not the plan's loss
    st2 = FakeStep(_params(), x)
    decoded, reg, _ = _lazy_outputs(st2)
    with pytest.warns(UserWarning):
        e = F.l1_loss(decoded[3], y)
    assert not isinstance(e, lazy.LossExpr)


def test_metadata_reads_do_not_materialise():
    p = _params()
    x, y = _batch()
    st = FakeStep(p, x)
    decoded, reg, _ = _lazy_outputs(st)
    xl = decoded[0]
    assert tuple(xl.shape) == (4, 5, 6) and xl.dim() == 3 and xl.size(2) == 6 and xl.dtype == torch.float32
    assert xl.device.type == "cpu" and xl.numel() == 120 and len(xl) == 4 and xl.requires_grad and isinstance(xl, torch.Tensor)
    assert decoded[3].squeeze(1).shape == (5,) and reg.dim() == 0 and reg.shape == ()
    assert st.real is None
    # reading the numbers materialises (and is right)
    assert torch.allclose(xl.detach().cpu(), st.views[0]) and st.real is not None


def test_stale_step_raises():
    p = _params()
    x, y = _batch()
    st = FakeStep(p, x)
    loss = _reference_expression(*_lazy_outputs(st), x, y)[0]
    st.live = False
    with pytest.raises(RuntimeError):
        loss.backward()
    with pytest.raises(RuntimeError):
        loss.item()


def test_handover_permission_follows_the_guard_aware_optimizer():
    """only an optimizer that honours the gradient guard lets a model use its in-launch hand-overs (factorized_amd.optim.Adam
    marks the models it owns with a weak reference; gone or replaced -> separate launches again; a copy starts unmarked)"""
    import copy
    import gc
    from factorized_amd import configs
    from factorized_amd.mfm_model import MFM_KL_EF
    import factorized_amd.optim as optim
    m = MFM_KL_EF(*configs.canonical_configs(dropout=False))
    assert not m._handover_ok()
    o = optim.Adam(m.parameters())                 # (before .to(device), like the reference: mfm_mosi.py:403 / :414)
    assert m._handover_ok()
    assert not copy.deepcopy(m)._handover_ok()
    del o
    gc.collect()
    assert not m._handover_ok()
    o2 = torch.optim.Adam(m.parameters())
    assert not m._handover_ok() and o2 is not None


def test_a_foreign_optimizer_step_takes_the_handover_permission_away():
    """the permission is granted per step by the optimizer that steps the model.  A stock optimizer built over the same
    parameters WHILE the guard-aware one is still alive (optimizer swap, LR finder) clears it in its step; the guard-aware
    optimizer's next step grants it again."""
    from factorized_amd import configs
    from factorized_amd.mfm_model import MFM_KL_EF
    import factorized_amd.optim as optim
    m = MFM_KL_EF(*configs.canonical_configs(dropout=False))
    o = optim.Adam(m.parameters())
    assert m._handover_ok()
    sgd = torch.optim.SGD(m.parameters(), lr=0.0)
    assert m._handover_ok()                      # building it changes nothing ...
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    sgd.step()                                   # ... stepping the model with it does
    assert not m._handover_ok()
    assert o is not None


def test_second_criterion_on_y_hat_does_not_repoint_the_first(monkeypatch):
    """the step has one discriminative slot; l1_loss(y_hat, y) followed by l1_loss(y_hat, y_other) must leave the first
    expression on y (the second call takes the ordinary path)."""
    from factorized_amd import lazy

    class St(lazy.StepBase):
        loss_kind = 0
        real = None
        def __init__(self):
            self.x = torch.zeros(2, 3, 4)
            self.scalar_view = torch.zeros(())
        def check_live(self, what):
            pass
        def realize(self):
            self.real = [None, None, None, torch.ones(3, 1, requires_grad=True), torch.zeros(())]
            return self.real
    st = St()
    y_hat = lazy.LazyOut(torch.zeros(3, 1), st, 3)
    ya, yb = torch.zeros(3, 1), torch.full((3, 1), 5.0)
    l1 = torch.nn.functional.l1_loss(y_hat, ya)
    assert isinstance(l1, lazy.LossExpr) and st.disc[1] is ya
    l2 = torch.nn.functional.l1_loss(y_hat, yb)          # a different label tensor: materialises, evaluated by torch
    assert not isinstance(l2, lazy.LossExpr) and st.disc[1] is ya
    assert float(l2.detach()) == pytest.approx(4.0)
    assert isinstance(torch.nn.functional.l1_loss(y_hat, ya), (lazy.LossExpr, torch.Tensor))


def test_outputs_read_under_no_grad_still_backpropagate():
    """realize() caches the materialised outputs; a read under torch.no_grad() must not cache them without grad_fn."""
    from factorized_amd import lazy
    seen = {}

    class St(lazy.PlanStep):
        def __init__(self):
            self.real = None
            self.module = type("M", (), {"_flat_leaf": torch.zeros((), requires_grad=True)})()
        def check_live(self, what):
            pass
    import factorized_amd.mfm_model as mm

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, leaf, step):
            seen["grad_mode"] = torch.is_grad_enabled()
            return leaf * 1.0
        @staticmethod
        def backward(ctx, g):
            return g, None
    old = mm._LazyRealFn
    mm._LazyRealFn = Fn
    try:
        st = St()
        with torch.no_grad():
            r = st.realize()
        assert r.requires_grad and r.grad_fn is not None
    finally:
        mm._LazyRealFn = old
