"""What happens when an in-launch hand-over of the B <= 32 step FAILS (csrc/proj_role_dev.h, dw_role_dev.h: role workgroups
inside the encoder launches, consumers that spin on epoch flags).  The mechanism needs the launch to have the GPU to itself;
another process / stream can keep the producers off the device until a consumer's wait gives up.  Contract tested here:

  * the consumer that gives up sets the plan's sticky STATUS word and stores a NaN into the GUARD word of the gradient buffer;
  * every Adam form (fused step, flat, spans, drop-in optimizer, P2P all-reduce + Adam) reads the guard first and leaves
    parameters and moments bit-unchanged;
  * while the status word is set every further backward poisons the guard again (nothing trains on a plan in error);
  * the host sees the word at the sync it already has (loss_dict) or in check_status(): it clears the word, switches the engine
    to the separate launches for the rest of the run and raises MfmError -- the next step is green on the fallback.

The failures are INJECTED (plan option "inject_fault": one producer does not raise its flag), then provoked for real with a busy
co-tenant stream; plus a bounded long run of the role workgroups against the separate launches."""
import os

import numpy as np
import pytest
import torch

from factorized_amd import configs as C
from factorized_amd import synth
from factorized_amd._lib import MfmError
from tests import cases

pytestmark = pytest.mark.gpu
B, T = 32, 20


def _engine(timeout_us=3000, seed=1234, handover=True):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    e = engine.MFMEngine(C.canonical_configs(dropout=False))
    e.handover = handover
    e.handover_timeout_us = timeout_us          # (tests: a consumer gives up after 3 ms instead of 50)
    e.load_weights(synth.make_weights(e.layout.shapes, seed=seed))
    return e


def _batches(n=4):
    cfg = C.canonical_configs(dropout=False)[0]
    out = []
    for i in range(n):
        xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=100 + i)
        out.append((torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()))
    return out


def _snap(e):
    torch.cuda.synchronize()
    return [t.detach().cpu().numpy().copy() for t in (e.params, e.adam_m, e.adam_v)]


def _same(a, b):
    return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))


@pytest.mark.parametrize("fault", [1, 2])
def test_injected_fault_skips_the_update_raises_and_falls_back(fault):
    """fault 1: a projection producer never raises its flag (forward launch); 2: a BPTT workgroup never stamps its last gate
    gradients (backward launch).  Parameters / moments bit-unchanged, status raised, error at the next loss read, the engine
    on separate launches afterwards, and the following steps equal a run that never had role workgroups."""
    e = _engine()
    ref = _engine(handover=False)
    data = _batches()
    for i in range(3):
        e.train_step(*data[i % 4]); ref.train_step(*data[i % 4])
    p = e.plan(T, B)
    assert p.get_option("proj_roles_active") == 1 and p.get_option("dw_roles_active") == 1
    before = _snap(e)
    p.set_option("inject_fault", fault)
    losses = e.train_step(*data[3])
    after = _snap(e)
    assert _same(before, after), "a step whose hand-over failed reached the parameters"
    assert np.isnan(e.grads[e.layout.guard].item())
    assert int(p.state[8:9].view(torch.int32).item()) == fault           # status bit 0 / bit 1
    with pytest.raises(MfmError, match="hand-over"):
        e.loss_dict(losses)
    assert e.handover is False and e.handover_failures == 1
    assert p.get_option("handover") == 0
    torch.cuda.synchronize()
    assert int(p.state[8:9].view(torch.int32).item()) == 0                # cleared by the host
    # the skipped step consumed an Adam step count on the host; mirror that in the reference run, then both train on
    ref.step_count += 1
    for g in ref.group_steps:
        ref.group_steps[g] = ref.step_count
    for i in range(4):
        l1 = e.train_step(*data[i]); l2 = ref.train_step(*data[i])
    d1, d2 = e.loss_dict(l1), ref.loss_dict(l2)
    assert p.get_option("proj_roles_active") == 0 and p.get_option("dw_roles_active") == 0
    assert np.isfinite(d1["loss"]) and abs(d1["loss"] - d2["loss"]) < 1e-5 * abs(d2["loss"])
    a, b = _snap(e)[0], _snap(ref)[0]
    # elements whose gradient is rounding noise move by up to lr per step in a direction the noise decides
    assert np.max(np.abs(a - b)) < 5e-3 and np.mean(np.abs(a - b)) < 1e-6


def test_plan_in_error_state_keeps_skipping_until_the_host_looks():
    e = _engine()
    data = _batches()
    for i in range(2):
        e.train_step(*data[i])
    p = e.plan(T, B)
    before = _snap(e)
    p.set_option("inject_fault", 1)
    for i in range(4):                        # nobody reads a loss: the status word stays set, every backward re-poisons the guard
        e.train_step(*data[i])
    assert _same(before, _snap(e))
    assert e.check_status(raise_on_error=False) == 1
    assert e.handover is False
    e.train_step(*data[0])
    assert not _same(before, _snap(e))
    assert e.check_status() == 0
    assert np.isfinite(_snap(e)[0]).all()


def test_grad_step_then_guarded_adam_and_spans():
    """the data-parallel split (grad_step + flat Adam) and the staged form (Adam over spans) honour the guard as well"""
    e = _engine()
    data = _batches()
    e.train_step(*data[0])
    p = e.plan(T, B)
    before = _snap(e)
    p.set_option("inject_fault", 2)
    e.grad_step(*data[1])
    e.adam(lr=1e-3)
    assert _same(before, _snap(e))
    assert e.check_status(raise_on_error=False) == 2
    # staged step on a plan whose status is raised again by hand (fault in the forward of a stage-1 step)
    e.set_handover(True)
    p.set_option("inject_fault", 1)
    before = _snap(e)
    e.train_step(*data[2], stage=1)
    assert _same(before, _snap(e))
    with pytest.raises(MfmError):
        e.check_status()


def test_world1_role_step_through_the_p2p_adam_entry():
    """The deployment composition of the data-parallel step -- mfm_plan_grad_step WITH role workgroups (5 launches) followed by
    the P2P all-reduce + Adam launch -- cannot run on one device with several ranks (ranks that share a GPU switch the
    hand-overs off, train._mark_shared_device).  What can: the same two calls at world size 1 (the P2P entry point then is
    its Adam-only launch, same arithmetic, same guard word) against the fused single-GPU step, and a guard raised by an
    injected fault travelling through that entry."""
    from factorized_amd import comm
    e1, e2 = _engine(), _engine()
    data = _batches()
    ar = comm.P2PAllReduce(1, 0, e1.grads.numel())
    for i in range(3):
        e1.grad_step(*data[i]); ar.allreduce_adam(e1, 1e-3, 1.0)
        e2.train_step(*data[i])
    p = e1.plan(T, B)
    assert p.get_option("proj_roles_active") == 1 and p.get_option("dw_roles_active") == 1
    a, b = _snap(e1), _snap(e2)
    assert np.max(np.abs(a[0] - b[0])) < 2e-5                   # (atomics order of the decoder-side sums)
    p.set_option("inject_fault", 2)
    e1.grad_step(*data[3]); ar.allreduce_adam(e1, 1e-3, 1.0)
    assert _same(a, _snap(e1))
    with pytest.raises(MfmError):
        e1.check_status()
    ar.close()


def test_module_path_with_dropin_optimizer_skips_and_reports():
    """The reference's unchanged loop (model.forward, torch losses, loss.backward(), optimizer.step()) on MFM_KL_EF with
    factorized_amd.optim.Adam: a failed hand-over in the backward launch leaves the parameters alone; with the fault in the
    forward the returned regulariser is NaN as well (the loss is loud), and the engine reports both."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn as nn
    import factorized_amd.optim as optim
    from factorized_amd import mfm_model as M
    cfgs = C.canonical_configs(dropout=False)
    cfg = cfgs[0]
    model = M.MFM_KL_EF(*cfgs)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = synth.make_weights(shapes, seed=1234)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    opt = optim.Adam(model.parameters())
    model = model.to("cuda").train()
    eng = model.engine
    eng.handover_timeout_us = 3000
    (X, y), = _batches(1)
    d_l, d_a, d_v = cfg["input_dims"]
    mse, l1 = nn.MSELoss(), nn.L1Loss()

    def step():
        opt.zero_grad()
        (xl, xa, xv, yh), kld, miss = model.forward(X)
        loss = l1(yh.squeeze(1), y) + cfg["lda_xl"] * mse(xl, X[:, :, :d_l]) + cfg["lda_xa"] * mse(xa, X[:, :, d_l:d_l + d_a]) \
            + cfg["lda_xv"] * mse(xv, X[:, :, d_l + d_a:]) + cfg["lda_mmd"] * kld + miss
        loss.backward()
        opt.step()
        return loss, kld

    for _ in range(2):
        step()
    p = eng.plan(T, B)
    for o in ("handover_timeout_us",):
        p.set_option(o, 3000)
    torch.cuda.synchronize()
    before = [q.detach().cpu().numpy().copy() for q in model.parameters()]
    p.set_option("inject_fault", 2)
    loss, _ = step()
    torch.cuda.synchronize()
    after = [q.detach().cpu().numpy().copy() for q in model.parameters()]
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
    with pytest.raises(MfmError):
        eng.check_status()
    step()                                                   # separate launches now
    torch.cuda.synchronize()
    assert any(not np.array_equal(a, q.detach().cpu().numpy()) for a, q in zip(after, model.parameters()))
    # forward fault: what the loop reads is NaN
    eng.set_handover(True)
    step()
    p.set_option("inject_fault", 1)
    before = [q.detach().cpu().numpy().copy() for q in model.parameters()]
    loss, kld = step()
    # (round 5: the loop's own `.item()` brings the status word along -- it warns, clears and falls back by itself)
    with pytest.warns(RuntimeWarning, match="hand-over"):
        k = kld.item()
    assert np.isnan(k) and np.isnan(loss.item())
    assert all(np.array_equal(a, q.detach().cpu().numpy()) for a, q in zip(before, model.parameters()))
    assert not eng.handover and eng.handover_failures == 2 and eng.check_status(raise_on_error=False) == 0
    step()
    torch.cuda.synchronize()
    assert any(not np.array_equal(a, q.detach().cpu().numpy()) for a, q in zip(before, model.parameters()))


def test_bounded_stress_roles_on_vs_off():
    """600 fused steps on 8 batches with the role workgroups against the separate launches (scripts/stress_role.py runs
    6000): no wait gives up, losses finite, trajectories agree as closely as two separate-launch runs agree with each other."""
    steps = 600
    data = _batches(4) + _batches(4)
    runs = []
    for handover in (True, False, False):
        e = _engine(timeout_us=50000, handover=handover)
        tr = []
        for i in range(steps):
            l = e.train_step(*data[i % 8], lr=1e-4)
            if i % 100 == 99:
                tr.append(e.loss_dict(l)["loss"])             # (raises if a hand-over failed)
        assert e.check_status() == 0 and e.handover is handover and e.handover_failures == 0
        runs.append((np.array(tr), _snap(e)[0]))
    assert np.isfinite(runs[0][0]).all() and np.isfinite(runs[0][1]).all()
    dl = np.abs(runs[0][0] - runs[1][0]).max() / np.abs(runs[1][0]).max()
    dp = np.abs(runs[0][1] - runs[1][1]).max()
    nl = np.abs(runs[2][0] - runs[1][0]).max() / np.abs(runs[1][0]).max()
    npar = np.abs(runs[2][1] - runs[1][1]).max()
    assert dl < max(10 * nl, 2e-2) and dp < max(10 * npar, 5e-2), (dl, dp, nl, npar)


def test_cotenant_stream_is_survived():
    """A second stream of this process keeps the GPU busy with large GEMMs while 200 fused steps run with the role workgroups
    (default 50 ms time-out).  Either no hand-over fails, or the failures are survived: never a NaN / garbage update, an
    MfmError at a loss read, separate launches afterwards, training completes."""
    e = _engine(timeout_us=50000)
    ref = _engine(timeout_us=50000, handover=False)
    data = _batches()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.randn(4096, 4096, device="cuda")
    raised = 0
    torch.cuda.synchronize()
    for i in range(200):
        if i % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(6):
                    a = torch.mm(a, b) * 1e-2
        l = e.train_step(*data[i % 4], lr=1e-4)
        ref.train_step(*data[i % 4], lr=1e-4)
        if i % 10 == 9:
            try:
                assert np.isfinite(e.loss_dict(l)["loss"])
            except MfmError:
                raised += 1
    try:
        e.check_status()
    except MfmError:
        raised += 1
    torch.cuda.synchronize()
    assert raised == e.handover_failures and raised <= 1          # (after the first failure nothing can wait any more)
    pa, pb = _snap(e)[0], _snap(ref)[0]
    assert np.isfinite(pa).all()
    if raised == 0:
        assert np.abs(pa - pb).max() < 5e-2
    from tests import cases
    cases.report("cotenant_handover_failures", float(raised))


@pytest.mark.parametrize("B,T", [(32, 1), (16, 1), (32, 2)])
def test_counting_loop_short_sequences(B, T):
    """The loop that found round 4's store / flag race (scripts/handover_flake.py; 22 wrong steps in 1500 at T = 1 before the
    fix, profiles/r04_handover_safety.txt), now part of every GPU run: 1500 role-workgroup gradient steps per shape against the
    separate-launch gradients of the same batch -- at T <= 2 the consumers sit right behind their producers, the case where a
    flag can overtake its data.  Not one tensor of one step may be off."""
    cfgs = C.canonical_configs(dropout=False)
    cfg = cfgs[0]
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    ref = _engine(handover=False, timeout_us=50000)
    ref.grad_step(x, y)
    torch.cuda.synchronize()
    rg = ref.grads.clone()
    scale = {n: max(float(v.abs().max()), 1e-12) for n, v in ref.grad_views().items()}
    e = _engine(timeout_us=50000)
    bad, names = 0, {}
    for r in range(1500):
        e.grad_step(x, y)
        d = (e.grads - rg).abs()
        if float(d.max()) > 1e-5:
            nb = [n for n, v in e.layout.views(d).items() if float(v.max()) > 1e-4 * scale[n] + 1e-7]
            if nb:
                bad += 1
                for n in nb:
                    names[n] = names.get(n, 0) + 1
    assert e.plan(T, B).get_option("dw_roles_active") == 1 and e.plan(T, B).get_option("proj_roles_active") == 1
    assert bad == 0, (bad, names)
    assert e.check_status() == 0


_COTENANT = r"""
import os, sys, time, torch
ready, stop = sys.argv[1], sys.argv[2]
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
torch.mm(a, b); torch.cuda.synchronize()
open(ready, "w").write("1")
t0 = time.time()
while not os.path.exists(stop) and time.time() - t0 < 60:
    for _ in range(8):
        a = torch.mm(a, b) * 1e-2
    torch.cuda.synchronize()
"""


def test_cotenant_process_is_survived(tmp_path):
    """A SECOND PROCESS on the same GPU (the realistic co-tenant of a 0.5 M-parameter model: its queue is scheduled by the
    hardware against ours, it shares no stream order with us) runs 4096^3 GEMMs back to back while 300 fused steps go through
    the role workgroups with the default 50 ms time-out.  Either no hand-over fails, or the failure is survived: an MfmError at
    a loss read, separate launches afterwards, never a NaN or a garbage update, the run finishes on the reference's losses."""
    import subprocess
    import sys
    import time
    ready, stop = str(tmp_path / "ready"), str(tmp_path / "stop")
    child = subprocess.Popen([sys.executable, "-c", _COTENANT, ready, stop])
    try:
        t0 = time.time()
        while not os.path.exists(ready):
            assert child.poll() is None, "co-tenant process died"
            assert time.time() - t0 < 180, "co-tenant process did not come up"
            time.sleep(0.1)
        e = _engine(timeout_us=50000)
        ref = _engine(timeout_us=50000, handover=False)
        data = _batches()
        raised = 0
        for i in range(300):
            l = e.train_step(*data[i % 4], lr=1e-4)
            ref.train_step(*data[i % 4], lr=1e-4)
            if i % 10 == 9:
                try:
                    assert np.isfinite(e.loss_dict(l)["loss"])
                except MfmError:
                    raised += 1
        try:
            e.check_status()
        except MfmError:
            raised += 1
        torch.cuda.synchronize()
        assert child.poll() is None, "the co-tenant was meant to be running the whole time"
    finally:
        open(stop, "w").write("1")
        try:
            child.wait(timeout=60)
        except Exception:
            child.kill()
    # "never fails" and "fails and recovers" must be told apart: the observed count goes on record with every GPU run
    # (profiles/*parity_worst.jsonl: cotenant_handover_failures, cotenant_steps)
    cases.report("cotenant_handover_failures", float(e.handover_failures))
    cases.report("cotenant_steps", 300.0)
    assert raised == e.handover_failures and raised <= 1
    assert np.isfinite(_snap(e)[0]).all()
    # skipped steps aside (at most the ones between a failure and the loss read that reported it), the run followed the
    # separate-launch reference
    la, lb = e.loss_dict(e.train_step(*data[0], lr=1e-4))["loss"], ref.loss_dict(ref.train_step(*data[0], lr=1e-4))["loss"]
    assert abs(la - lb) < 5e-2 * abs(lb), (la, lb, raised)
