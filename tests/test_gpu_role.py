"""Role workgroups of the fold launches at B <= 32 (proj_role_dev.h, dw_role_dev.h): the input projections produced inside the
encoder recurrence launch and every weight-gradient product run inside the encoder BPTT launch, handed over through epoch
flags.  Checked here: (a) they are what runs at the reference's batch size, and the two launches they replace are gone;
(b) results equal the separate-launch form (MFM_PROJ_FOLD=0 / MFM_DW_FOLD=0) to rounding order and the CPU oracle at 1e-4
relative fp32 (BASELINE.json north_star), on ragged sizes (B = 1 .. 32, T = 1 .. 21) and with fewer role workgroups than the
default; (c) trajectories over Adam steps agree; (d) bf16 plans (operands rounded to bf16 in the role blocks)."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle as O
from factorized_amd import configs as C
from factorized_amd import synth
from tests.cases import grad_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
SIZES = [(1, 1), (1, 6), (7, 3), (13, 2), (19, 21), (32, 1), (32, 20), (31, 5),
         (33, 20), (36, 7), (37, 20), (48, 2)]      # round 4: the weight-gradient role form up to B = 38 at T = 20 (projections: B <= 32)


def _engine(cfgs, precision="fp32"):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    e = engine.MFMEngine(cfgs, precision=precision)
    w = synth.make_weights(e.layout.shapes, seed=1234)
    e.load_weights(w)
    return e, w


def _off(monkeypatch, off):
    for k in ("MFM_PROJ_FOLD", "MFM_DW_FOLD", "MFM_WT_IMG"):
        if off:
            monkeypatch.setenv(k, "0")
        else:
            monkeypatch.delenv(k, raising=False)


def _grads(cfgs, B, T, precision="fp32"):
    cfg = cfgs[0]
    e, w = _engine(cfgs, precision)
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    out = e.forward(x, y, train=True, want_xhat=False)
    e.backward(x, y, stage=0)
    torch.cuda.synchronize()
    ld = e.loss_dict(out["losses"])
    g = {n: v.cpu().numpy().copy() for n, v in e.grad_views().items()}
    return e, w, xn, yn, ld, g


def test_role_workgroups_replace_two_launches(monkeypatch):
    cfgs = C.canonical_configs(dropout=False)
    B, T = 32, 20
    for off in (False, True):
        _off(monkeypatch, off)
        e, _ = _engine(cfgs)
        xn, yn = synth.make_batch(cfgs[0]["input_dims"], B, T, seed=7)
        x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
        e.train_step(x, y)
        e.set_timing(T, B, (1 << 30) - 1)
        for _ in range(3):
            e.train_step(x, y)
        torch.cuda.synchronize()
        tab = e.collect_timing(T, B)
        e.set_timing(T, B, 0)
        launches = {k: v["count"] for k, v in tab.items() if v["count"]}
        if off:
            assert launches.get("proj_gemm", 0) == 3 and launches.get("dw_gemm", 0) == 3, launches
        else:
            assert "proj_gemm" not in launches and "dw_gemm" not in launches, launches
            assert sum(launches.values()) == 3 * 6, launches          # enc fwd, dec fwd, fc1, dec bwd, enc bwd, adam
            # the fold launches are charged with the work they took over
            assert tab["enc_seq_bwd"]["flops"] > 4 * tab["dec_seq_bwd"]["flops"]


@pytest.mark.parametrize("B,T", SIZES)
def test_role_form_equals_separate_launches_and_oracle(B, T, monkeypatch):
    cfgs = C.canonical_configs(dropout=False)
    _off(monkeypatch, False)
    _, w, xn, yn, ld1, g1 = _grads(cfgs, B, T)
    _off(monkeypatch, True)
    _, _, _, _, ld0, g0 = _grads(cfgs, B, T)
    for k in ld1:
        assert abs(ld1[k] - ld0[k]) <= 1e-5 * max(abs(ld0[k]), 1e-3), (k, ld1[k], ld0[k])
    worst = max(grad_err(g1[n], g0[n]) for n in g0)
    assert worst < 2e-5, worst
    # and the oracle (restated reference path) on the same inputs
    torch.set_num_threads(4)
    m = O.build("kl_ef", cfgs)
    O.load_numpy_weights(m, w)
    m.train()
    terms = O.loss_terms(m, torch.from_numpy(xn), torch.from_numpy(yn), cfgs[0])
    terms["loss"].backward()
    assert abs(ld1["loss"] - terms["loss"].item()) <= TOL * abs(terms["loss"].item())
    for n, p in m.named_parameters():
        assert grad_err(g1[n], p.grad.numpy()) < TOL, n


@pytest.mark.parametrize("B,T", [(33, 20), (40, 20), (48, 7), (48, 1)])
def test_weight_gradient_roles_beyond_32_rows(B, T, monkeypatch):
    """Round 4: the weight-gradient role workgroups take up to 64 batch rows (64 stamp words per (encoder, time step), chunks
    that no longer end on time-step boundaries, two accumulator rounds when the tile sets outnumber the idle CUs).  By default
    they run only while their work fits behind the BPTT (B <= 38 at T = 20); MFM_DW_FOLD_MAXITER lifts that for the test.  The
    projections keep their own launch beyond 32 rows."""
    cfgs = C.canonical_configs(dropout=False)
    _off(monkeypatch, False)
    monkeypatch.setenv("MFM_DW_FOLD_MAXITER", "99")
    e, w, xn, yn, ld1, g1 = _grads(cfgs, B, T)
    p = e.plan(T, B)
    assert p.get_option("dw_roles_active") == 1 and p.get_option("proj_roles_active") == 0
    monkeypatch.delenv("MFM_DW_FOLD_MAXITER")
    _off(monkeypatch, True)
    _, _, _, _, ld0, g0 = _grads(cfgs, B, T)
    assert max(grad_err(g1[n], g0[n]) for n in g0) < 2e-5
    torch.set_num_threads(4)
    m = O.build("kl_ef", cfgs)
    O.load_numpy_weights(m, w)
    m.train()
    terms = O.loss_terms(m, torch.from_numpy(xn), torch.from_numpy(yn), cfgs[0])
    terms["loss"].backward()
    for n, q in m.named_parameters():
        assert grad_err(g1[n], q.grad.numpy()) < TOL, n
    assert e.check_status() == 0


@pytest.mark.parametrize("roles", ["32", "64"])
def test_fewer_role_workgroups(roles, monkeypatch):
    """Fewer role workgroups than idle CUs: more time steps per projection producer, two encoder tiles per slot."""
    cfgs = C.canonical_configs(dropout=False)
    _off(monkeypatch, True)
    _, _, _, _, ld0, g0 = _grads(cfgs, 32, 20)
    _off(monkeypatch, False)
    monkeypatch.setenv("MFM_PROJ_FOLD_ROLES", roles)
    monkeypatch.setenv("MFM_DW_FOLD_ROLES", "48" if roles == "32" else "100")
    e, _, _, _, ld1, g1 = _grads(cfgs, 32, 20)
    assert abs(ld1["loss"] - ld0["loss"]) <= 1e-5 * abs(ld0["loss"])
    assert max(grad_err(g1[n], g0[n]) for n in g0) < 2e-5


def test_role_trajectory_equals_separate_launches(monkeypatch):
    cfgs = C.canonical_configs(dropout=False)
    B, T, steps = 32, 20, 12
    xs = [synth.make_batch(cfgs[0]["input_dims"], B, T, seed=20 + i) for i in range(3)]
    traj = []
    for off in (False, True):
        _off(monkeypatch, off)
        e, _ = _engine(cfgs)
        losses = []
        for i in range(steps):
            xn, yn = xs[i % 3]
            l = e.train_step(torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda())
            losses.append(e.loss_dict(l)["loss"])
        torch.cuda.synchronize()
        traj.append((np.array(losses), e.params.cpu().numpy().copy()))
    assert np.allclose(traj[0][0], traj[1][0], rtol=2e-5)
    assert np.abs(traj[0][1] - traj[1][1]).max() < 2e-5


def test_role_workgroups_bf16_plan(monkeypatch):
    """bf16 plans at B <= 32: the role blocks round both operands to bf16 (what the bf16-operand GEMMs they replace compute);
    against the separate launches only the accumulation order differs."""
    cfgs = C.canonical_configs(dropout=False)
    _off(monkeypatch, False)
    _, _, _, _, ld1, g1 = _grads(cfgs, 32, 20, precision="bf16")
    _off(monkeypatch, True)
    _, _, _, _, ld0, g0 = _grads(cfgs, 32, 20, precision="bf16")
    assert abs(ld1["loss"] - ld0["loss"]) <= 2e-3 * abs(ld0["loss"])
    for n in g0:
        den = max(np.linalg.norm(g0[n]), 1e-9)
        assert np.linalg.norm(g1[n] - g0[n]) / den < 2e-2, n


@pytest.mark.parametrize("B,T", [(16, 1), (16, 7), (32, 2), (32, 20)])
def test_one_call_step_equals_forward_plus_backward(B, T, monkeypatch):
    """grad_step (one enqueue; the role workgroups also clear the gradient buffer and the loss slots) repeated on the same
    batch: same losses and gradients every time as forward() + backward(), with the role workgroups and without."""
    cfgs = C.canonical_configs(dropout=False)
    outs = []
    for off in (False, True):
        _off(monkeypatch, off)
        e, _ = _engine(cfgs)
        xn, yn = synth.make_batch(cfgs[0]["input_dims"], B, T, seed=7)
        x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
        for rep in range(3):                       # repeated: a hand-over that is only usually in time would show up
            l = e.grad_step(x, y)
            torch.cuda.synchronize()
            outs.append((e.loss_dict(l), e.grads.cpu().numpy().copy()))
        out = e.forward(x, y, train=True, want_xhat=False)
        e.backward(x, y, stage=0)
        torch.cuda.synchronize()
        outs.append((e.loss_dict(out["losses"]), e.grads.cpu().numpy().copy()))
    ref_l, ref_g = outs[-1]
    for ld, g in outs[:-1]:
        for k in ref_l:
            assert abs(ld[k] - ref_l[k]) <= 1e-5 * max(abs(ref_l[k]), 1e-3), (k, ld[k], ref_l[k])
        assert grad_err(g, ref_g) < 2e-5


def test_shared_device_switch_restores_separate_launches(monkeypatch):
    """MFM_SHARED_DEVICE=1 (several processes / streams on one GPU: two queues' launches could block each other's producers):
    no in-launch hand-overs -- the projection and weight-gradient launches are back; the weight images (no hand-over) stay."""
    cfgs = C.canonical_configs(dropout=False)
    B, T = 32, 20
    _off(monkeypatch, False)
    monkeypatch.setenv("MFM_SHARED_DEVICE", "1")
    e, _ = _engine(cfgs)
    xn, yn = synth.make_batch(cfgs[0]["input_dims"], B, T, seed=7)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    e.train_step(x, y)
    e.set_timing(T, B, (1 << 30) - 1)
    for _ in range(2):
        e.train_step(x, y)
    torch.cuda.synchronize()
    tab = e.collect_timing(T, B)
    e.set_timing(T, B, 0)
    launches = {k: v["count"] for k, v in tab.items() if v["count"]}
    assert launches.get("proj_gemm", 0) == 2 and launches.get("dw_gemm", 0) == 2, launches


@pytest.mark.parametrize("switch", ["MFM_WF_IMG", "MFM_LATENT_PRELOAD", "MFM_LATENT_SPLIT"])
@pytest.mark.parametrize("B,T", [(32, 20), (7, 3), (19, 21), (1, 2), (48, 5)])
def test_round6_forms_equal_the_forms_they_replaced(monkeypatch, switch, B, T):
    """Round 6 changed three things inside the small-batch launches, each with a switch that restores the previous form: the
    decoders' weights from forward-order images (MFM_WF_IMG=0: strided gathers + add), the latent forward chain's tables
    preloaded in front of the time loop with h_T taken from LDS (MFM_LATENT_PRELOAD=0: loaded behind the last step, h_T read back
    from memory), the latent backward chain handing d h_T over through LDS with its stores behind the BPTT's weight requests
    (MFM_LATENT_SPLIT=0: stores first, d h_T from memory).  Same arithmetic in the same order: losses identical, gradients equal
    to rounding order of the atomic sums, both forms within 1e-4 of the CPU oracle."""
    cfgs = C.canonical_configs(dropout=False)
    monkeypatch.delenv(switch, raising=False)
    _, w, xn, yn, ld1, g1 = _grads(cfgs, B, T)
    monkeypatch.setenv(switch, "0")
    _, _, _, _, ld0, g0 = _grads(cfgs, B, T)
    monkeypatch.delenv(switch, raising=False)
    for k in ("disc", "gen", "reg", "loss"):
        assert abs(ld1[k] - ld0[k]) <= 1e-6 * max(abs(ld0[k]), 1.0), (k, ld1[k], ld0[k])
    m = O.build("kl_ef", cfgs)
    O.load_numpy_weights(m, w)
    m.train()
    O.loss_terms(m, torch.from_numpy(xn), torch.from_numpy(yn), cfgs[0])["loss"].backward()
    for n, p in m.named_parameters():
        r = p.grad.numpy()
        assert grad_err(g1[n], r) < TOL, (switch, "new form", n)
        assert grad_err(g0[n], r) < TOL, (switch, "old form", n)
        assert grad_err(g1[n], g0[n]) < 1e-5, (switch, "new vs old", n)


@pytest.mark.parametrize("switch", ["MFM_LATENT_TAIL", "MFM_LATENT_BWD_HEAD"])
@pytest.mark.parametrize("B,T", [(32, 20), (7, 3), (19, 21), (1, 2), (36, 5)])
def test_chain_tails_on_the_decoder_launches_equal_the_whole_chains(monkeypatch, switch, B, T):
    """Round 6: the latent chains' stages that nothing of the decoders waits for run on idle CUs of the DECODER launches -- forward:
    classifier, logvar heads, losses, y_hat as tail blocks of the decoder recurrence launch (MFM_LATENT_TAIL=0: inside the encoder
    launch); backward: the discriminative / KLD seeds and the classifier / logvar stages as head blocks of the decoder BPTT launch
    (MFM_LATENT_BWD_HEAD=0: in front of the encoder BPTT).  Same arithmetic: losses, y_hat and gradients equal to rounding order,
    both forms within 1e-4 of the CPU oracle (B = 36: beyond the projection role form, the plain launches)."""
    cfgs = C.canonical_configs(dropout=False)
    monkeypatch.delenv(switch, raising=False)
    e1, w, xn, yn, ld1, g1 = _grads(cfgs, B, T)
    monkeypatch.setenv(switch, "0")
    e0, _, _, _, ld0, g0 = _grads(cfgs, B, T)
    monkeypatch.delenv(switch, raising=False)
    for k in ("disc", "gen", "reg", "loss"):
        assert abs(ld1[k] - ld0[k]) <= 1e-6 * max(abs(ld0[k]), 1.0), (k, ld1[k], ld0[k])
    m = O.build("kl_ef", cfgs)
    O.load_numpy_weights(m, w)
    m.train()
    terms = O.loss_terms(m, torch.from_numpy(xn), torch.from_numpy(yn), cfgs[0])
    terms["loss"].backward()
    ref_loss = float(terms["loss"].detach())
    assert abs(ld1["loss"] - ref_loss) <= TOL * abs(ref_loss)
    for n, p in m.named_parameters():
        r = p.grad.numpy()
        assert grad_err(g1[n], r) < TOL, (switch, "split form", n)
        assert grad_err(g0[n], r) < TOL, (switch, "whole chain", n)
        assert grad_err(g1[n], g0[n]) < 1e-5, (switch, "split vs whole", n)
