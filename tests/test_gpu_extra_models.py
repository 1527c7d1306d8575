"""Ablations M_A..M_D and the missing-modality family (MFM_missing, seq2seq, basic_missing; reference
mfm_model.py:201-467, 766-1017) composed from the HIP ops: every output tensor and every parameter gradient against
the CPU oracle restatement, and the reference's own output / gradient summaries (tests/golden/extra_*.npz).
1e-4 relative fp32."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle_extra as X
from factorized_amd import synth
from tests import cases
from tests.cases import grad_err, rel_err
from tests.extra_cases import EXTRA, SIZES, load_extra

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("name", EXTRA)
def test_extra_model_matches_oracle_and_reference(name, size):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import mfm_model as M
    cfgs, gold, xn, gauss = load_extra(name, size)
    ref = X.CLASSES[name](*cfgs)
    w = synth.make_weights({k: tuple(v.shape) for k, v in ref.state_dict().items()}, seed=1234)
    ref.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    ref.train()
    ref.mmd_gauss = gauss
    model = getattr(M, name)(*cfgs)
    model.load_state_dict(ref.state_dict())
    model = model.cuda()
    model.train()
    model.mmd_gauss = [g.cuda() for g in gauss]
    x = torch.from_numpy(xn)
    rout = ref.forward(x)
    out = model.forward(x.cuda())
    rflat, flat = X.flatten_outputs(rout), X.flatten_outputs(out)
    assert len(flat) == len(rflat) == gold["out_summary"].shape[0]
    worst = 0.0
    for i, (a, b) in enumerate(zip(flat, rflat)):
        assert tuple(a.shape) == tuple(b.shape), i
        if b.numel() and float(b.detach().abs().max()) > 0:
            worst = max(worst, rel_err(a.detach().cpu().numpy(), b.detach().numpy()))
    cases.report("extra_outputs_rel_%s_%s" % (name, size), worst)
    assert worst < TOL
    got = np.stack([cases.summarize(o.detach().cpu().numpy()) for o in flat])
    scale = np.maximum(np.abs(gold["out_summary"][:, :1]), 1e-6)
    assert np.max(np.abs(got - gold["out_summary"]) / scale) < 5 * TOL
    X.test_objective(rout).backward()
    obj = X.test_objective(out)
    assert abs(obj.item() - float(gold["objective"])) < TOL * max(abs(float(gold["objective"])), 1e-3)
    obj.backward()
    rp = dict(ref.named_parameters())
    wg = ("", 0.0)
    rows = []
    for n, p in model.named_parameters():
        q = rp[n]
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            rows.append(np.full(10, np.nan))
            continue
        assert p.grad is not None, n
        rows.append(cases.summarize(p.grad.cpu().numpy()))
        err = grad_err(p.grad.cpu().numpy(), q.grad.numpy(), abs_slack=1e-8)
        if err > wg[1]:
            wg = (n, err)
    cases.report("extra_grad_%s_%s" % (name, size), wg[1])
    assert wg[1] < TOL, wg
    gs = gold["grad_summary"]
    ok = ~np.isnan(gs[:, 0])
    sc = np.maximum(np.abs(gs[ok][:, :1]), 1e-7)
    assert np.max(np.abs(np.stack(rows)[ok] - gs[ok]) / sc) < 5 * TOL
