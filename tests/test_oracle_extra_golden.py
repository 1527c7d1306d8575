"""Pins oracle/mfm_oracle_extra.py (ablations M_A..M_D, MFM_missing, seq2seq, basic_missing) to the reference's own
outputs (tests/golden/extra_*.npz, make_golden.py::run_extra).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import mfm_oracle_extra as X
from factorized_amd import synth
from tests import cases
from tests.extra_cases import EXTRA, SIZES, extra_configs, load_extra


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("name", EXTRA)
def test_oracle_extra_matches_reference(name, size):
    torch.set_num_threads(1)
    cfgs, gold, x, gauss = load_extra(name, size)
    m = X.CLASSES[name](*cfgs)
    assert [n for n, _ in m.named_parameters()] == list(gold["param_names"])
    w = synth.make_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    m.train()
    m.mmd_gauss = gauss
    out = m.forward(torch.from_numpy(x))
    flat = X.flatten_outputs(out)
    assert len(flat) == gold["out_summary"].shape[0]
    got = np.stack([cases.summarize(o.detach().numpy()) for o in flat])
    assert np.allclose(got, gold["out_summary"], rtol=1e-5, atol=1e-6)
    obj = X.test_objective(out)
    assert abs(obj.item() - float(gold["objective"])) < 1e-6 * max(1.0, abs(float(gold["objective"])))
    obj.backward()
    gs = np.stack([cases.summarize(p.grad.numpy()) if p.grad is not None else np.full(10, np.nan) for p in m.parameters()])
    both_nan = np.isnan(gs) & np.isnan(gold["grad_summary"])
    assert np.allclose(np.where(both_nan, 0, gs), np.where(both_nan, 0, gold["grad_summary"]), rtol=1e-5, atol=1e-7)
