"""The nn.Module mirror keeps the reference's class surface: constructor signature, attribute /
state_dict names and shapes (CPU-only checks; the compute runs in tests/test_gpu_module.py)."""
import pytest
import torch

from factorized_amd import configs, mfm_model as M
from oracle import mfm_oracle as O


def test_state_dict_matches_reference_keys_and_shapes():
    cfgs = configs.canonical_configs()
    m = M.MFM_KL_EF(*cfgs)
    ref = O.build("kl_ef", cfgs)          # pinned to the reference by tests/test_oracle_golden.py
    sd, rd = m.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(rd.keys())
    assert all(tuple(sd[k].shape) == tuple(rd[k].shape) for k in sd)
    m.load_state_dict(rd)                 # a reference checkpoint loads cleanly


def test_mfn_models_state_dict_matches_reference():
    cfgs = configs.canonical_configs()
    for cls, variant, nparam in ((M.MFM_KL, "kl", 741415), (M.MFM, "mmd", 717719)):
        m, ref = cls(*cfgs), O.build(variant, cfgs)
        assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
        assert sum(v.numel() for v in m.state_dict().values()) == nparam
        m.load_state_dict(ref.state_dict())


def test_blocks_have_reference_attributes():
    e = M.encoderLSTM(5, 8)
    d = M.decoderLSTM(24, 5)
    assert isinstance(e.lstm, torch.nn.LSTMCell) and isinstance(e.fc1, torch.nn.Linear) and e.h == 8
    assert d.lstm.weight_ih.shape == (96, 24) and d.fc1.weight.shape == (5, 24) and d.h == 24 and d.d == 5


def test_no_cpu_fallback():
    cfgs = configs.canonical_configs()
    m = M.MFM_KL_EF(*cfgs)
    with pytest.raises(Exception) as ei:
        m.forward(torch.zeros(3, 2, 325))
    assert "no CPU fallback" in str(ei.value) or "GPU" in str(ei.value)
    with pytest.raises(Exception):
        M.encoderLSTM(5, 8).forward(torch.zeros(3, 2, 5))


def test_loss_helpers_match_oracle():
    g = torch.Generator().manual_seed(0)
    mu, lv = torch.randn(7, 5, generator=g), torch.randn(7, 5, generator=g)
    assert torch.allclose(M.loss_KLD(mu, lv), O.kld_sum(mu, lv))
    z, gs = torch.randn(6, 4, generator=g), torch.randn(6, 4, generator=g)
    # loss_MMD itself is a HIP op (mmd_kernel) and refuses CPU tensors; its formula helper is the reference's compute_kernel
    mmd = M.compute_kernel(gs, gs).mean() + M.compute_kernel(z, z).mean() - 2.0 * M.compute_kernel(gs, z).mean()
    assert torch.allclose(mmd, O.mmd(z, gs))
    with pytest.raises(Exception) as ei:
        M.loss_MMD(z, gs)
    assert "no CPU fallback" in str(ei.value)


def test_ablation_and_missing_modality_classes_keep_reference_state_dict():
    """M_A..M_D, MFM_missing, seq2seq, basic_missing (reference mfm_model.py:201-467, 766-1017): same keys, order and
    shapes as the oracle restatements, which tests/test_oracle_extra_golden.py pins to the reference itself; and they
    are importable from mfm_model like in the reference."""
    from oracle import mfm_oracle_extra as X
    from tests.extra_cases import EXTRA, extra_configs
    cfgs = extra_configs()
    for name in EXTRA:
        m = getattr(M, name)(*cfgs)
        ref = X.CLASSES[name](*cfgs)
        sd, rd = m.state_dict(), ref.state_dict()
        assert list(sd.keys()) == list(rd.keys()), name
        assert all(tuple(sd[k].shape) == tuple(rd[k].shape) for k in sd), name
        m.load_state_dict(rd)


def test_flat_layout_guard_word_sits_behind_every_tensor():
    """The guard word of the guarded optimizers (include/mfm_hip.h, mfm_adam_flat_guarded) is a spare granule of the flat buffer:
    inside it (so that a data-parallel all-reduce carries it), behind every tensor, outside every staged-training span."""
    from factorized_amd import configs, engine
    for variant in ("kl_ef", "kl", "mmd"):
        cfgs = configs.canonical_configs(dropout=False)
        lay = engine.FlatLayout(engine.param_shapes(cfgs, variant), variant)
        assert lay.guard % engine.ALIGN == 0 and lay.total == lay.guard + engine.ALIGN
        for (off, n, _) in lay.slots:
            assert off + n <= lay.guard
        assert max(e for _, _, e in lay.group_spans) <= lay.guard
        assert lay.numel <= lay.guard
