"""CPU-side checks of the native boundary: the C-ABI library builds/loads and exports every
symbol include/mfm_hip.h declares; host-only entry points behave (no GPU compute here)."""
import ctypes as C
import os
import re

import pytest

from factorized_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mfm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mfm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libmfm_hip.so does not export %s" % n
    # and the ctypes binding covers the header (no drift between the two)
    assert sorted(_lib.exported_names()) == names


def test_abi_version_and_error_text():
    L = _lib.lib()
    assert L.mfm_abi_version() == 4
    rc = L.mfm_gemm_grouped_f32(None, 0, None)
    assert rc == -1
    assert b"no problems" in L.mfm_last_error()


def test_plan_create_is_host_only_and_validates():
    from factorized_amd import configs, engine
    L = _lib.lib()
    cfg = configs.canonical_configs()[0]
    lay = engine.FlatLayout(engine.klef_param_shapes(cfg))
    assert lay.numel == 477294            # reference MFM_KL_EF parameter count
    assert all(o % 64 == 0 for o in lay.offsets.values())
    pc = _lib.PlanConfig()
    pc.d_l, pc.d_a, pc.d_v = cfg["input_dims"]
    pc.zl, pc.za, pc.zv, pc.zy = 32, 8, 80, 32
    pc.fl, pc.fa, pc.fv, pc.fy = 88, 8, 8, 16
    pc.output_dim, pc.loss_kind, pc.T, pc.B = 1, 0, 20, 32
    pc.lda_xl, pc.lda_xa, pc.lda_xv, pc.lda_reg, pc.reg_scale = 1.0, 0.01, 0.5, 1.0, 1.0
    offs = (C.c_int64 * _lib.MFM_KLEF_NPARAM)(*lay.offsets.values())
    h = C.c_void_p(0)
    assert L.mfm_plan_create(C.byref(pc), offs, lay.total, C.byref(h)) == 0
    assert L.mfm_plan_workspace_bytes(h) > 0
    # 3 x forward FLOPs per sample (16,756,192 fwd at the canonical sizes)
    assert abs(L.mfm_plan_flops_per_step(h) / (3 * 32) - 16756192) < 1.0
    L.mfm_plan_destroy(h)
    pc.zl = 300                            # hidden size beyond the register-resident kernels: step-by-step path
    assert L.mfm_plan_create(C.byref(pc), offs, lay.total, C.byref(h)) == 0
    L.mfm_plan_destroy(h)
    pc.zl, pc.T = 32, 0
    assert L.mfm_plan_create(C.byref(pc), offs, lay.total, C.byref(h)) == -1


def test_engine_refuses_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from factorized_amd import configs, engine
    with pytest.raises(_lib.MfmError):
        engine.MFMEngine(configs.canonical_configs())
