"""ctypes binding of libmfm_hip.so (the C ABI declared in include/mfm_hip.h).

The HIP library is the product: there is NO CPU fallback.  Importing this module
without the built library raises immediately with the build instruction.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MFM_LIB_PATH: another build of the same library -- instrumented debug builds, scripts/seq_step_timeline.py)
LIB_PATH = os.environ.get("MFM_LIB_PATH") or os.path.join(_HERE, "libmfm_hip.so")

MFM_KLEF_NPARAM = 78
MFM_LOSS_SLOTS = 8
MFM_MAX_SEQ = 6
ABI_VERSION = 4


class MfmError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("c2", C.c_void_p),
                ("bias", C.c_void_p), ("bias2", C.c_void_p),
                ("a_sz", C.c_int64), ("a_sm", C.c_int64), ("a_sk", C.c_int64),
                ("b_sz", C.c_int64), ("b_sk", C.c_int64), ("b_sn", C.c_int64),
                ("c_sz", C.c_int64), ("ldc", C.c_int64), ("bias_sz", C.c_int64),
                ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("n_valid", C.c_int32),
                ("batch", C.c_int32), ("split_k", C.c_int32), ("accumulate", C.c_int32),
                ("alpha", C.c_float), ("a_bf16", C.c_int32), ("c_bf16", C.c_int32), ("reserved_", C.c_int32 * 2)]


class SeqDesc(C.Structure):
    _fields_ = [("gates", C.c_void_p), ("hs", C.c_void_p), ("cs", C.c_void_p),
                ("w_hh", C.c_void_p), ("w_ih", C.c_void_p), ("b_ih", C.c_void_p), ("b_hh", C.c_void_p),
                ("h_init", C.c_void_p), ("ld_init", C.c_int64),
                ("dh_ext", C.c_void_p), ("ld_dh", C.c_int64),
                ("d_h_init", C.c_void_p), ("ld_dinit", C.c_int64),
                ("h", C.c_int32), ("is_dec", C.c_int32), ("dc_ext", C.c_void_p), ("w_pack", C.c_void_p),
                ("store_bf16", C.c_int32), ("bf16_dot", C.c_int32), ("h_last", C.c_void_p)]


class MemDesc(C.Structure):
    _fields_ = [("a1", C.c_void_p), ("a2", C.c_void_p), ("chat", C.c_void_p),
                ("w1m", C.c_void_p), ("w2m", C.c_void_p), ("w1b", C.c_void_p), ("b1b", C.c_void_p),
                ("w2b", C.c_void_p), ("b2b", C.c_void_p),
                ("gam1", C.c_void_p), ("gam2", C.c_void_p), ("mems", C.c_void_p), ("mem_out", C.c_void_p),
                ("dmem_out", C.c_void_p), ("du1", C.c_void_p), ("du2", C.c_void_p), ("dchat", C.c_void_p),
                ("T", C.c_int32), ("B", C.c_int32), ("M", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32),
                ("train", C.c_int32), ("p1", C.c_float), ("p2", C.c_float), ("seed", C.c_uint64),
                ("seed_dev", C.c_void_p), ("ld_wm", C.c_int64), ("dchat_pre_tanh", C.c_int32), ("reserved_", C.c_int32)]


class AdamSpan(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("step", C.c_int32), ("reserved", C.c_int32)]


MFM_ADAM_MAX_SPANS = 8


class LossWeights(C.Structure):
    _fields_ = [("disc", C.c_float), ("gen_l", C.c_float), ("gen_a", C.c_float), ("gen_v", C.c_float), ("reg", C.c_float),
                ("write_disc_loss", C.c_int32)]


class PlanConfig(C.Structure):
    _fields_ = [("d_l", C.c_int32), ("d_a", C.c_int32), ("d_v", C.c_int32),
                ("zl", C.c_int32), ("za", C.c_int32), ("zv", C.c_int32), ("zy", C.c_int32),
                ("fl", C.c_int32), ("fa", C.c_int32), ("fv", C.c_int32), ("fy", C.c_int32),
                ("output_dim", C.c_int32), ("loss_kind", C.c_int32),
                ("T", C.c_int32), ("B", C.c_int32),
                ("lda_xl", C.c_float), ("lda_xa", C.c_float), ("lda_xv", C.c_float), ("lda_reg", C.c_float),
                ("drop_zy", C.c_float), ("drop_zl", C.c_float), ("drop_za", C.c_float),
                ("drop_zv", C.c_float), ("drop_y", C.c_float),
                ("reg_scale", C.c_float), ("precision", C.c_int32),
                ("variant", C.c_int32), ("hl", C.c_int32), ("ha", C.c_int32), ("hv", C.c_int32), ("mem_dim", C.c_int32),
                ("nn1", C.c_int32), ("nn2", C.c_int32), ("g1", C.c_int32), ("g2", C.c_int32),
                ("drop_nn1", C.c_float), ("drop_nn2", C.c_float), ("drop_g1", C.c_float), ("drop_g2", C.c_float)]


_SIGS = {
    "mfm_abi_version": (C.c_int, []),
    "mfm_last_error": (C.c_char_p, []),
    "mfm_device_cus": (C.c_int, []),
    "mfm_gemm_grouped_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_int, C.c_void_p]),
    "mfm_gemm_grouped_bf16": (C.c_int, [C.POINTER(GemmDesc), C.c_int, C.c_void_p]),
    "mfm_lstm_pack_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "mfm_lstm_pack_bf16": (C.c_int, [C.POINTER(SeqDesc), C.c_int, C.c_void_p]),
    "mfm_lstm_seq_fwd_bf16": (C.c_int, [C.POINTER(SeqDesc), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mfm_lstm_seq_bwd_bf16": (C.c_int, [C.POINTER(SeqDesc), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mfm_lstm_seq_fwd": (C.c_int, [C.POINTER(SeqDesc), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mfm_lstm_seq_bwd": (C.c_int, [C.POINTER(SeqDesc), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mfm_dw_bf16_lstm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_dw_f32_lstm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_mmd_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_mfn_mem_fwd": (C.c_int, [C.POINTER(MemDesc), C.c_void_p]),
    "mfm_mfn_mem_bwd": (C.c_int, [C.POINTER(MemDesc), C.c_void_p]),
    "mfm_mse_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float,
                                  C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_adam_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mfm_adam_flat_spans": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AdamSpan), C.c_int32,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mfm_adam_flat_guarded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "mfm_adam_flat_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "mfm_adam_flat_spans_guarded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AdamSpan), C.c_int32,
                                              C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "mfm_p2p_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]),
    "mfm_p2p_handle_bytes": (C.c_int, []),
    "mfm_p2p_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mfm_p2p_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mfm_p2p_local_base": (C.c_void_p, [C.c_void_p]),
    "mfm_p2p_connect_bases": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mfm_p2p_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mfm_p2p_allreduce_adam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_void_p]),
    "mfm_p2p_allreduce_adam_guarded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                                 C.c_int64, C.c_void_p]),
    "mfm_p2p_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mfm_p2p_wait_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "mfm_p2p_destroy": (None, [C.c_void_p]),
    "mfm_plan_num_params": (C.c_int, [C.c_int32]),
    "mfm_plan_set_gauss": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mfm_plan_create": (C.c_int, [C.POINTER(PlanConfig), C.POINTER(C.c_int64), C.c_int64,
                                  C.POINTER(C.c_void_p)]),
    "mfm_plan_destroy": (None, [C.c_void_p]),
    "mfm_plan_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "mfm_plan_debug_offset": (C.c_int64, [C.c_void_p]),
    "mfm_plan_init_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_plan_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "mfm_plan_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "mfm_plan_backward_ext": (C.c_int, [C.c_void_p] * 11),
    "mfm_plan_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_plan_backward_weighted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(LossWeights),
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_plan_out_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mfm_plan_host_status": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32))]),
    "mfm_plan_grad_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_plan_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_float,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_plan_train_step_staged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(AdamSpan),
                                             C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_plan_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "mfm_plan_set_option_str": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "mfm_plan_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "mfm_plan_state_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mfm_plan_clear_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfm_plan_latent_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mfm_plan_seq_layout": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "mfm_plan_mfn_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "mfm_plan_flops_per_step": (C.c_double, [C.c_void_p]),
    "mfm_plan_bytes_per_step": (C.c_double, [C.c_void_p]),
    "mfm_plan_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "mfm_plan_set_timing_every": (C.c_int, [C.c_void_p, C.c_int]),
    "mfm_plan_num_kernels": (C.c_int, []),
    "mfm_plan_kernel_name": (C.c_char_p, [C.c_int]),
    "mfm_plan_collect_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mfm_timing_bracket_overhead_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "mfm_plan_kernel_flops": (C.c_double, [C.c_void_p, C.c_int]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfmError(
                "factorized_amd: %s is missing. Build the gfx950 HIP library first:\n"
                "    python -c 'import __graft_entry__ as g; g.build()'   (or factorized_amd/csrc/build.sh)\n"
                "There is no CPU fallback for this package." % LIB_PATH)
        # torch first: its ROCm runtime (libamdhip64) must be the one this process initialises -- a process that loads this
        # library before torch ends up with two HIP runtimes and "no ROCm-capable device" in the second one
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)   # AttributeError if the ABI and the binding drift apart
            fn.restype = res
            fn.argtypes = args
        if L.mfm_abi_version() != ABI_VERSION:
            raise MfmError("libmfm_hip.so ABI version %d, binding expects %d" % (L.mfm_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mfm_last_error()
        raise MfmError("%s failed (%d): %s" % (what or "libmfm_hip call", rc,
                                               msg.decode() if msg else "?"))


def exported_names():
    return list(_SIGS)
