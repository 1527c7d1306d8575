"""Host side of the fused MFM_KL_EF step.

`MFMEngine` owns ONE flat fp32 parameter buffer (every tensor of the reference
model's `state_dict()` is a view into it, in the reference's order, each start
padded to 64 floats so kernels can use 16-byte loads), the matching gradient /
Adam-moment buffers, and per-(T,B) plans + workspaces of libmfm_hip.so.  One
Python call = one C call = the whole forward/backward/Adam chain enqueued on the
current HIP stream (reference: MFM_KL_EF.forward mfm_model.py:619-660 plus the
hot loop mfm_mosi.py:424-442).

PyTorch is used for device memory, streams and torch.distributed only.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

ALIGN = 64  # floats


def klef_param_shapes(cfg):
    """Ordered name -> shape of MFM_KL_EF's 78 parameters (reference mfm_model.py:579-617)."""
    d_l, d_a, d_v = cfg["input_dims"]
    zl, za, zv, zy = cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]
    fl, fa, fv, fy = cfg["fl_size"], cfg["fa_size"], cfg["fv_size"], cfg["fy_size"]
    od = cfg["output_dim"]
    s = OrderedDict()

    def lstm(prefix, d, h, out):
        s[prefix + ".lstm.weight_ih"] = (4 * h, d)
        s[prefix + ".lstm.weight_hh"] = (4 * h, h)
        s[prefix + ".lstm.bias_ih"] = (4 * h,)
        s[prefix + ".lstm.bias_hh"] = (4 * h,)
        s[prefix + ".fc1.weight"] = (out, h)
        s[prefix + ".fc1.bias"] = (out,)

    def lin(name, i, o):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    lstm("encoder_l", d_l, zl, zl)
    lstm("encoder_a", d_a, za, za)
    lstm("encoder_v", d_v, zv, zv)
    lstm("decoder_l", fy + fl, fy + fl, d_l)
    lstm("decoder_a", fy + fa, fy + fa, d_a)
    lstm("decoder_v", fy + fv, fy + fv, d_v)
    ze = zl + za + zv
    lstm("ef_encoder", d_l + d_a + d_v, ze, ze)
    lin("last_to_zy_fc1", ze, zy)
    lin("last_to_logvarzy_fc1", ze, zy)
    lin("last_to_zl_fc1", zl, zl)
    lin("last_to_za_fc1", za, za)
    lin("last_to_zv_fc1", zv, zv)
    lin("last_to_logvarzl_fc1", zl, zl)
    lin("last_to_logvarza_fc1", za, za)
    lin("last_to_logvarzv_fc1", zv, zv)
    for tag, zi, fo in (("zy_to_fy", zy, fy), ("zl_to_fl", zl, fl), ("za_to_fa", za, fa), ("zv_to_fv", zv, fv)):
        lin(tag + "_fc1", zi, fo)
        lin(tag + "_fc2", fo, fo)
    lin("fy_to_y_fc1", fy, fy)
    lin("fy_to_y_fc2", fy, od)
    assert len(s) == _lib.MFM_KLEF_NPARAM
    return s


VARIANTS = {"kl_ef": 0, "kl": 1, "mmd": 2}


def mfn_param_shapes(configs, variant):
    """Ordered name -> shape of MFM_KL ('kl', 104 tensors, reference mfm_model.py:662-721) or MFM ('mmd', 90 tensors,
    :469-520), both with the Memory Fusion Network encoder (:93-138; out_fc1 / out_fc2 exist, forward never uses
    them)."""
    cfg, nn1, nn2, g1, g2, outc = configs
    d_l, d_a, d_v = cfg["input_dims"]
    hl, ha, hv = cfg["h_dims"]
    zl, za, zv, zy = cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]
    fl, fa, fv, fy = cfg["fl_size"], cfg["fa_size"], cfg["fv_size"], cfg["fy_size"]
    od, M = cfg["output_dim"], cfg["memsize"]
    assert cfg.get("windowsize", 2) == 2, "the fused MFN plan is built for windowsize 2 (cStar = [c_{t-1}, c_t])"
    tot = hl + ha + hv
    A2 = 2 * tot
    s = OrderedDict()

    def cell(prefix, d, h):
        s[prefix + ".weight_ih"] = (4 * h, d)
        s[prefix + ".weight_hh"] = (4 * h, h)
        s[prefix + ".bias_ih"] = (4 * h,)
        s[prefix + ".bias_hh"] = (4 * h,)

    def lstm(prefix, d, h, out):
        cell(prefix + ".lstm", d, h)
        s[prefix + ".fc1.weight"] = (out, h)
        s[prefix + ".fc1.bias"] = (out,)

    def lin(name, i, o):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    lstm("encoder_l", d_l, zl, zl)
    lstm("encoder_a", d_a, za, za)
    lstm("encoder_v", d_v, zv, zv)
    lstm("decoder_l", fy + fl, fy + fl, d_l)
    lstm("decoder_a", fy + fa, fy + fa, d_a)
    lstm("decoder_v", fy + fv, fy + fv, d_v)
    cell("mfn_encoder.lstm_l", d_l, hl)
    cell("mfn_encoder.lstm_a", d_a, ha)
    cell("mfn_encoder.lstm_v", d_v, hv)
    lin("mfn_encoder.att1_fc1", A2, nn1["shapes"])
    lin("mfn_encoder.att1_fc2", nn1["shapes"], A2)
    lin("mfn_encoder.att2_fc1", A2, nn2["shapes"])
    lin("mfn_encoder.att2_fc2", nn2["shapes"], M)
    lin("mfn_encoder.gamma1_fc1", A2 + M, g1["shapes"])
    lin("mfn_encoder.gamma1_fc2", g1["shapes"], M)
    lin("mfn_encoder.gamma2_fc1", A2 + M, g2["shapes"])
    lin("mfn_encoder.gamma2_fc2", g2["shapes"], M)
    lin("mfn_encoder.out_fc1", tot + M, outc["shapes"])
    lin("mfn_encoder.out_fc2", outc["shapes"], od)
    lin("last_to_zy_fc1", tot + M, zy)
    if variant == "kl":
        lin("last_to_logvarzy_fc1", tot + M, zy)
        lin("last_to_zl_fc1", zl, zl)
        lin("last_to_za_fc1", za, za)
        lin("last_to_zv_fc1", zv, zv)
        lin("last_to_logvarzl_fc1", zl, zl)
        lin("last_to_logvarza_fc1", za, za)
        lin("last_to_logvarzv_fc1", zv, zv)
    for tag, zi, fo in (("zy_to_fy", zy, fy), ("zl_to_fl", zl, fl), ("za_to_fa", za, fa), ("zv_to_fv", zv, fv)):
        lin(tag + "_fc1", zi, fo)
        lin(tag + "_fc2", fo, fo)
    lin("fy_to_y_fc1", fy, fy)
    lin("fy_to_y_fc2", fy, od)
    return s


def param_shapes(configs, variant="kl_ef"):
    return klef_param_shapes(configs[0]) if variant == "kl_ef" else mfn_param_shapes(configs, variant)


# Latent-stack tensors grouped by the stage of the fused latent kernel that consumes them
# (csrc/latent.hip): each group is laid out contiguously so a stage's weights are ONE linear
# global->LDS copy.  Everything else (the LSTM tensors) keeps state_dict order in front.
_LATENT_STAGE_KEYS_MFN = {
    "kl": [
        ("encoder_l.fc1", "encoder_a.fc1", "encoder_v.fc1"),
        ("last_to_zl_fc1", "last_to_za_fc1", "last_to_zv_fc1"),
        ("zl_to_fl_fc1", "za_to_fa_fc1", "zv_to_fv_fc1", "zy_to_fy_fc1"),
        ("zl_to_fl_fc2", "za_to_fa_fc2", "zv_to_fv_fc2", "zy_to_fy_fc2"),
        ("fy_to_y_fc1", "last_to_logvarzl_fc1", "last_to_logvarza_fc1", "last_to_logvarzv_fc1"),
        ("fy_to_y_fc2",),
    ],
    "mmd": [
        ("encoder_l.fc1", "encoder_a.fc1", "encoder_v.fc1"),
        ("zl_to_fl_fc1", "za_to_fa_fc1", "zv_to_fv_fc1", "zy_to_fy_fc1"),
        ("zl_to_fl_fc2", "za_to_fa_fc2", "zv_to_fv_fc2", "zy_to_fy_fc2"),
        ("fy_to_y_fc1",),
        ("fy_to_y_fc2",),
    ],
}

_LATENT_STAGE_KEYS = [
    ("encoder_l.fc1", "encoder_a.fc1", "encoder_v.fc1", "ef_encoder.fc1"),
    ("last_to_zl_fc1", "last_to_za_fc1", "last_to_zv_fc1", "last_to_zy_fc1"),
    ("zl_to_fl_fc1", "za_to_fa_fc1", "zv_to_fv_fc1", "zy_to_fy_fc1"),
    ("zl_to_fl_fc2", "za_to_fa_fc2", "zv_to_fv_fc2", "zy_to_fy_fc2"),
    ("fy_to_y_fc1", "last_to_logvarzl_fc1", "last_to_logvarza_fc1", "last_to_logvarzv_fc1", "last_to_logvarzy_fc1"),
    ("fy_to_y_fc2",),
]


def _staged_group(name):
    """Which stage losses of train_beta_vae (reference mfm_mosi.py:278-281) reach a tensor: 'disc' = only the
    discriminative term (the classifier), 'gen' = only the reconstruction terms (the three decoders and the
    modality z->f MLPs), 'shared' = every stage (everything on the way to the KLD and to f_y)."""
    if name.startswith("fy_to_y_"):
        return "disc"
    if name.startswith("decoder_") or name.startswith(("zl_to_fl_", "za_to_fa_", "zv_to_fv_")):
        return "gen"
    return "shared"


class FlatLayout:
    """Placement of the named tensors in one flat fp32 buffer.  `shapes`/`offsets` iterate in the
    reference's state_dict order (what the C plan expects); the PHYSICAL order is chosen here."""

    def __init__(self, shapes, variant="kl_ef"):
        self.shapes = OrderedDict(shapes)
        stage_of = {}
        stage_keys = _LATENT_STAGE_KEYS if variant == "kl_ef" else _LATENT_STAGE_KEYS_MFN[variant]
        for st, prefixes in enumerate(stage_keys):
            for pre in prefixes:
                stage_of[pre + ".weight"] = st
                stage_of[pre + ".bias"] = st
        order = [n for n in self.shapes if n not in stage_of]
        for st in range(len(stage_keys)):
            order += [n for n in self.shapes if stage_of.get(n) == st]
        assert sorted(order) == sorted(self.shapes)
        placed = {}
        cur = 0
        for name in order:
            placed[name] = cur
            n = int(np.prod(self.shapes[name]))
            cur = (cur + n + ALIGN - 1) // ALIGN * ALIGN
        self.offsets = OrderedDict((n, placed[n]) for n in self.shapes)
        # (offset, numel, shape) in state_dict order: per-tensor views of any buffer with this layout
        self.slots = [(placed[n], int(np.prod(self.shapes[n])), tuple(self.shapes[n])) for n in self.shapes]
        # one spare granule behind the last tensor: element `guard` of a GRADIENT buffer with this layout is the guard word of
        # the guarded Adam launches (include/mfm_hip.h): the fused plan stores a NaN there when an in-launch hand-over of the
        # step gave up, the optimizer then leaves the parameters alone; being part of the buffer it rides through the
        # data-parallel all-reduce, so every rank takes the same decision
        self.guard = cur
        cur += ALIGN
        self.total = cur
        self.numel = sum(int(np.prod(s)) for s in self.shapes.values())
        # contiguous runs of tensors that belong to the same staged-training group, in physical order:
        # [(group, begin, end)] with 64-float aligned bounds (Adam over spans, mfm_adam_flat_spans)
        self.group_spans = []
        ends = [placed[n] for n in order[1:]] + [self.guard]
        for name, end in zip(order, ends):
            g = _staged_group(name)
            if self.group_spans and self.group_spans[-1][0] == g:
                self.group_spans[-1] = (g, self.group_spans[-1][1], end)
            else:
                self.group_spans.append((g, placed[name], end))

    def views(self, flat):
        out = OrderedDict()
        for name, shp in self.shapes.items():
            o = self.offsets[name]
            out[name] = flat[o:o + int(np.prod(shp))].view(*shp)
        return out


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    # raw handle of the current stream of the current device: torch.cuda.current_stream() builds a Stream object through
    # three layers of device-index helpers (~10-20 us of host time per call; measured in the unchanged-loop profile)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class _Plan:
    def __init__(self, engine, T, B, reg_scale):
        cfg = engine.cfg
        pc = _lib.PlanConfig()
        pc.d_l, pc.d_a, pc.d_v = cfg["input_dims"]
        pc.zl, pc.za, pc.zv, pc.zy = cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]
        pc.fl, pc.fa, pc.fv, pc.fy = cfg["fl_size"], cfg["fa_size"], cfg["fv_size"], cfg["fy_size"]
        pc.output_dim = cfg["output_dim"]
        pc.loss_kind = 1 if cfg.get("loss", "l1") == "ce" else 0
        pc.T, pc.B = T, B
        pc.lda_xl, pc.lda_xa, pc.lda_xv = cfg["lda_xl"], cfg["lda_xa"], cfg["lda_xv"]
        pc.lda_reg = cfg["lda_mmd"]
        pc.drop_zy, pc.drop_zl = cfg["zy_to_fy_dropout"], cfg["zl_to_fl_dropout"]
        pc.drop_za, pc.drop_zv = cfg["za_to_fa_dropout"], cfg["zv_to_fv_dropout"]
        pc.drop_y = cfg["fy_to_y_dropout"]
        pc.reg_scale = reg_scale
        pc.precision = 1 if engine.precision == "bf16" else 0
        pc.variant = VARIANTS[engine.variant]
        if engine.variant != "kl_ef":
            nn1, nn2, g1, g2 = engine.configs[1:5]
            pc.hl, pc.ha, pc.hv = cfg["h_dims"]
            pc.mem_dim = cfg["memsize"]
            pc.nn1, pc.nn2, pc.g1, pc.g2 = nn1["shapes"], nn2["shapes"], g1["shapes"], g2["shapes"]
            pc.drop_nn1, pc.drop_nn2, pc.drop_g1, pc.drop_g2 = nn1["drop"], nn2["drop"], g1["drop"], g2["drop"]
        nparam = _lib.lib().mfm_plan_num_params(pc.variant)
        assert nparam == len(engine.layout.offsets), (nparam, len(engine.layout.offsets))
        offs = (C.c_int64 * nparam)(*engine.layout.offsets.values())
        handle = C.c_void_p(0)
        _lib.check(_lib.lib().mfm_plan_create(C.byref(pc), offs, engine.layout.total, C.byref(handle)),
                   "mfm_plan_create")
        self.handle = handle
        self._pid = os.getpid()          # a forked child (DataLoader worker, multiprocessing.Manager) must not tear the plan down
        self.T, self.B = T, B
        nbytes = _lib.lib().mfm_plan_workspace_bytes(handle)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=engine.device)
        _lib.check(_lib.lib().mfm_plan_init_workspace(handle, _ptr(self.workspace), _stream()),
                   "mfm_plan_init_workspace")
        self.set_option("grad_guard_offset", engine.layout.guard)
        self.set_option("handover", 1 if engine.handover else 0)
        if engine.handover_timeout_us is not None:
            self.set_option("handover_timeout_us", int(engine.handover_timeout_us))
        # the plan's device-side state lives in its workspace (mfm_plan_state_layout): the loss slots and, right behind them,
        # the sticky status word of the in-launch hand-overs -- one 64-byte block, one copy when the host looks at the losses
        lay = (C.c_int64 * 8)()
        _lib.check(_lib.lib().mfm_plan_state_layout(handle, lay), "mfm_plan_state_layout")
        assert lay[1] == lay[0] + 4 * _lib.MFM_LOSS_SLOTS
        wf = self.workspace.view(torch.float32)
        self.state = wf[lay[0] // 4: lay[0] // 4 + 16]
        self.losses = self.state[:_lib.MFM_LOSS_SLOTS]
        self.tick = self.workspace[lay[2]: lay[2] + 8].view(torch.int64)          # replay counters (captured steps)
        self.dw_tick = self.workspace[lay[3]: lay[3] + 4].view(torch.int32)
        self.fwd_serial = 0        # forwards run on this workspace so far (autograd path: whose activations it holds)
        self.consumed = False      # the last forward's activations were overwritten by a backward
        # where the last forward left x_hat_l/a/v and y_hat inside the workspace (mfm_plan_out_layout): what the module path's
        # lazy outputs are views of (factorized_amd/lazy.py); None on bf16-resident plans, which do not keep x_hat
        ol = (C.c_int64 * 8)()
        _lib.check(_lib.lib().mfm_plan_out_layout(handle, ol), "mfm_plan_out_layout")
        self.out_views = None
        if min(ol[0], ol[1], ol[2], ol[3]) >= 0:
            dd = list(cfg["input_dims"]) + [cfg["output_dim"]]
            shp = [(T, B, dd[0]), (T, B, dd[1]), (T, B, dd[2]), (B, dd[3])]
            self.out_views = tuple(wf[ol[i] // 4: ol[i] // 4 + int(np.prod(shp[i]))].view(shp[i]) for i in range(4))
        self.loss0d = self.state[7]          # a spare word: the storage behind the symbolic loss expressions (never read)
        # host-coherent copy of "a hand-over gave up" (mfm_plan_host_status): polled for free by the optimizers
        hp = C.POINTER(C.c_uint32)()
        _lib.check(_lib.lib().mfm_plan_host_status(handle, C.byref(hp)), "mfm_plan_host_status")
        self.host_words = (C.c_uint32 * 2).from_address(C.addressof(hp.contents)) if hp else None
        self.handover_now = bool(engine.handover)

    def ensure_handover(self, on):
        """the module path switches the in-launch hand-overs per call: only a step whose optimizer honours the gradient guard
        may use them (factorized_amd.optim.Adam does, torch.optim.Adam does not)"""
        on = bool(on)
        if self.handover_now != on:
            self.set_option("handover", 1 if on else 0)
            self.handover_now = on

    def set_option(self, key, value):
        _lib.check(_lib.lib().mfm_plan_set_option(self.handle, key.encode(), int(value)), "mfm_plan_set_option(%s)" % key)

    def set_switch(self, name, value):
        """one of the plan's MFM_* tuning / test switches (a string, or None to remove it): the plan reads its OWN table, filled
        from the environment when it was created -- never the environment itself -- on every launch"""
        v = None if value is None else str(value).encode()
        _lib.check(_lib.lib().mfm_plan_set_option_str(self.handle, name.encode(), v), "mfm_plan_set_option_str(%s)" % name)

    def get_option(self, key):
        v = C.c_int64(0)
        _lib.check(_lib.lib().mfm_plan_get_option(self.handle, key.encode(), C.byref(v)), "mfm_plan_get_option(%s)" % key)
        return int(v.value)

    def __del__(self):
        # Only the process that created the plan destroys it: in a forked child the HIP runtime the handle points into is not
        # usable (a garbage collection there used to end the child with a segmentation fault inside mfm_plan_destroy --
        # seen as an EOFError of multiprocessing.Manager() in a process that holds live engines).
        try:
            if self.handle and getattr(self, "_pid", None) == os.getpid():
                _lib.lib().mfm_plan_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


class MFMEngine:
    """The fused training / inference step of MFM_KL_EF (variant "kl_ef"), MFM_KL ("kl") or MFM ("mmd") on one MI355X.
    `configs` is the reference's six-dict list."""

    def __init__(self, configs, device="cuda", reg_scale=1.0, precision="fp32", variant="kl_ef"):
        if not torch.cuda.is_available():
            raise _lib.MfmError("MFMEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                                "there is no CPU fallback")
        _lib.lib()
        self.configs = configs
        self.cfg = configs[0]
        self.device = torch.device(device)
        if variant not in VARIANTS:
            raise ValueError("variant must be one of %s" % sorted(VARIANTS))
        self.variant = variant
        self.layout = FlatLayout(param_shapes(configs, variant), variant)
        self.gauss = None            # variant "mmd": fixed N(0,1) sample [B, zl+za+zv+zy] (parity runs); None = draw per call
        n = self.layout.total
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.step_count = 0
        self.reg_scale = float(reg_scale)
        # "fp32": the reference's arithmetic (1e-4 parity).  "bf16": bf16 MFMA operands in every GEMM and recurrence,
        # fp32 accumulation / master weights / Adam / cell state / latent stack / losses (BASELINE configs 2-4)
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision
        self.seed = 1234
        self._plans = {}
        # In-launch hand-overs of the B <= 32 step (role workgroups, csrc/proj_role_dev.h / dw_role_dev.h): on by default; they
        # need the launch to have the GPU to itself.  A consumer that gives up waiting raises the plan's status word and poisons
        # the gradient guard (the optimizer skips the step); check_status() / loss_dict() then raise and switch every plan of
        # this engine to the separate launches for the rest of the run.
        self.handover = os.environ.get("MFM_SHARED_DEVICE", "0") in ("", "0")
        self.handover_timeout_us = None      # None = the library's default (50 ms)
        self.handover_failures = 0
        self._status_message = ""
        # Staged training (train_beta_vae): Adam step counters per tensor group.  "frozen" = torch >= 2 semantics
        # (zero_grad sets .grad to None, Adam skips such parameters: a group without a gradient in the current
        # stage keeps its values and moments); "legacy" = torch 0.4 semantics (zero_grad leaves zero tensors
        # behind once a gradient has existed: such a group keeps moving on its decaying momentum).  See DESIGN.md.
        self.staged_adam = "frozen"
        self.group_steps = {"shared": 0, "gen": 0, "disc": 0}

    # ------------------------------------------------------------------ parameters
    def param_views(self):
        return self.layout.views(self.params)

    def grad_views(self):
        return self.layout.views(self.grads)

    def load_weights(self, weights):
        """weights: ordered mapping name -> ndarray/tensor with the reference's state_dict keys."""
        assert list(weights.keys()) == list(self.layout.shapes.keys()), "state_dict keys differ"
        host = np.zeros(self.layout.total, dtype=np.float32)
        for name, w in weights.items():
            a = w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
            assert tuple(a.shape) == tuple(self.layout.shapes[name]), name
            o = self.layout.offsets[name]
            host[o:o + a.size] = a.ravel()
        self.params.copy_(torch.from_numpy(host))
        self.adam_m.zero_(); self.adam_v.zero_(); self.step_count = 0
        self.group_steps = {"shared": 0, "gen": 0, "disc": 0}

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.param_views().items())

    # ------------------------------------------------------------------ plans
    def plan(self, T, B):
        key = (int(T), int(B), self.reg_scale, self.precision, self.variant)
        p = self._plans.get(key)
        if p is None:
            p = _Plan(self, int(T), int(B), self.reg_scale)
            self._plans[key] = p
        return p

    def set_handover(self, on):
        """Allow / forbid the in-launch hand-overs on every plan of this engine (existing and future ones)."""
        self.handover = bool(on)
        for p in self._plans.values():
            p.set_option("handover", 1 if on else 0)
            p.handover_now = bool(on)

    def poll_status(self):
        """Did a hand-over of any plan give up?  A read of host memory (mfm_plan_host_status): no copy, no synchronisation --
        cheap enough for every optimizer step.  Non-zero: call check_status()."""
        for p in self._plans.values():
            hw = p.host_words
            if hw is not None and (hw[0] | hw[1]):
                return True
        return False

    def status_message(self):
        return self._status_message

    def check_status(self, state_host=None, raise_on_error=True):
        """Look at the status words of this engine's plans (synchronises unless `state_host` -- a host copy of one plan's
        state block, (plan, ndarray) -- is given).  A non-zero word means a hand-over inside a launch gave up waiting (another
        process / stream kept its producers off the GPU): the steps since then were skipped by the optimizer.  The word is
        cleared, the hand-overs are switched off for the rest of the run (separate launches: nothing can wait any more) and
        MfmError is raised; the caller may catch it and carry on training."""
        bad = 0
        items = [state_host] if state_host is not None else [(p, None) for p in self._plans.values()]
        for p, host in items:
            if host is None:
                host = p.state.detach().cpu().numpy()
            word = int(host[_lib.MFM_LOSS_SLOTS:_lib.MFM_LOSS_SLOTS + 1].view(np.uint32)[0])
            if word:
                bad |= word
                _lib.check(_lib.lib().mfm_plan_clear_status(p.handle, _ptr(p.workspace), _stream()), "mfm_plan_clear_status")
        if bad:
            self.handover_failures += 1
            self.set_handover(False)
            self._status_message = (
                "an in-launch hand-over of the fused step gave up waiting (status 0x%x: %s): other work on this GPU kept the "
                "producer workgroups off the device.  The affected steps were NOT applied to the parameters; the engine now "
                "uses separate launches (set_handover(True) re-enables the role workgroups)."
                % (bad, " + ".join(n for b, n in ((1, "projections"), (2, "weight gradients")) if bad & b)))
            if raise_on_error:
                raise _lib.MfmError(self._status_message)
        return bad

    def _gauss_for(self, p, B):
        """variant "mmd": hand the plan the N(0,1) sample loss_MMD draws per forward (reference mfm_model.py:26)."""
        if self.variant != "mmd":
            return
        c = self.cfg
        gl = c["zl_size"] + c["za_size"] + c["zv_size"] + c["zy_size"]
        g = self.gauss
        if g is None:
            g = torch.randn(B, gl, device=self.device)
        assert g.shape == (B, gl) and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()
        p.gauss_ref = g              # keep it alive until the next call
        _lib.check(_lib.lib().mfm_plan_set_gauss(p.handle, _ptr(g)), "mfm_plan_set_gauss")

    def _check_inputs(self, x, y):
        """the kernels take raw pointers and index the batch with the plan's sizes: anything that does not fit is refused here
        (exceptions, not asserts: they must survive python -O)"""
        D = sum(self.cfg["input_dims"])
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()):
            raise _lib.MfmError("batch must be a contiguous float32 CUDA tensor [T, B, %d]; got %s" % (
                D, "%s %s %s" % (tuple(x.shape), x.dtype, x.device) if torch.is_tensor(x) else type(x)))
        if x.shape[2] != D:
            raise _lib.MfmError("batch has %d features per time step, the model's input_dims sum to %d" % (x.shape[2], D))
        if self.device.index is not None and x.device != self.device:
            raise _lib.MfmError("batch lives on %s, the engine on %s" % (x.device, self.device))
        if y is not None:
            want = torch.int64 if self.cfg.get("loss", "l1") == "ce" else torch.float32
            if not (y.is_cuda and y.device == x.device and y.is_contiguous() and y.shape[0] == x.shape[1] and y.dtype == want):
                raise _lib.MfmError("labels must be a contiguous %s tensor on %s with %d rows; got %s %s %s" % (
                    want, x.device, x.shape[1], tuple(y.shape), y.dtype, y.device))

    # ------------------------------------------------------------------ the three entry points
    def forward(self, x, y=None, train=False, want_xhat=True, handover=None):
        """x [T,B,D] -> dict(x_l_hat, x_a_hat, x_v_hat, y_hat, losses[8] (device tensor)).  `handover`: override of the engine's
        hand-over switch for this call (the module path: only under a guard-aware optimizer)."""
        self._check_inputs(x, y)
        T, B, _ = x.shape
        p = self.plan(T, B)
        p.ensure_handover(self.handover if handover is None else handover)
        d_l, d_a, d_v = self.cfg["input_dims"]
        out = {}
        xh = [None, None, None]
        if want_xhat:
            xh = [torch.empty(T, B, d, dtype=torch.float32, device=self.device) for d in (d_l, d_a, d_v)]
        y_hat = torch.empty(B, self.cfg["output_dim"], dtype=torch.float32, device=self.device)
        p.fwd_serial += 1
        p.consumed = False
        self._gauss_for(p, B)
        _lib.check(_lib.lib().mfm_plan_forward(p.handle, _ptr(self.params), _ptr(x), _ptr(y), int(bool(train)),
                                               C.c_uint64(self.seed), _ptr(p.workspace), _ptr(xh[0]), _ptr(xh[1]),
                                               _ptr(xh[2]), _ptr(y_hat), _ptr(p.losses), _stream()),
                   "mfm_plan_forward")
        out["x_l_hat"], out["x_a_hat"], out["x_v_hat"] = xh
        out["y_hat"] = y_hat
        out["losses"] = p.losses
        return out

    def forward_train(self, x, p, grads_to_zero=None):
        """Training-mode forward of the module path's lazy losses (mfm_plan_forward_train): no labels, no output tensors --
        x_hat / y_hat stay in the plan's workspace (`p.out_views`), the loss slots 1..4 are filled; `grads_to_zero`: a flat
        gradient buffer cleared inside the first launch."""
        self._check_inputs(x, None)
        p.fwd_serial += 1
        p.consumed = False
        self._gauss_for(p, x.shape[1])
        _lib.check(_lib.lib().mfm_plan_forward_train(p.handle, _ptr(self.params), _ptr(x), C.c_uint64(self.seed), _ptr(p.workspace),
                                                     _ptr(grads_to_zero), _stream()), "mfm_plan_forward_train")

    def backward_weighted(self, x, y, w, p, out=None):
        """Backward of  w.disc * L_disc(y_hat, y) + sum_m w.gen_m MSE_m + w.reg * reg  on the last forward of plan `p`
        (mfm_plan_backward_weighted; w: _lib.LossWeights).  `out`: flat buffer to receive the gradients (default self.grads)."""
        g = self.grads if out is None else out
        _lib.check(_lib.lib().mfm_plan_backward_weighted(p.handle, _ptr(self.params), _ptr(x), _ptr(y), C.byref(w), _ptr(p.workspace),
                                                         _ptr(g), _stream()), "mfm_plan_backward_weighted")
        return g

    def backward(self, x, y, stage=0):
        """Backward of the last forward() with the same (T,B); fills self.grads."""
        self._check_inputs(x, y)
        T, B, _ = x.shape
        p = self.plan(T, B)
        p.ensure_handover(self.handover)
        _lib.check(_lib.lib().mfm_plan_backward(p.handle, _ptr(self.params), _ptr(x), _ptr(y), int(stage),
                                                _ptr(p.workspace), _ptr(self.grads), _stream()),
                   "mfm_plan_backward")
        return self.grads

    def backward_ext(self, x, d_xl, d_xa, d_xv, d_yhat, d_reg, out=None):
        """Backward of the last forward() for arbitrary upstream gradients (autograd module path).  `out`: another flat
        buffer with this engine's layout to receive the gradients instead of self.grads (it is overwritten)."""
        T, B, _ = x.shape
        p = self.plan(T, B)
        g = self.grads if out is None else out
        assert g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == self.layout.total
        _lib.check(_lib.lib().mfm_plan_backward_ext(p.handle, _ptr(self.params), _ptr(x), _ptr(d_xl), _ptr(d_xa),
                                                    _ptr(d_xv), _ptr(d_yhat), _ptr(d_reg), _ptr(p.workspace),
                                                    _ptr(g), _stream()), "mfm_plan_backward_ext")
        return g

    def _staged_spans(self, stage):
        """Advance the per-group Adam step counters for one step of `stage` and return the spans to update."""
        has_grad = {"shared": True, "gen": stage != 2, "disc": stage != 1}
        gs = self.group_steps
        active = {}
        for g in gs:
            if has_grad[g] or (self.staged_adam == "legacy" and gs[g] > 0):
                gs[g] += 1
                active[g] = gs[g]
        spans = [(b, e, active[g]) for g, b, e in self.layout.group_spans if g in active]
        merged = []
        for b, e, st in spans:                       # neighbours with the same step count become one span
            if merged and merged[-1][1] == b and merged[-1][2] == st:
                merged[-1] = (merged[-1][0], e, st)
            else:
                merged.append((b, e, st))
        return merged

    def train_step(self, x, y, lr=1e-3, grad_scale=1.0, check=True, stage=0):
        """forward(train) + backward + Adam, one enqueue.  stage 0 = the joint loss of train_mfm
        (mfm_mosi.py:439); 1 = gen + reg, 2 = disc + reg (train_beta_vae, mfm_mosi.py:278-281), with Adam
        skipping the tensors the stage loss does not reach (`staged_adam`).  Returns the device tensor of loss
        slots (no sync).

        Caveat on staged runs: the default `staged_adam = "frozen"` reproduces the reference code on torch >= 2
        (`zero_grad()` sets .grad to None, Adam skips the parameter).  The reference was written for PyTorch 0.4,
        whose `zero_grad()` left zero tensors behind: there the decoders and the modality z->f MLPs keep moving in
        stage 2 on their decaying first moment and their step counters keep advancing.  `staged_adam = "legacy"`
        (driver flag --legacy-adam) is that behaviour; both are pinned to trajectories of the reference itself
        (tests/golden/*_staged_*.npz: `frozen_*` from zero_grad(set_to_none=True), `legacy_*` from set_to_none=False)."""
        if check:
            self._check_inputs(x, y)
        T, B, _ = x.shape
        p = self.plan(T, B)
        p.ensure_handover(self.handover)
        self._gauss_for(p, B)
        gs = self.group_steps
        if stage != 0 or not (gs["shared"] == gs["gen"] == gs["disc"]):
            spans = self._staged_spans(stage)
            self.step_count = gs["shared"]
            if len(spans) > _lib.MFM_ADAM_MAX_SPANS:
                raise _lib.MfmError("staged Adam: %d spans (max %d)" % (len(spans), _lib.MFM_ADAM_MAX_SPANS))
            arr = (_lib.AdamSpan * len(spans))()
            for i, (b, e_, st) in enumerate(spans):
                arr[i].begin, arr[i].end, arr[i].step = b, e_, st
            p.fwd_serial += 1
            p.consumed = True
            _lib.check(_lib.lib().mfm_plan_train_step_staged(
                p.handle, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v), _ptr(x), _ptr(y),
                C.c_uint64(self.seed), int(stage), arr, len(spans), lr, grad_scale, _ptr(p.workspace), _ptr(p.losses),
                _stream()), "mfm_plan_train_step_staged")
            return p.losses
        self.step_count += 1
        for g in gs:
            gs[g] = self.step_count
        p.fwd_serial += 1
        p.consumed = True
        _lib.check(_lib.lib().mfm_plan_train_step(p.handle, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m),
                                                  _ptr(self.adam_v), _ptr(x), _ptr(y), C.c_uint64(self.seed),
                                                  self.step_count, lr, grad_scale, _ptr(p.workspace),
                                                  _ptr(p.losses), _stream()),
                   "mfm_plan_train_step")
        return p.losses

    def grad_step(self, x, y, check=True):
        """forward(train) + backward(joint loss), one enqueue, no optimizer (data-parallel step:
        all-reduce self.grads, then adam()).  Returns the device tensor of loss slots (no sync)."""
        if check:
            self._check_inputs(x, y)
        T, B, _ = x.shape
        p = self.plan(T, B)
        p.ensure_handover(self.handover)
        self._gauss_for(p, B)
        p.fwd_serial += 1
        p.consumed = True
        _lib.check(_lib.lib().mfm_plan_grad_step(p.handle, _ptr(self.params), _ptr(self.grads), _ptr(x), _ptr(y),
                                                 C.c_uint64(self.seed), _ptr(p.workspace), _ptr(p.losses), _stream()),
                   "mfm_plan_grad_step")
        return p.losses

    def _require_uniform_steps(self, what):
        """The flat Adam launches use ONE bias-correction step for every tensor.  After staged steps the per-group
        counters differ (torch.optim.Adam counts per parameter): continuing with a flat update would silently jump
        the gen / disc groups' step, so it is refused -- staged training goes through train_step(stage=...)."""
        gs = self.group_steps
        if not (gs["shared"] == gs["gen"] == gs["disc"] == self.step_count):
            raise _lib.MfmError("%s: the Adam step counters of the tensor groups differ (%s, step_count %d) after "
                                "staged training; the flat / data-parallel Adam update has one counter -- use "
                                "train_step(stage=...)" % (what, dict(gs), self.step_count))

    def adam(self, lr=1e-3, grad_scale=1.0):
        self._require_uniform_steps("MFMEngine.adam")
        self.step_count += 1
        for g in self.group_steps:
            self.group_steps[g] = self.step_count
        guard = C.c_void_p(self.grads.data_ptr() + 4 * self.layout.guard)
        _lib.check(_lib.lib().mfm_adam_flat_guarded(_ptr(self.params), _ptr(self.grads), _ptr(self.adam_m),
                                                    _ptr(self.adam_v), self.layout.total, self.step_count, lr,
                                                    0.9, 0.999, 1e-8, grad_scale, guard, _stream()), "mfm_adam_flat_guarded")

    def latent_record(self, T, B):
        """(activation record, gradient record, layout dict) of the latent stack for the plan at (T,B): views into
        the plan workspace [B, rec] plus record offsets (mfm_plan_latent_layout).  Tests / tuning aids."""
        p = self.plan(T, B)
        out = (C.c_int64 * 32)()
        _lib.check(_lib.lib().mfm_plan_latent_layout(p.handle, out), "mfm_plan_latent_layout")
        rs = int(out[2])
        ws = p.workspace.view(torch.float32)
        rec = ws[out[0] // 4: out[0] // 4 + B * rs].view(B, rs)
        grd = ws[out[1] // 4: out[1] // 4 + B * rs].view(B, rs)
        sites = ("zl_to_fl", "za_to_fa", "zv_to_fv", "zy_to_fy", "fy_to_y")
        lay = dict(rec_size=rs, row_path=bool(out[22]),
                   mask={s: int(out[3 + i]) for i, s in enumerate(sites)},
                   act={s: int(out[8 + i]) for i, s in enumerate(sites)},
                   width={s: int(out[13 + min(i, 3)]) for i, s in enumerate(sites)},
                   f={s: int(out[17 + i]) for i, s in enumerate(("l", "a", "v", "y"))},
                   mu={s: int(out[23 + i]) for i, s in enumerate(("l", "a", "v", "y"))},
                   z_n={s: int(out[27 + i]) for i, s in enumerate(("l", "a", "v", "y"))},
                   y_hat=int(out[21]))
        lay["width"]["fy_to_y"] = int(out[16])
        return rec, grd, lay

    def seq_buffers(self, T, B, which):
        """Saved activations of LSTM `which` (encoders in plan order, then the three decoders) as views of the plan workspace
        (mfm_plan_seq_layout): gates [T,B,4,Hp], hs [T,B,Hp] (bf16 tensors on a bf16-resident plan), cs (fp32); decoders also
        dhs [T,B,Hp] and dxhat [T*B, ld]; encoders h_last [B,Hp] or None.  Tests / tuning aids."""
        p = self.plan(T, B)
        out = (C.c_int64 * 12)()
        _lib.check(_lib.lib().mfm_plan_seq_layout(p.handle, int(which), out), "mfm_plan_seq_layout")
        h, Hp, st16, dec = int(out[3]), int(out[4]), bool(out[5]), bool(out[6])
        ws = p.workspace

        def v(off, shape, half):
            n = int(np.prod(shape))
            if half:
                return ws[off: off + 2 * n].view(torch.bfloat16).view(*shape)
            return ws[off: off + 4 * n].view(torch.float32).view(*shape)
        res = dict(h=h, Hp=Hp, bf16_resident=st16, bf16_recurrence=bool(out[11]),
                   gates=v(out[0], (T, B, 4, Hp), st16), hs=v(out[1], (T, B, Hp), st16), cs=v(out[2], (T, B, Hp), False))
        if dec:
            res["dhs"] = v(out[7], (T, B, Hp), st16)
            res["dxhat"] = v(out[8], (T * B, int(out[9])), st16)
            res["d"] = int(out[10])
        else:
            res["h_last"] = v(out[7], (B, Hp), False) if out[7] >= 0 else None
        return res

    def mfn_buffers(self, T, B):
        """variants "kl" / "mmd": views of the MFN's [T*B, .] workspace tensors (mfm_plan_mfn_layout).  Tests / tuning."""
        p = self.plan(T, B)
        out = (C.c_int64 * 16)()
        _lib.check(_lib.lib().mfm_plan_mfn_layout(p.handle, out), "mfm_plan_mfn_layout")
        ws = p.workspace.view(torch.float32)
        TB, A2, n1, n2, M, nzy = int(out[15]), int(out[10]), int(out[11]), int(out[12]), int(out[13]), int(out[14])

        def v(i, rows, cols):
            return ws[out[i] // 4: out[i] // 4 + rows * cols].view(rows, cols)
        return dict(cstar=v(0, TB, A2), h1=v(1, TB, n1), m1=v(2, TB, n1), att=v(3, TB, A2), attended=v(4, TB, A2),
                    h2=v(5, TB, n2), m2=v(6, TB, n2), chat=v(7, TB, M), mem_out=v(8, B, M), zyin=v(9, B, nzy))

    def loss_dict(self, losses):
        """Host view of the loss slots (synchronises).  When `losses` is a plan's own loss tensor the same copy brings the
        plan's status word along (check_status)."""
        plan = None
        for p in self._plans.values():
            if p.losses.data_ptr() == losses.data_ptr():
                plan = p
        if plan is not None:
            st = plan.state.detach().cpu().numpy()
            self.check_status((plan, st))
            l = st[:_lib.MFM_LOSS_SLOTS]
        else:
            l = losses.detach().cpu().numpy()
        c = self.cfg
        gen = c["lda_xl"] * l[1] + c["lda_xa"] * l[2] + c["lda_xv"] * l[3]
        return dict(disc=float(l[0]), gen_l=float(l[1]), gen_a=float(l[2]), gen_v=float(l[3]), gen=float(gen),
                    reg=float(l[4]), loss=float(l[0] + gen + c["lda_mmd"] * l[4]))

    # ------------------------------------------------------------------ timing (bench.py)
    def set_timing(self, T, B, mask, every=1):
        """HIP-event brackets around the kernels in `mask`, on every `every`-th step (a bracket costs ~4.6 us of stream
        time: sampled, it stays out of most steps of a timed region)."""
        h = self.plan(T, B).handle
        _lib.check(_lib.lib().mfm_plan_set_timing(h, int(mask)), "mfm_plan_set_timing")
        _lib.check(_lib.lib().mfm_plan_set_timing_every(h, int(every)), "mfm_plan_set_timing_every")

    def collect_timing(self, T, B):
        L = _lib.lib()
        n = L.mfm_plan_num_kernels()
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        p = self.plan(T, B)
        _lib.check(L.mfm_plan_collect_timing(p.handle, ms, cnt), "mfm_plan_collect_timing")
        return {L.mfm_plan_kernel_name(i).decode(): dict(ms=ms[i], count=cnt[i], kid=i,
                                                         flops=L.mfm_plan_kernel_flops(p.handle, i))
                for i in range(n)}

    def bracket_overhead_ms(self):
        """Median cost of an empty HIP-event bracket on the current stream (what the per-kernel brackets of
        set_timing add to each kernel they surround)."""
        v = C.c_double(0.0)
        _lib.check(_lib.lib().mfm_timing_bracket_overhead_ms(_stream(), C.byref(v)), "mfm_timing_bracket_overhead_ms")
        return v.value

    def work_per_step(self, T, B):
        p = self.plan(T, B)
        return dict(flops=_lib.lib().mfm_plan_flops_per_step(p.handle),
                    bytes=_lib.lib().mfm_plan_bytes_per_step(p.handle))


# ---------------------------------------------------------------------- granular ops (tests, module path)
def gemm_grouped(descs):
    arr = (_lib.GemmDesc * len(descs))(*descs)
    _lib.check(_lib.lib().mfm_gemm_grouped_f32(arr, len(descs), _stream()), "mfm_gemm_grouped_f32")


def make_gemm(a, b, c, m, n, k, a_sm, a_sk, b_sk, b_sn, ldc, bias=None, bias2=None, n_valid=None, batch=1,
              a_sz=0, b_sz=0, c_sz=0, bias_sz=0, accumulate=0, split_k=1, alpha=1.0, c2=None, a_bf16=False, c_bf16=False):
    d = _lib.GemmDesc()
    d.a, d.b, d.c = a.data_ptr(), b.data_ptr(), c.data_ptr()
    d.c2 = c2.data_ptr() if c2 is not None else None
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias2 = bias2.data_ptr() if bias2 is not None else None
    d.a_sz, d.a_sm, d.a_sk = a_sz, a_sm, a_sk
    d.b_sz, d.b_sk, d.b_sn = b_sz, b_sk, b_sn
    d.c_sz, d.ldc, d.bias_sz = c_sz, ldc, bias_sz
    d.m, d.n, d.k = m, n, k
    d.n_valid = n if n_valid is None else n_valid
    d.batch, d.split_k, d.accumulate, d.alpha = batch, split_k, accumulate, alpha
    d.a_bf16, d.c_bf16 = int(bool(a_bf16)), int(bool(c_bf16))
    return d


def make_seq(gates, hs, cs, w_hh, h, w_ih=None, b_ih=None, b_hh=None, h_init=None, is_dec=False,
             dh_ext=None, ld_dh=0, d_h_init=None, dc_ext=None, w_pack=None, store_bf16=False, h_last=None):
    d = _lib.SeqDesc()
    d.gates, d.hs, d.cs = gates.data_ptr(), hs.data_ptr(), cs.data_ptr()
    d.w_hh = w_hh.data_ptr()
    d.w_ih = w_ih.data_ptr() if w_ih is not None else None
    d.b_ih = b_ih.data_ptr() if b_ih is not None else None
    d.b_hh = b_hh.data_ptr() if b_hh is not None else None
    if h_init is not None:
        d.h_init, d.ld_init = h_init.data_ptr(), h_init.stride(0)
    if dh_ext is not None:
        d.dh_ext, d.ld_dh = dh_ext.data_ptr(), ld_dh
    if d_h_init is not None:
        d.d_h_init, d.ld_dinit = d_h_init.data_ptr(), d_h_init.stride(0)
    d.h, d.is_dec = h, int(is_dec)
    d.dc_ext = dc_ext.data_ptr() if dc_ext is not None else None
    d.w_pack = w_pack.data_ptr() if w_pack is not None else None
    d.store_bf16 = int(bool(store_bf16))
    d.h_last = h_last.data_ptr() if h_last is not None else None
    return d


def lstm_seq(descs, T, B, backward=False):
    arr = (_lib.SeqDesc * len(descs))(*descs)
    fn = _lib.lib().mfm_lstm_seq_bwd if backward else _lib.lib().mfm_lstm_seq_fwd
    _lib.check(fn(arr, len(descs), T, B, _stream()), "mfm_lstm_seq_bwd" if backward else "mfm_lstm_seq_fwd")
