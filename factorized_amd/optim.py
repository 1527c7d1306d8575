"""Drop-in `optim.Adam` for the reference's UNCHANGED training loops (reference mfm_mosi.py:403, 427-441):

    import factorized_amd.optim as optim                       # instead of: import torch.optim as optim
    from factorized_amd.mfm_model import MFM_KL_EF              # instead of: from mfm_model import MFM_KL_EF
    ...
    optimizer = optim.Adam(model.parameters())                  # :403, before model.to(device) as in the reference
    ...
    optimizer.zero_grad(); decoded, reg, missing = model.forward(batch_X); ...; loss.backward(); optimizer.step()

`Adam` is a `torch.optim.Optimizer` (so `ReduceLROnPlateau(optimizer, 'min')`, `param_groups[0]['lr']`, `zero_grad()` work as
in the reference).  Parameters that belong to a model with the fused engine (`MFM_KL_EF`, `MFM_KL`, `MFM`: every
nn.Parameter is a view into ONE flat buffer) and whose `.grad`s are the views of the model's flat gradient buffer (what
`MFM_KL_EF`'s backward leaves behind) are updated by ONE launch of `mfm_adam_flat` -- or `mfm_adam_flat_spans` when some
tensors received no gradient: like torch.optim.Adam, a parameter without a gradient is skipped and step counts are kept
per tensor.  Everything else (other modules, models on the composed autograd path) goes through a stock
`torch.optim.Adam` with the same hyper-parameters.  `torch.optim.Adam` itself keeps working too -- it is just host-bound
(78 tensors per step); see INTEGRATION.md for the measured step times.

`zero_grad()` of this class clears a fused model's flat gradient buffer with one launch and marks every tensor "no gradient
yet" (= torch's `set_to_none=True`: the next `step()` skips tensors the next backward does not reach) while leaving the
`.grad` views attached; `zero_grad(set_to_none=False)` keeps zero gradients in place, which is the reference's PyTorch-0.4
behaviour (a tensor that once had a gradient keeps moving on its decaying first moment)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

ReduceLROnPlateau = torch.optim.lr_scheduler.ReduceLROnPlateau      # convenience: `optim.lr_scheduler` users import torch's
lr_scheduler = torch.optim.lr_scheduler
SGD = torch.optim.SGD


def _owner(p):
    from .mfm_model import _owner_of
    return _owner_of(p)


# Who may keep the in-launch hand-overs on is decided PER STEP by the optimizer that actually steps the model: this class marks
# the models it owns in every step(); any OTHER torch optimizer that steps parameters of a fused model -- an optimizer swap, an
# LR finder, per-stage optimizers built while the first one is still referenced -- takes the permission away before its
# update (a global step pre-hook: it knows nothing of the gradient guard and would apply a step whose hand-over gave up).
_FOREIGN = {}


def _foreign_step_hook(opt, args, kwargs):
    if isinstance(opt, Adam) or getattr(opt, "_mfm_inner", False):
        return
    key = id(opt)
    n = sum(len(g["params"]) for g in opt.param_groups)
    hit = _FOREIGN.get(key)
    if hit is None or hit[0] != n or hit[1]() is not opt:
        import weakref
        mods, seen = [], set()
        for g in opt.param_groups:
            for p in g["params"]:
                m = _owner(p)
                if m is not None and id(m) not in seen:
                    seen.add(id(m))
                    mods.append(weakref.ref(m))
        if len(_FOREIGN) > 64:
            _FOREIGN.clear()
        hit = _FOREIGN[key] = (n, weakref.ref(opt), mods)
    for r in hit[2]:
        m = r()
        if m is not None:
            m._guarded = False


import importlib as _il
_OPT_MOD = _il.import_module("torch.optim.optimizer")
_OPT_MOD.register_optimizer_step_pre_hook(_foreign_step_hook)
_GLOBAL_PRE, _GLOBAL_POST = _OPT_MOD._global_optimizer_pre_hooks, _OPT_MOD._global_optimizer_post_hooks


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, capturable=False):
        if weight_decay != 0 or amsgrad:
            raise ValueError("factorized_amd.optim.Adam: weight_decay / amsgrad are not used by the reference and not built")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        super().__init__(params, defaults)
        import weakref
        # module -> state of a fused model: flat moments, per-tensor step counts (weak keys: a model that is gone takes its
        # optimizer state with it)
        self._fused = weakref.WeakKeyDictionary()
        self._fallback = None       # stock torch.optim.Adam over everything that is not fused
        self._fallback_ids = None
        self._fm_cache = {}
        self._pending_fused = None  # fused states of a load_state_dict() waiting for their models' first step
        # capturable=True (torch.optim.Adam's flag): step count and learning rate of the fused update live in device memory
        # (mfm_adam_flat_dev), so a whole training step can be captured into a hipGraph and replayed (train.GraphedModuleStep)
        self._capturable = bool(capturable)
        # this optimizer honours the gradient guard: the models it owns may run their in-launch hand-overs (a model under
        # any other optimizer stays on separate launches, mfm_model._FusedEngineMixin._guarded)
        for group in self.param_groups:
            self._fused_modules(group)

    # ------------------------------------------------------------------ helpers
    def _fused_modules(self, group):
        """models whose parameter list lies entirely inside `group` (the reference has one group: model.parameters());
        cached per group while its parameter list is the same list of the same length"""
        key = (id(group["params"]), len(group["params"]), id(group["params"][0]) if group["params"] else 0)
        hit = self._fm_cache.get(id(group))
        if hit is not None and hit[0] == key and all(r() is not None for r in hit[2]):
            return hit[1]
        import weakref
        ids = {id(p) for p in group["params"]}
        seen, out = set(), []
        for p in group["params"]:
            m = _owner(p)
            if m is None or id(m) in seen:
                continue
            seen.add(id(m))
            if all(id(q) in ids for q in m._plist):
                out.append(m)
                import weakref as _wr
                m._guarded = _wr.ref(self)
        self._fm_cache[id(group)] = (key, out, [weakref.ref(m) for m in out])
        return out

    def _state_for(self, m, eng):
        st = self._fused.get(m)
        if st is None or st["m"].numel() != eng.layout.total or st["m"].device != eng.params.device:
            st = dict(m=torch.zeros_like(eng.params), v=torch.zeros_like(eng.params),
                      steps=np.zeros(len(eng.layout.slots), dtype=np.int64))
            if self._pending_fused:                     # state restored by load_state_dict(), in the order it was saved
                saved = self._pending_fused.pop(0)
                if saved["m"].numel() == eng.layout.total and len(saved["steps"]) == len(eng.layout.slots):
                    st["m"].copy_(saved["m"]); st["v"].copy_(saved["v"])
                    st["steps"][:] = np.asarray(saved["steps"], dtype=np.int64)
                else:
                    raise _lib.MfmError(
                        "factorized_amd.optim.Adam.load_state_dict: the saved fused state (%d elements, %d tensors) does not fit "
                        "this model's flat layout (%d elements, %d tensors) -- a checkpoint of another model / library version; "
                        "refusing to restart the moments silently" % (saved["m"].numel(), len(saved["steps"]), eng.layout.total,
                                                                      len(eng.layout.slots)))
            self._fused[m] = st
        return st

    def _device_scalars(self, st, eng, lr):
        """capturable mode: the step counter and the learning rate as device words of this model's state"""
        dev = eng.params.device
        if "step_dev" not in st:
            st["step_dev"] = torch.full((1,), int(st["steps"][0]), dtype=torch.int32, device=dev)
            st["lr_dev"] = torch.zeros(1, dtype=torch.float32, device=dev)
            st["lr_host"] = None
        if torch.is_tensor(lr):
            if lr.is_cuda and lr.dtype == torch.float32:
                return st["step_dev"], lr                     # the caller's own device scalar (GraphedModuleStep.set_lr)
            lr = float(lr)
        if st["lr_host"] != lr:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.MfmError("factorized_amd.optim.Adam(capturable=True): the learning rate changed inside a stream "
                                    "capture; pass lr as a float32 device tensor or change it between replays")
            st["lr_dev"].fill_(lr)
            st["lr_host"] = lr
        return st["step_dev"], st["lr_dev"]

    def _migrate_back(self, m, st, eng):
        """a model returns to the flat path after steps through the stock optimizer (a parameter was frozen and is trainable
        again): moments and step counts the stock optimizer holds for its tensors come back into the flat state"""
        fb = self._fallback
        if fb is None:
            return
        for i, p in enumerate(m._plist):
            s = fb.state.get(p)
            if not s:
                continue
            o, n, shp = eng.layout.slots[i]
            st["m"][o:o + n].view(shp).copy_(s["exp_avg"])
            st["v"][o:o + n].view(shp).copy_(s["exp_avg_sq"])
            st["steps"][i] = int(float(s["step"]))
            del fb.state[p]
        if "step_dev" in st:
            st["step_dev"].fill_(int(st["steps"].max()))

    def _fused_step(self, m, group):
        eng = m.engine
        import weakref
        m._guarded = weakref.ref(self)          # (per step: the optimizer that steps the model answers for the guard)
        gflat = getattr(m, "_grad_flat", None)
        if gflat is None or not m._grad_views_attached():
            return False                     # gradients are ordinary per-tensor tensors: the stock optimizer handles them
        if not m._fast_last:
            # a parameter was frozen / got a hook after fast-path steps: the flat path would move it (or skip everything);
            # hand the gradients back to per-tensor tensors and let the stock optimizer apply torch's rules
            m._detach_grad_views()
            return False
        if eng.poll_status():
            # a hand-over of this step (or an earlier one) gave up: its gradients carry the NaN guard, the launch below leaves
            # the parameters alone.  Clear the status, fall back to separate launches for the rest of the run, say so.
            import warnings
            eng.check_status(raise_on_error=False)
            warnings.warn(eng.status_message(), RuntimeWarning, stacklevel=3)
        st = self._state_for(m, eng)
        if self._fallback is not None and self._fallback.state:
            self._migrate_back(m, st, eng)
        lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
        if torch.is_tensor(lr) and not self._capturable:
            lr = float(lr)
        present = m._grad_present
        L = _lib.lib()
        stream = C.c_void_p(torch._C._cuda_getCurrentRawStream(eng.params.device.index))
        ptr = lambda t: C.c_void_p(t.data_ptr())
        steps = st["steps"]
        # guard word of the flat gradient buffer: the plan's backward stores a NaN there when a hand-over inside one of its
        # launches gave up -- the launch below then leaves parameters and moments alone (engine.check_status() reports it)
        guard = C.c_void_p(gflat.data_ptr() + 4 * eng.layout.guard)
        if self._capturable:
            if not (present.all() and (steps == steps[0]).all()):
                raise _lib.MfmError("factorized_amd.optim.Adam(capturable=True): every tensor needs a gradient in every step (one "
                                    "device-side step counter); staged losses train through the eager optimizer")
            step_dev, lr_dev = self._device_scalars(st, eng, group["lr"])
            _lib.check(L.mfm_adam_flat_dev(ptr(eng.params), ptr(gflat), ptr(st["m"]), ptr(st["v"]), eng.layout.total, ptr(step_dev),
                                           ptr(lr_dev), b1, b2, eps, 1.0, guard, stream), "mfm_adam_flat_dev")
            steps += 1           # (host mirror: exact in eager use, a lower bound under graph replay -- see state_dict())
            return True
        if present.all() and (steps == steps[0]).all():
            steps += 1
            _lib.check(L.mfm_adam_flat_guarded(ptr(eng.params), ptr(gflat), ptr(st["m"]), ptr(st["v"]), eng.layout.total,
                                               int(steps[0]), lr, b1, b2, eps, 1.0, guard, stream), "mfm_adam_flat_guarded")
            return True
        # some tensors have no gradient (stage losses, unused layers): contiguous runs of present tensors with equal
        # step counts become spans (tensor starts are 64-float aligned: span bounds are multiples of 4)
        order = np.argsort([o for o, _, _ in eng.layout.slots])
        starts = [eng.layout.slots[i][0] for i in order] + [eng.layout.guard]
        spans = []
        for k, i in enumerate(order):
            if not present[i]:
                continue
            steps[i] += 1
            b, e_, s_ = starts[k], starts[k + 1], int(steps[i])
            if spans and spans[-1][1] == b and spans[-1][2] == s_:
                spans[-1] = (spans[-1][0], e_, s_)
            else:
                spans.append((b, e_, s_))
        for k in range(0, len(spans), _lib.MFM_ADAM_MAX_SPANS):
            part = spans[k:k + _lib.MFM_ADAM_MAX_SPANS]
            arr = (_lib.AdamSpan * len(part))()
            for j, (b, e_, s_) in enumerate(part):
                arr[j].begin, arr[j].end, arr[j].step = b, e_, s_
            _lib.check(L.mfm_adam_flat_spans_guarded(ptr(eng.params), ptr(gflat), ptr(st["m"]), ptr(st["v"]), arr, len(part), lr, b1,
                                                     b2, eps, 1.0, guard, stream), "mfm_adam_flat_spans_guarded")
        return True

    def _fallback_step(self, rest):
        if not rest:
            return
        ids = tuple(id(p) for _, ps in rest for p in ps)
        if self._fallback is None or self._fallback_ids != ids:
            groups = [dict(params=ps, lr=g["lr"], betas=g["betas"], eps=g["eps"]) for g, ps in rest]
            self._fallback = torch.optim.Adam(groups)
            self._fallback._mfm_inner = True        # (steps on behalf of this class: not a foreign optimizer)
            self._fallback_ids = ids
            self._migrate_fused_state(rest)
            fb = getattr(self, "_pending_fallback", None)
            if fb is not None:
                self._fallback.load_state_dict(fb)
                self._pending_fallback = None
        for fg, (g, _) in zip(self._fallback.param_groups, rest):
            fg["lr"], fg["betas"], fg["eps"] = g["lr"], g["betas"], g["eps"]      # schedulers act on OUR groups
        self._fallback.step()

    def _migrate_fused_state(self, rest):
        """parameters of a fused model that now go through the stock optimizer (a parameter was frozen / got a hook after
        fast-path steps): their moments and step counts move along -- Adam must not restart"""
        for _, ps in rest:
            for p in ps:
                m = _owner(p)
                st = self._fused.get(m) if m is not None else None
                if st is None or p in self._fallback.state:
                    continue
                i = next((k for k, q in enumerate(m._plist) if q is p), None)
                if i is None:
                    continue
                o, n, shp = m.engine.layout.slots[i]
                steps = int(st["step_dev"].item()) if "step_dev" in st else int(st["steps"][i])
                if steps == 0:
                    continue
                self._fallback.state[p] = dict(step=torch.tensor(float(steps)), exp_avg=st["m"][o:o + n].view(shp).clone(),
                                               exp_avg_sq=st["v"][o:o + n].view(shp).clone())
        # (the flat state stays: when the model returns to the flat path, _migrate_back brings the moments home instead of
        #  restarting them from zero)

    def reset_state(self):
        """moments and step counters of every fused model back to zero, in place (a captured graph keeps pointing at them)"""
        for st in self._fused.values():
            st["m"].zero_(); st["v"].zero_(); st["steps"][:] = 0
            if "step_dev" in st:
                st["step_dev"].zero_()

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """torch's dict plus the flat Adam state of every fused model under "fused" (in the order the models appear in the
        parameter groups): first / second moments and per-tensor step counts (capturable mode: the device counter)."""
        sd = super().state_dict()
        fused = []
        for group in self.param_groups:
            for m in self._fused_modules(group):
                st = self._fused.get(m)
                if st is None:
                    continue
                steps = st["steps"].copy()
                if "step_dev" in st:
                    steps[:] = int(st["step_dev"].item())
                fused.append(dict(m=st["m"].detach().clone(), v=st["v"].detach().clone(), steps=steps.tolist()))
        if self._pending_fused:          # loaded, not stepped yet: what was loaded is still the state
            fused += [dict(m=f["m"], v=f["v"], steps=list(f["steps"])) for f in self._pending_fused]
        sd["fused"] = fused
        if self._fallback is not None:
            sd["fallback"] = self._fallback.state_dict()
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        fused = sd.pop("fused", None)
        fb = sd.pop("fallback", None)
        super().load_state_dict(sd)
        self._fused.clear()
        self._pending_fused = [dict(m=f["m"], v=f["v"], steps=list(f["steps"])) for f in fused] if fused else None
        self._pending_fallback = fb

    # ------------------------------------------------------------------ Optimizer interface
    def step(self, closure=None):
        """(round 6) Not wrapped by torch's `profile_hook_step` (`step.hooked` below): that wrapper opens a record_function scope
        around every step, ~10 us of host time against a 150 us device step the unchanged loop has to keep fed.  Step pre / post
        hooks registered on this optimizer or globally are still honoured."""
        hooks = self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or len(_GLOBAL_PRE) > 1 or _GLOBAL_POST
        if hooks:
            for h in list(_GLOBAL_PRE.values()) + list(self._optimizer_step_pre_hooks.values()):
                if h is not _foreign_step_hook:
                    h(self, (closure,) if closure is not None else (), {})
        prev = torch.is_grad_enabled()
        torch._C._set_grad_enabled(False)
        try:
            loss = self._step(closure)
        finally:
            torch._C._set_grad_enabled(prev)
        if hooks:
            for h in list(self._optimizer_step_post_hooks.values()) + list(_GLOBAL_POST.values()):
                h(self, (closure,) if closure is not None else (), {})
        return loss
    step.hooked = True

    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rest = []
        for group in self.param_groups:
            stepped, ndone = [], 0
            for m in self._fused_modules(group):
                if m._plist[0].is_cuda and self._fused_step(m, group):
                    stepped.append(m)
                    ndone += len(m._plist)
            if ndone == len(group["params"]):
                continue                              # (the reference's case: one model, one group, nothing left)
            done = {id(p) for m in stepped for p in m._plist}
            left = [p for p in group["params"] if id(p) not in done]
            if left:
                rest.append((group, left))
        self._fallback_step(rest)
        return loss

    def zero_grad(self, set_to_none=True):
        cleared, nh, total = [], 0, 0
        for group in self.param_groups:
            total += len(group["params"])
            for m in self._fused_modules(group):
                if getattr(m, "_grad_flat", None) is not None and m._grad_views_attached():
                    m._zero_flat_grads(set_to_none)
                    cleared.append(m)
                    nh += len(m._plist)
        if nh == total:
            return
        handled = {id(p) for m in cleared for p in m._plist}
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in handled or p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.zero_()
