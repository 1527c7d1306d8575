"""The fused engine on module storage: what gives `MFM_KL_EF` / `MFM_KL` / `MFM` their `engine` property, flat gradients, and the
autograd / lazy bridges between a reference-style loop and the plan's forward / backward calls (see mfm_model.py's docstring)."""
import os
import weakref
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import engine as E


# ----------------------------------------------------------------------------------- fused engine on module storage
# id(Parameter) -> (weakref to the Parameter, weakref to its model): lets factorized_amd.optim.Adam find the fused model a
# parameter belongs to without putting an (unpicklable) attribute on the Parameter itself
_PARAM_OWNERS = {}


def _owner_of(p):
    ent = _PARAM_OWNERS.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    return ent[1]()


# MFM_MODULE_NO_HANDOVER=1: the module path never uses the in-launch hand-overs (read once: the forward asks on every call)
_NO_HANDOVER = bool(os.environ.get("MFM_MODULE_NO_HANDOVER"))


class _FusedEngineMixin:
    """Gives a model class the `engine` property: an MFMEngine (the one-call fused plan) whose flat parameter buffer
    IS the module's parameter storage -- every nn.Parameter becomes a view into it on first CUDA use, so
    `model.engine.train_step(x, y)` and a reference-style `loss.backward(); optimizer.step()` update the same numbers.
    Also whole-module checkpoints (torch.save(model, path) / torch.load, reference mfm_mosi.py:342-346, 473-481) and
    copy.deepcopy: the engine holds native plan handles and device workspaces, which are dropped from the pickled
    state and re-adopted lazily.

    Flat gradients (round 3): the model also owns ONE flat gradient buffer with the engine's layout; a backward that
    produces all gradients at once (MFM_KL_EF's fused plan) writes into it and every `p.grad` is a persistent view of it, so
    the reference's unchanged loop costs no per-tensor host work in backward / zero_grad / optimizer.step
    (factorized_amd.optim.Adam).  `fast_grads = False` restores the per-tensor autograd path (parameter hooks,
    torch.autograd.grad on parameters)."""
    _engine_variant = "kl_ef"
    fast_grads = True
    # training-mode forwards return lazy outputs / symbolic loss expressions (factorized_amd/lazy.py): the reference's unchanged
    # loop then runs on the launches of the fused step alone.  False: ordinary tensors (round-4 behaviour)
    lazy_losses = True
    _fast_last = True
    # set by factorized_amd.optim.Adam when it owns this model's parameters: its update honours the gradient guard, so the
    # in-launch hand-overs of the small-batch step may be used.  Any other optimizer (torch.optim.Adam, SGD, ...) would apply
    # the gradients of a step whose hand-over gave up: the module path then runs on separate launches, where nothing can fail
    _guarded = False

    def _init_engine_slots(self):
        self._param_names = [n for n, _ in self.named_parameters()]
        self._plist = [p for _, p in self.named_parameters()]      # Parameter objects survive .to()/.cuda()
        self._engine = None
        self._grad_flat = None          # flat gradient buffer (engine layout); p.grad = views of it
        self._grad_present = np.ones(len(self._plist), dtype=bool)     # tensors that received a gradient since zero_grad
        self._grad_fresh = True         # the flat buffer holds zeros: the next backward may overwrite instead of add
        self._flat_leaf = None
        self._register_params()

    def _register_params(self):
        me = weakref.ref(self)
        for p in self._plist:
            _PARAM_OWNERS[id(p)] = (weakref.ref(p), me)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None
        state["_grad_flat"] = None
        state["_flat_leaf"] = None
        state["_guarded"] = False          # (a weak reference; the optimizer of the restored model marks it again)
        return state

    def __setstate__(self, state):
        nn.Module.__setstate__(self, state)
        self._engine = None
        self._grad_flat = None
        self._flat_leaf = None
        self._grad_fresh = True
        # `_plist` must hold the SAME Parameter objects as the sub-modules (pickle keeps identity through its memo;
        # rebuild defensively in case a custom unpickler did not)
        self._plist = [p for _, p in self.named_parameters()]
        if not hasattr(self, "_grad_present") or len(self._grad_present) != len(self._plist):
            self._grad_present = np.ones(len(self._plist), dtype=bool)
        self._register_params()

    def _handover_ok(self):
        g = self._guarded
        if g is not False and g is not True:          # a weak reference to the guard-aware optimizer that owns the parameters:
            g = g() is not None                       # gone (replaced by another optimizer) -> separate launches again
        return bool(g) and not _NO_HANDOVER

    def _fast_ok(self):
        """The flat-gradient path bypasses autograd for the parameters: every tensor gets a gradient view and the fused optimizer
        updates it.  That is wrong for a frozen parameter (requires_grad=False must stay without a gradient and untouched) and
        invisible to parameter hooks, so both fall back to the per-tensor autograd path (`fast_grads = False` semantics)."""
        ok = bool(self.fast_grads)
        if ok:
            for p in self._plist:
                if not p.requires_grad or p._backward_hooks or p._post_accumulate_grad_hooks:
                    ok = False
                    break
        self._fast_last = ok          # (what zero_grad / optimizer.step of the same iteration go by: one walk over the tensors per step)
        return ok

    def _flat_ok(self):
        if self._engine is None:
            return False
        eng = self._engine
        base = eng.params.data_ptr()
        o0, ol = eng.layout.slots[0][0], eng.layout.slots[-1][0]
        return (self._plist[0].data_ptr() == base + 4 * o0 and self._plist[-1].data_ptr() == base + 4 * ol)

    def _adopt(self, device):
        cfg = dict(self._configs[0])
        for k, dflt in (("lda_xl", 1.0), ("lda_xa", 1.0), ("lda_xv", 1.0), ("lda_mmd", 1.0)):
            cfg.setdefault(k, dflt)
        eng = E.MFMEngine([cfg] + list(self._configs[1:]), device=device, variant=self._engine_variant)
        assert list(eng.layout.shapes.keys()) == self._param_names, "parameter naming drifted from the reference"
        pd = OrderedDict(self.named_parameters())
        eng.load_weights(OrderedDict((n, p.detach()) for n, p in pd.items()))
        views = eng.param_views()
        for n, p in pd.items():
            p.data = views[n]
        self._engine = eng
        self._grad_flat = None
        self._grad_fresh = True

    @property
    def engine(self):
        """The fused engine sharing this module's parameter storage (built on first CUDA use)."""
        if not self._flat_ok():
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise _lib.MfmError("%s: parameters are on %s; move the model to the GPU first" % (type(self).__name__, dev))
            self._adopt(dev)
        return self._engine

    # ------------------------------------------------------------------ flat gradients
    def _flat_grads(self):
        eng = self.engine
        if self._grad_flat is None or self._grad_flat.device != eng.params.device or self._grad_flat.numel() != eng.layout.total:
            self._grad_flat = torch.zeros_like(eng.params)
            self._grad_fresh = True
        return self._grad_flat

    def _grad_views_attached(self):
        g = self._grad_flat
        if g is None or self._engine is None:
            return False
        lay = self._engine.layout
        g0, g1 = self._plist[0].grad, self._plist[-1].grad
        return (g0 is not None and g1 is not None and g0.data_ptr() == g.data_ptr() + 4 * lay.slots[0][0]
                and g1.data_ptr() == g.data_ptr() + 4 * lay.slots[-1][0])

    def _attach_grad_views(self):
        g = self._flat_grads()
        for p, (o, n, shp) in zip(self._plist, self._engine.layout.slots):
            p.grad = g[o:o + n].view(shp)

    def _zero_flat_grads(self, set_to_none=True):
        """optimizer.zero_grad() of factorized_amd.optim.Adam: one launch; set_to_none=True marks every tensor as
        'no gradient yet' (the optimizer skips what the next backward does not reach, like torch with .grad = None)"""
        if not (set_to_none and self.lazy_losses and self.training and self._fast_last):
            # (set_to_none on a lazily-training model: no launch -- torch would leave `.grad = None` behind, here the views stay
            # attached and hold the previous step's values until the next forward's first launch clears the buffer; every
            # backward that follows OVERWRITES it, and `_grad_present` makes the optimizer skip what no backward reached)
            self._grad_flat.zero_()
        self._grad_fresh = True
        if set_to_none:
            self._grad_present[:] = False

    def _detach_grad_views(self):
        """hand the gradients back to plain per-tensor autograd (a frozen parameter or a hook appeared after fast-path steps):
        accumulated values survive as clones, 'nothing yet' becomes None; the flat buffer is dropped"""
        if self._grad_flat is None or not self._grad_views_attached():
            return
        for i, p in enumerate(self._plist):
            keep = (not self._grad_fresh) and bool(self._grad_present[i]) and p.requires_grad
            p.grad = p.grad.detach().clone() if keep else None
        self._grad_flat = None
        self._grad_fresh = True
        self._grad_present[:] = True

    def _group_masks(self):
        """which tensors each upstream gradient of the factorized model reaches exclusively: d y_hat -> the classifier;
        d x_hat_m -> decoder m and its z -> f MLP (the staged losses of train_beta_vae, reference mfm_mosi.py:278-281)"""
        mk = getattr(self, "_masks", None)
        if mk is None:
            names = self._param_names
            def sel(*prefixes):
                return np.array([n.startswith(prefixes) for n in names], dtype=bool)
            mk = dict(disc=sel("fy_to_y_"), l=sel("decoder_l.", "zl_to_fl_"), a=sel("decoder_a.", "za_to_fa_"),
                      v=sel("decoder_v.", "zv_to_fv_"))
            mk["shared"] = ~(mk["disc"] | mk["l"] | mk["a"] | mk["v"])
            self._masks = mk
        return mk


# ----------------------------------------------------------------------------------- MFM_KL_EF
class _KLEFFn(torch.autograd.Function):
    """The whole MFM_KL_EF forward as ONE plan call; backward = mfm_plan_backward_ext with the
    upstream gradients autograd hands us (any user loss)."""

    @staticmethod
    def forward(ctx, x, module, *params):
        if x.requires_grad:
            raise _lib.MfmError("MFM_KL_EF.forward: the input requires grad; the fused plan does not produce d loss / d x "
                                "(the reference never asks for it) -- detach the batch")
        eng = module.engine
        # (per-tensor gradients: whatever optimizer applies them knows nothing of the gradient guard -> separate launches)
        out = eng.forward(x, None, train=module.training, want_xhat=True, handover=False)
        kld = out["losses"][4].clone()
        ctx.module = module
        # the plan's workspace for (T,B) holds the activations of the LAST forward only: remember which one
        # this graph belongs to, so that backward can refuse to differentiate somebody else's activations
        plan = eng.plan(x.shape[0], x.shape[1])
        ctx.plan, ctx.serial = plan, plan.fwd_serial
        ctx.save_for_backward(x)
        return out["x_l_hat"], out["x_a_hat"], out["x_v_hat"], out["y_hat"], kld

    @staticmethod
    def backward(ctx, d_xl, d_xa, d_xv, d_y, d_kld):
        (x,) = ctx.saved_tensors
        module = ctx.module
        eng = module.engine
        T, B, _ = x.shape
        plan = ctx.plan
        if eng.plan(T, B) is not plan or plan.fwd_serial != ctx.serial:
            raise RuntimeError("MFM_KL_EF backward: another forward with the same (T=%d, B=%d) ran on this model since "
                               "the graph was built; its activations replaced this one's in the plan workspace.  Call "
                               "backward() before the next forward (gradient accumulation over several forwards: "
                               "backward each one first)" % (T, B))
        if plan.consumed:
            raise RuntimeError("MFM_KL_EF backward: this graph was already back-propagated (BPTT overwrites the saved "
                               "gates in place; retain_graph is not supported on the fused plan)")
        plan.consumed = True
        d_l, d_a, d_v = eng.cfg["input_dims"]
        dev = x.device

        def z(t, shape):
            return torch.zeros(shape, device=dev) if t is None else t.contiguous().float()
        d_xl, d_xa, d_xv = z(d_xl, (T, B, d_l)), z(d_xa, (T, B, d_a)), z(d_xv, (T, B, d_v))
        d_y = z(d_y, (B, eng.cfg["output_dim"]))
        d_kld = z(d_kld, ()).reshape(1)
        plan.ensure_handover(False)
        eng.backward_ext(x, d_xl, d_xa, d_xv, d_y, d_kld)
        # one copy of the flat gradient buffer, handed out as per-parameter views (the plan overwrites its own
        # buffer on the next call; 78 separate clones cost ~0.4 ms of host time per step)
        flat = eng.grads.clone()
        lay = eng.layout
        return (None, None) + tuple(flat[o:o + n].view(shp) for o, n, shp in lay.slots)


def _check_plan_live(plan, eng, serial, T, B):
    if eng.plan(T, B) is not plan or plan.fwd_serial != serial:
        raise RuntimeError("MFM_KL_EF backward: another forward with the same (T=%d, B=%d) ran on this model since "
                           "the graph was built; its activations replaced this one's in the plan workspace.  Call "
                           "backward() before the next forward (gradient accumulation over several forwards: "
                           "backward each one first)" % (T, B))
    if plan.consumed:
        raise RuntimeError("MFM_KL_EF backward: this graph was already back-propagated (BPTT overwrites the saved "
                           "gates in place; retain_graph is not supported on the fused plan)")
    plan.consumed = True


def _into_flat(module, eng, present, run):
    """run(out) fills a flat gradient buffer; route it into the model's flat gradients (overwrite when nothing accumulated since
    zero_grad, else add) and keep every `p.grad` a view of that buffer"""
    flat = module._flat_grads()
    attached = module._grad_views_attached()
    if module._grad_fresh or not attached:
        run(flat)
        if not attached:
            module._attach_grad_views()
            module._grad_present[:] = False
    else:
        run(None)
        flat.add_(eng.grads)
        # the guard word is a flag, not a sum: a NaN added here would never leave (per-tensor zeroing does not reach the
        # guard granule) and the guarded optimizer would skip every later step
        g = eng.layout.guard
        flat[g:g + 1].copy_(eng.grads[g:g + 1])
    module._grad_fresh = False
    module._grad_present |= present


def _flat_backward_ext(module, plan, serial, x, d_xl, d_xa, d_xv, d_y, d_kld):
    """backward of one fused forward for arbitrary upstream gradients (None = that output is unused) into the flat buffer"""
    eng = module.engine
    T, B, _ = x.shape
    _check_plan_live(plan, eng, serial, T, B)
    d_l, d_a, d_v = eng.cfg["input_dims"]
    dev = x.device
    mk = module._group_masks()
    present = mk["shared"].copy()
    for key, g in (("l", d_xl), ("a", d_xa), ("v", d_xv), ("disc", d_y)):
        if g is not None:
            present |= mk[key]

    def z(t, shape):
        return torch.zeros(shape, device=dev) if t is None else t.contiguous().float()
    d_xl, d_xa, d_xv = z(d_xl, (T, B, d_l)), z(d_xa, (T, B, d_a)), z(d_xv, (T, B, d_v))
    d_y = z(d_y, (B, eng.cfg["output_dim"]))
    d_kld = z(d_kld, ()).reshape(1)
    plan.ensure_handover(eng.handover and module._handover_ok())
    _into_flat(module, eng, present, lambda out: eng.backward_ext(x, d_xl, d_xa, d_xv, d_y, d_kld, out=out))


class _KLEFFastFn(torch.autograd.Function):
    """_KLEFFn without per-tensor autograd traffic: the only differentiable input is a dummy leaf; backward writes ALL
    parameter gradients into the model's flat gradient buffer (adding when something is already there) and makes sure
    every `p.grad` is its view of that buffer."""

    @staticmethod
    def forward(ctx, x, module, leaf):
        if x.requires_grad:
            raise _lib.MfmError("%s.forward: the input requires grad; the fused plan does not produce d loss / d x "
                                "(the reference never asks for it) -- detach the batch" % type(module).__name__)
        eng = module.engine
        plan = eng.plan(x.shape[0], x.shape[1])
        out = eng.forward(x, None, train=module.training, want_xhat=True, handover=eng.handover and module._handover_ok())
        kld = out["losses"][4].clone()
        ctx.module = module
        ctx.plan, ctx.serial = plan, plan.fwd_serial
        ctx.save_for_backward(x)
        # an output the loss does not use must arrive in backward as None, not as a zero tensor: that is how the stage
        # losses (gen + reg: y_hat unused; disc + reg: the reconstructions unused) tell which tensors get NO gradient
        ctx.set_materialize_grads(False)
        return out["x_l_hat"], out["x_a_hat"], out["x_v_hat"], out["y_hat"], kld

    @staticmethod
    def backward(ctx, d_xl, d_xa, d_xv, d_y, d_kld):
        (x,) = ctx.saved_tensors
        _flat_backward_ext(ctx.module, ctx.plan, ctx.serial, x, d_xl, d_xa, d_xv, d_y, d_kld)
        return None, None, None


class _LazyRealFn(torch.autograd.Function):
    """The outputs of a LAZY forward (factorized_amd/lazy.py) as ordinary tensors of one autograd node -- what a lazy output or
    loss expression turns into when it is used in a way the symbolic path does not cover.  The plan already ran: forward only
    clones its buffers; backward is _KLEFFastFn's."""

    @staticmethod
    def forward(ctx, leaf, step):
        ctx.module, ctx.plan, ctx.serial = step.module, step.plan, step.serial
        ctx.save_for_backward(step.x)
        ctx.set_materialize_grads(False)
        v = step.plan.out_views
        return v[0].clone(), v[1].clone(), v[2].clone(), v[3].clone(), step.plan.losses[4].clone()

    @staticmethod
    def backward(ctx, d_xl, d_xa, d_xv, d_y, d_kld):
        (x,) = ctx.saved_tensors
        _flat_backward_ext(ctx.module, ctx.plan, ctx.serial, x, d_xl, d_xa, d_xv, d_y, d_kld)
        return None, None


def _lazy_forward(module, x):
    """training-mode forward with lazy outputs (factorized_amd/lazy.py), or None when this plan cannot serve them"""
    from . import lazy
    eng = module.engine
    eng._check_inputs(x, None)          # (shape / dtype / device / contiguity: the kernels index x with the plan's D)
    if x.shape[0] < 1 or x.shape[1] < 1:
        raise _lib.MfmError("empty batch %s" % (tuple(x.shape),))
    T, B, _ = x.shape
    plan = eng.plan(T, B)
    if plan.out_views is None:
        return None
    plan.ensure_handover(eng.handover and module._handover_ok())
    if module._flat_leaf is None or module._flat_leaf.device != x.device:
        module._flat_leaf = torch.zeros((), device=x.device, requires_grad=True)
    flat = module._flat_grads()
    # nothing accumulated since zero_grad: the forward's first launch clears the flat gradient buffer (for free, on its role
    # workgroups) and the backward writes straight into it -- the launches of engine.train_step, nothing else
    zero = flat if (module._grad_fresh and module._grad_views_attached()) else None
    eng.forward_train(x, plan, zero)
    step = lazy.PlanStep(module, eng, plan, x)
    v = plan.out_views
    outs = [lazy.LazyOut(v[0], step, 0), lazy.LazyOut(v[1], step, 1), lazy.LazyOut(v[2], step, 2), lazy.LazyOut(v[3], step, 3)]
    return outs, lazy.LossExpr(step, {lazy.REG: 1.0}), 0.0


def _lazy_backward(step, coef, labels, terms):
    """loss.backward() of a symbolic loss expression: one mfm_plan_backward_weighted call into the flat gradient buffer"""
    module, eng, plan, x = step.module, step.eng, step.plan, step.x
    T, B, _ = x.shape
    _check_plan_live(plan, eng, step.serial, T, B)
    # (host time counts: the unchanged loop is host-bound against a 0.15 ms device step -- the mask of a given set of terms and
    #  the weight struct of a given expression are built once and reused)
    mk = module._group_masks()
    tkey = tuple(sorted(terms))
    present = mk.get(tkey)
    if present is None:
        present = mk["shared"].copy()
        for k, key in ((1, "l"), (2, "a"), (3, "v"), (0, "disc")):
            if k in terms:
                present |= mk[key]
        mk[tkey] = present
    gen_on = any(coef.get(k, 0.0) != 0.0 for k in (1, 2, 3))
    wkey = (float(coef.get(0, 0.0)), gen_on, float(coef.get(4, 0.0)), labels is not None)
    cache = module.__dict__.setdefault("_lw_cache", {})
    w = cache.get(wkey)
    if w is None:
        if len(cache) > 16:
            cache.clear()
        w = _lib.LossWeights()
        w.disc = wkey[0]
        w.gen_l, w.gen_a, w.gen_v = step.lda if gen_on else (0.0, 0.0, 0.0)      # (checked equal by LossExpr._fast_backward_ok)
        w.reg = wkey[2]
        w.write_disc_loss = 1 if labels is not None else 0
        cache[wkey] = w
    plan.ensure_handover(eng.handover and module._handover_ok())
    _into_flat(module, eng, present, lambda out: eng.backward_weighted(x, labels, w, plan, out=out))
