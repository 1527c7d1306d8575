"""Lazy losses for the reference's UNCHANGED training loop (reference mfm_mosi.py:427-441, mfm_you.py:470-488, the stage
losses of train_beta_vae mfm_mosi.py:255-285).

The loop calls `model.forward(batch_X)`, then builds

    loss = l1_loss(y_hat.squeeze(1), batch_y) + sum_m lda_m * l2_loss(x_m_hat, batch_X[:, :, slice_m]) + lda_mmd * reg + missing

from the outputs and calls `loss.backward()`.  The fused plan's forward has already computed the three MSE terms, the
regulariser and the lda_m-scaled d x_hat in the decoder fc1 epilogue (csrc/plan.hip), so none of the ~35 torch loss / autograd
kernels of that expression -- and no x_hat tensor -- is needed.  In training mode the fused models therefore return

    LazyOut   tensor subclasses for x_l_hat / x_a_hat / x_v_hat / y_hat: views of the plan's own output buffers.  Passing one
              to `F.mse_loss` with a target that provably ALIASES the matching column slice of the batch the forward ran on
              (same storage, offset, shape, strides; reduction 'mean'), or y_hat (optionally `.squeeze(1)`) to `F.l1_loss` /
              `F.cross_entropy` with plain labels, yields a
    LossExpr  a 0-d tensor subclass that is a SYMBOLIC weighted sum of the plan's loss slots (host-side coefficients).  Python
              scalar arithmetic (`lda * expr`, `expr + expr`, `+ 0.0`, `/ k`, unary minus) stays symbolic; `.backward()` is one
              call of `mfm_plan_backward_weighted` (the weights become kernel arguments), `.item()` one copy of the plan's 64-byte
              state block (which also brings the hand-over status word along).

Anything else -- another reduction, a target that is not the batch slice, weights that differ from the plan's lda_m, an output
used a second way, `torch.*` functions on an expression, `backward(retain_graph=...)` -- MATERIALISES: the outputs become
ordinary tensors of one autograd node (clones of the plan's buffers; backward = mfm_plan_backward_ext with whatever upstream
gradients autograd delivers, the round-3/4 path), every expression is re-evaluated on them with torch ops, and the original
function is called on the results.  So the lazy path only ever changes speed, never what is computed.

An output or expression of an EARLIER forward (the plan's buffers hold the latest one only) raises when it is read.

The mechanics live behind a small step interface (`StepBase`), so that the algebra / aliasing / fallback logic is tested on the
CPU with a torch stand-in (tests/test_lazy_host.py); `PlanStep` binds it to the engine.
"""
import warnings

import torch
import torch.nn.functional as F

_NUM = (int, float)
DISC, GEN_L, GEN_A, GEN_V, REG = 0, 1, 2, 3, 4
_DisableTF = torch._C.DisableTorchFunctionSubclass


def _is_num(v):
    if isinstance(v, bool):
        return False
    if isinstance(v, _NUM):
        return True
    try:                                  # numpy scalars (the reference's configs come out of json / numpy)
        import numpy as np
        return isinstance(v, np.generic) and np.isscalar(v)
    except Exception:
        return False


class StepBase:
    """One forward of a fused model, shared by its lazy outputs and loss expressions.

    Subclasses provide: `views` (x_l_hat, x_a_hat, x_v_hat, y_hat plain tensors), `x` [T,B,D], `dims` (d_l, d_a, d_v), `lda`
    (plan's reconstruction weights), `loss_kind` (0 L1, 1 CE), and the methods below."""
    real = None            # the materialised outputs (x_l_hat, x_a_hat, x_v_hat, y_hat, reg) once anything fell back
    disc_in_slot = False   # a lazy backward ran with the labels: loss slot 0 holds the discriminative loss
    disc = None            # (kind, labels, squeezed) once y_hat met its criterion
    targets = None         # {term: target tensor} of the intercepted mse terms (for re-evaluation)

    def check_live(self, what):
        raise NotImplementedError

    def realize(self):
        """-> the five ordinary autograd tensors (x_l_hat, x_a_hat, x_v_hat, y_hat, reg) of this forward"""
        raise NotImplementedError

    def backward_weighted(self, coef, labels, terms):
        """gradients of sum_k coef[k] * term_k into the model's flat gradient buffer; `terms`: which k appear at all"""
        raise NotImplementedError

    def host_slots(self):
        """-> list of 5 floats [disc, mse_l, mse_a, mse_v, reg] as the device holds them now (synchronises)"""
        raise NotImplementedError

    def device_slots(self):
        """-> 1-d device tensor of (at least) the 5 loss slots"""
        raise NotImplementedError

    # ------------------------------------------------------------------ shared logic
    def slice_of(self, term):
        d_l, d_a, d_v = self.dims
        lo = (0, d_l, d_l + d_a)[term - 1]
        return lo, lo + (d_l, d_a, d_v)[term - 1]

    def aliases_batch(self, target, term):
        """does `target` provably alias x[:, :, lo:hi] of the batch the forward ran on?"""
        x = self.x
        if type(target) is not torch.Tensor or target.requires_grad or target.dtype != x.dtype or target.device != x.device:
            return False
        lo, hi = self.slice_of(term)
        T, B, D = x.shape
        return (tuple(target.shape) == (T, B, hi - lo) and target.stride() == x.stride()
                and target.data_ptr() == x.data_ptr() + lo * x.element_size() and x._version == self.x_version)

    def disc_real(self, y_real):
        kind, labels, squeezed = self.disc
        inp = y_real.squeeze(1) if squeezed else y_real
        return F.l1_loss(inp, labels) if kind == 0 else F.cross_entropy(inp, labels)

    def term_real(self, k):
        """term k re-evaluated with torch ops on the materialised outputs"""
        r = self.realize()
        if k == REG:
            return r[4]
        if k == DISC:
            return self.disc_real(r[3])
        return F.mse_loss(r[k - 1], self.targets[k])

    def term_value(self, k):
        """term k as a detached device scalar without materialising anything"""
        if k == DISC and not self.disc_in_slot:
            with torch.no_grad():
                return self.disc_real(self.views[3])
        return self.device_slots()[k]


def _swap(v):
    if isinstance(v, (LazyOut, LossExpr)):
        return v._real()
    if isinstance(v, (list, tuple)):
        return type(v)(_swap(u) for u in v)
    if isinstance(v, dict):
        return {k: _swap(u) for k, u in v.items()}
    return v


# metadata reads that must not materialise anything: answered from the underlying view
_META = set()
for _n in ("shape", "device", "dtype", "ndim", "is_cuda", "requires_grad", "layout", "is_leaf", "grad_fn", "grad", "_version",
           "names", "is_sparse", "is_quantized", "is_meta", "output_nr", "_base", "is_cpu", "itemsize", "nbytes"):
    _p = getattr(torch.Tensor, _n, None)
    if _p is not None and hasattr(_p, "__get__"):
        _META.add(_p.__get__)
for _n in ("size", "dim", "numel", "nelement", "stride", "is_contiguous", "data_ptr", "element_size", "__len__",
           "is_floating_point", "is_complex", "get_device", "storage_offset", "ndimension", "type", "is_pinned", "is_shared",
           "is_same_size", "has_names", "__hash__", "_is_view", "is_inference", "__reduce_ex__", "__deepcopy__"):
    _p = getattr(torch.Tensor, _n, None)
    if _p is not None:
        _META.add(_p)
_META.discard(None)
for _n in ("__reduce_ex__", "__deepcopy__", "type"):         # (these read data or build new tensors: through the fallback)
    _META.discard(getattr(torch.Tensor, _n, None))


class LazyOut(torch.Tensor):
    """x_l_hat / x_a_hat / x_v_hat / y_hat of a training-mode forward: a view of the plan's output buffer that waits for its
    criterion (module docstring)."""

    @staticmethod
    def __new__(cls, view, step, idx, squeezed=False):
        t = torch.Tensor._make_subclass(cls, view, True)
        t._step, t._idx, t._squeezed = step, idx, squeezed
        return t

    def _real(self):
        r = self._step.realize()[self._idx]
        return r.squeeze(1) if self._squeezed else r

    # --- the three criteria
    def _mse(self, target, size_average=None, reduce=None, reduction="mean", weight=None):
        st = self._step
        if (self._idx > 2 or self._squeezed or reduction != "mean" or size_average is not None or reduce is not None
                or weight is not None or st.real is not None or not st.aliases_batch(target, self._idx + 1)):
            return NotImplemented
        st.check_live("a lazy output")
        if st.targets is None:
            st.targets = {}
        st.targets[self._idx + 1] = target
        return LossExpr(st, {self._idx + 1: 1.0})

    def _labels_ok(self, target, kind):
        st = self._step
        if self._idx != 3 or st.real is not None or st.loss_kind != kind or type(target) is not torch.Tensor:
            return False
        if target.requires_grad or target.device != st.x.device or not target.is_contiguous():
            return False
        with _DisableTF():
            shp = tuple(self.shape)
        if kind == 0:
            return target.dtype == torch.float32 and tuple(target.shape) == shp
        return (not self._squeezed) and target.dtype == torch.int64 and tuple(target.shape) == shp[:1]

    def _claim_disc(self, kind, target, squeezed):
        """y_hat meets its criterion: the step has ONE discriminative slot, so only one (criterion, labels, squeeze) combination
        can stay symbolic.  A second, different one (another label tensor, the other criterion) goes through the ordinary path --
        the outputs materialise and both losses are evaluated by torch -- instead of silently re-pointing the first."""
        st = self._step
        if st.disc is not None and not (st.disc[0] == kind and st.disc[1] is target and st.disc[2] == squeezed):
            return False
        st.check_live("a lazy output")
        st.disc = (kind, target, squeezed)
        return True

    def _l1(self, target, size_average=None, reduce=None, reduction="mean", weight=None):
        if reduction != "mean" or size_average is not None or reduce is not None or weight is not None \
                or not self._labels_ok(target, 0) or not self._claim_disc(0, target, self._squeezed):
            return NotImplemented
        return LossExpr(self._step, {DISC: 1.0})

    def _ce(self, target, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean",
            label_smoothing=0.0):
        if (reduction != "mean" or size_average is not None or reduce is not None or weight is not None
                or ignore_index != -100 or label_smoothing != 0.0 or not self._labels_ok(target, 1)
                or not self._claim_disc(1, target, False)):
            return NotImplemented
        return LossExpr(self._step, {DISC: 1.0})

    def _squeeze(self, *dims, **kw):
        if self._idx != 3 or kw or len(dims) != 1 or not isinstance(dims[0], int) or self._squeezed:
            return NotImplemented
        with _DisableTF():
            nd, shp = self.dim(), tuple(self.shape)
        d = dims[0] + nd if dims[0] < 0 else dims[0]
        if nd != 2 or d != 1:
            return NotImplemented
        if shp[1] != 1:
            return self                     # squeeze of a dimension that is not 1 is the identity (output_dim 3: mfm_you.py:478)
        with _DisableTF():
            v = torch.Tensor.squeeze(self, 1)
        return LazyOut(v, self._step, 3, True)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _META:
            with _DisableTF():
                return func(*args, **kwargs)
        h = _LAZY_HANDLERS.get(func)
        if h is not None and isinstance(args[0], LazyOut):
            r = h(*args, **kwargs)
            if r is not NotImplemented:
                return r
        return func(*_swap(args), **_swap(kwargs))


_LAZY_HANDLERS = {
    F.mse_loss: LazyOut._mse,
    F.l1_loss: LazyOut._l1,
    F.cross_entropy: LazyOut._ce,
    torch.Tensor.squeeze: LazyOut._squeeze,
    torch.squeeze: LazyOut._squeeze,
}


class LossExpr(torch.Tensor):
    """const + sum_k coef[k] * slot_k of ONE forward (k: 0 disc, 1..3 mse_l/a/v, 4 regulariser); module docstring."""

    @staticmethod
    def __new__(cls, step, coef, const=0.0):
        t = torch.Tensor._make_subclass(cls, step.scalar_view, True)
        t._step, t._coef, t._const = step, coef, const
        return t

    # ------------------------------------------------------------------ symbolic arithmetic
    def _scaled(self, f):
        f = float(f)
        return LossExpr(self._step, {k: c * f for k, c in self._coef.items()}, self._const * f)

    def _plus(self, o, sign):
        if _is_num(o):
            return LossExpr(self._step, self._coef, self._const + sign * float(o))
        if isinstance(o, LossExpr) and o._step is self._step:
            coef = dict(self._coef)
            for k, c in o._coef.items():
                coef[k] = coef.get(k, 0.0) + sign * c
            return LossExpr(self._step, coef, self._const + sign * o._const)
        return None

    def __mul__(self, o):
        return self._scaled(o) if _is_num(o) else torch.mul(self._real(), _swap(o))
    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._scaled(1.0 / float(o)) if _is_num(o) else torch.div(self._real(), _swap(o))

    def __rtruediv__(self, o):
        return torch.div(_swap(o), self._real())

    def __neg__(self):
        return self._scaled(-1.0)

    def __pos__(self):
        return self

    def __add__(self, o):
        r = self._plus(o, 1.0)
        return r if r is not None else torch.add(self._real(), _swap(o))
    __radd__ = __add__

    def __sub__(self, o):
        r = self._plus(o, -1.0)
        return r if r is not None else torch.sub(self._real(), _swap(o))

    def __rsub__(self, o):
        r = self._scaled(-1.0)._plus(o, 1.0)
        return r if r is not None else torch.sub(_swap(o), self._real())

    # ------------------------------------------------------------------ the two things the loop does with a loss
    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        st = self._step
        if gradient is None and not retain_graph and not create_graph and inputs is None and st.real is None \
                and self._fast_backward_ok():
            st.check_live("a loss expression")
            # (the labels ride along whenever y_hat met its criterion -- also under a stage loss without the discriminative
            # term: the backward then still leaves L_disc in slot 0 for `disc_loss.item()`)
            labels = st.disc[1] if st.disc is not None else None
            st.backward_weighted(self._coef, labels, tuple(self._coef))
            st.disc_in_slot = labels is not None
            return None
        return self._real().backward(gradient, retain_graph, create_graph, inputs)

    def _fast_backward_ok(self):
        co, st = self._coef, self._step
        if not co:
            return False
        has = [k in co for k in (GEN_L, GEN_A, GEN_V)]
        if any(has):
            # the forward baked lda_m into d x_hat_m: all three reconstruction terms with the plan's own weights (or all with
            # weight 0: zero gradients, like torch).  Anything else is autograd's
            if not all(has):
                return False
            g = [co[k] for k in (GEN_L, GEN_A, GEN_V)]
            if any(v != 0.0 for v in g) and any(abs(v - l) > 1e-6 * abs(l) for v, l in zip(g, st.lda)):
                return False
        return not (DISC in co and st.disc is None)

    def item(self):
        st = self._step
        st.check_live("a loss expression")
        if DISC in self._coef and not st.disc_in_slot:
            v = self._const
            for k, c in self._coef.items():
                v += c * float(st.term_value(k))
            return v
        s = st.host_slots()
        return self._const + sum(c * s[k] for k, c in self._coef.items())

    def __float__(self):
        return float(self.item())

    def tolist(self):
        return self.item()

    def _value(self):
        """detached device scalar (kernels only: capturable)"""
        st = self._step
        st.check_live("a loss expression")
        v = None
        for k, c in self._coef.items():
            t = st.term_value(k) * c
            v = t if v is None else v + t
        if v is None:
            v = torch.zeros((), device=st.x.device)
        return v + self._const if self._const != 0.0 else v.clone()

    def detach(self):
        return self._value()

    def _real(self):
        st = self._step
        v = None
        for k, c in self._coef.items():
            t = st.term_real(k)
            t = t if c == 1.0 else c * t
            v = t if v is None else v + t
        if v is None:
            return torch.full((), self._const, device=st.x.device, requires_grad=True)
        return v + self._const if self._const != 0.0 else v

    def __repr__(self):
        names = ("disc", "mse_l", "mse_a", "mse_v", "reg")
        return "LossExpr(%s%s)" % (" + ".join("%g*%s" % (c, names[k]) for k, c in sorted(self._coef.items())),
                                    (" + %g" % self._const) if self._const else "")

    def __format__(self, spec):
        return format(self.item(), spec) if spec else repr(self)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _META:
            with _DisableTF():
                return func(*args, **kwargs)
        if func is torch.Tensor.detach or func is torch.detach:
            return args[0]._value()
        return func(*_swap(args), **_swap(kwargs))


class SnapshotStep(StepBase):
    """Loss expressions over a COPY of a plan's loss slots (values only, nothing to back-propagate): what a captured training
    step hands out -- one copy node inside the graph keeps them valid across interleaved eager forwards on the same plan
    (train.GraphedModuleStep)."""
    disc_in_slot = True

    def __init__(self, slots, x):
        self.slots, self.x = slots, x
        self.scalar_view = slots[7]

    def check_live(self, what):
        pass

    def term_real(self, k):
        return self.slots[k]

    def term_value(self, k):
        return self.slots[k]

    def backward_weighted(self, coef, labels, terms):
        raise RuntimeError("this loss belongs to a captured training step: its backward already ran inside the graph")

    def host_slots(self):
        return [float(v) for v in self.slots[:5].cpu()]

    def device_slots(self):
        return self.slots


# ---------------------------------------------------------------------------------------- the engine binding
class PlanStep(StepBase):
    """StepBase on the fused plan of an engine-backed module (mfm_model._FusedEngineMixin)."""

    def __init__(self, module, eng, plan, x):
        self.module, self.eng, self.plan, self.x = module, eng, plan, x
        self.serial = plan.fwd_serial
        self.x_version = x._version
        st = plan.__dict__.get("_lazy_static")
        if st is None:               # (per plan, not per step)
            c = eng.cfg
            st = plan._lazy_static = (tuple(c["input_dims"]), (float(c["lda_xl"]), float(c["lda_xa"]), float(c["lda_xv"])),
                                      1 if c.get("loss", "l1") == "ce" else 0)
        self.dims, self.lda, self.loss_kind = st
        self.views = plan.out_views
        self.scalar_view = plan.loss0d

    def check_live(self, what):
        p = self.plan
        if self.eng.plan(p.T, p.B) is not p or p.fwd_serial != self.serial:
            raise RuntimeError("%s of an EARLIER forward was used after another forward with the same (T=%d, B=%d) ran on this "
                               "model: the fused plan keeps the outputs of its latest forward only.  Read / back-propagate each "
                               "forward's outputs before the next forward, or set model.lazy_losses = False to get ordinary "
                               "tensors back" % (what, p.T, p.B))

    def realize(self):
        if self.real is None:
            self.check_live("a lazy output")
            from . import mfm_model
            # (grad mode ON whatever the caller's is: the materialised outputs are cached for the rest of the step, and a read under
            #  torch.no_grad() -- an accuracy / NaN check in front of loss.backward() -- must not leave them without a grad_fn)
            with torch.enable_grad():
                self.real = mfm_model._LazyRealFn.apply(self.module._flat_leaf, self)
        return self.real

    def backward_weighted(self, coef, labels, terms):
        from . import mfm_model
        mfm_model._lazy_backward(self, coef, labels, terms)

    def host_slots(self):
        st = self.plan.state.detach().cpu().numpy()
        if self.eng.check_status((self.plan, st), raise_on_error=False):
            warnings.warn(self.eng.status_message(), RuntimeWarning, stacklevel=3)
        return [float(v) for v in st[:5]]

    def device_slots(self):
        return self.plan.losses
