"""Drop-in mirror of the reference's `mfm_model.py` class surface on the MI355X HIP library.

    from factorized_amd.mfm_model import MFM_KL_EF, encoderLSTM, decoderLSTM, loss_KLD, loss_MMD

Same constructor signatures (six positional dicts, reference mfm_model.py:470/558/663), same
sub-module / parameter names (so `state_dict()` keys equal the reference's and reference
checkpoints `load_state_dict` cleanly), same `forward` contracts:

    encoderLSTM(d, h).forward(x[T,B,d])            -> [B,h]                (mfm_model.py:40-62)
    decoderLSTM(h, d).forward(hT[B,h], t)          -> [t,B,d]              (mfm_model.py:64-91)
    MFM_KL_EF(...).forward(x[T,B,D])               -> ([x_l_hat,x_a_hat,x_v_hat,y_hat], kld, 0.0)
                                                                           (mfm_model.py:619-660)

Modules are ordinary `nn.Module`s: `.train()/.eval()/.parameters()`, `optim.Adam(model.parameters())`
and `loss.backward()` of a reference-style driver work unchanged.  Underneath, every forward and
backward runs on libmfm_hip.so (autograd.Function wrappers); there is NO PyTorch fallback -- CPU
tensors or a missing library raise.  `.cuda()` calls inside the reference's forwards
(mfm_model.py:51-52,76-77) are replaced by "allocate on x.device".

The fastest way to train is not this autograd path but `factorized_amd.engine.MFMEngine.train_step`
(one C call per step); `MFM_KL_EF.engine` exposes it on the same parameter storage.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from . import engine as E


# ----------------------------------------------------------------------------------- losses
def loss_KLD(mu, logvar):
    """Sum (not mean) KL divergence to N(0,1) -- reference mfm_model.py:36-38."""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())


def compute_kernel(x, y):
    """Gaussian kernel with the reference's double division by dim -- mfm_model.py:14-23."""
    dim = x.size(1)
    diff = x.unsqueeze(1) - y.unsqueeze(0)
    return torch.exp(-(diff.pow(2).mean(2) / float(dim)))


def loss_MMD(zy, gauss=None):
    """MMD between zy and a N(0,1) sample of the same shape -- mfm_model.py:25-34.  The reference
    draws the sample on the host; pass `gauss` to inject it (parity tests)."""
    if gauss is None:
        gauss = torch.randn(zy.size(), device=zy.device, dtype=zy.dtype)
    return compute_kernel(gauss, gauss).mean() + compute_kernel(zy, zy).mean() \
        - 2.0 * compute_kernel(gauss, zy).mean()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise _lib.MfmError("%s: input is on %s; factorized_amd runs on the MI355X HIP library only "
                            "(no CPU fallback)" % (what, t.device))
    _lib.lib()


def _hp(h):
    return (h + 15) // 16 * 16


def _rows(x):
    """x [T,B,d] whose last dim is contiguous and whose (t,b) rows have ONE uniform stride
    (a column slice of a contiguous [T,B,D] batch, reference mfm_model.py:620-622) ->
    (tensor, row_stride).  Anything else is made contiguous."""
    T, B, d = x.shape
    if x.stride(2) == 1 and x.stride(0) == B * x.stride(1) and x.dtype == torch.float32:
        return x, x.stride(1)
    x = x.contiguous().float()
    return x, d


# ----------------------------------------------------------------------------------- encoder
class _EncoderSeqFn(torch.autograd.Function):
    """x -> fc1(h_T): input projection GEMM + whole-sequence recurrence + fc1 GEMM."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b):
        T, B, d = x.shape
        h = w_hh.shape[1]
        Hp = _hp(h)
        xr, ldx = _rows(x)
        dev = x.device
        gates = torch.empty(T, B, 4, Hp, device=dev, dtype=torch.float32)
        hs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
        cs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
        E.gemm_grouped([E.make_gemm(xr, w_ih, gates, T * B, Hp, d, a_sm=ldx, a_sk=1, b_sk=1, b_sn=d,
                                    ldc=4 * Hp, bias=b_ih, bias2=b_hh, n_valid=h, batch=4, b_sz=h * d,
                                    c_sz=Hp, bias_sz=h)])
        E.lstm_seq([E.make_seq(gates, hs, cs, w_hh, h)], T, B)
        out = torch.empty(B, fc_w.shape[0], device=dev, dtype=torch.float32)
        h_last = hs[T - 1]
        E.gemm_grouped([E.make_gemm(h_last, fc_w, out, B, fc_w.shape[0], h, a_sm=Hp, a_sk=1, b_sk=1, b_sn=h,
                                    ldc=fc_w.shape[0], bias=fc_b)])
        ctx.save_for_backward(xr, w_ih, w_hh, fc_w, gates, hs, cs)
        ctx.dims = (T, B, d, h, Hp, ldx, x.requires_grad, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, d_out):
        xr, w_ih, w_hh, fc_w, gates, hs, cs = ctx.saved_tensors
        T, B, d, h, Hp, ldx, need_dx, xshape = ctx.dims
        dev = d_out.device
        d_out = d_out.contiguous()
        n_out = fc_w.shape[0]
        h_last = hs[T - 1]
        g_fcw = torch.zeros_like(fc_w); g_fcb = torch.zeros(n_out, device=dev)
        dh_last = torch.empty(B, h, device=dev)
        ones = torch.ones(max(T * B, B), device=dev)
        E.gemm_grouped([
            E.make_gemm(d_out, fc_w, dh_last, B, h, n_out, a_sm=n_out, a_sk=1, b_sk=h, b_sn=1, ldc=h),
            E.make_gemm(d_out, h_last, g_fcw, n_out, h, B, a_sm=1, a_sk=n_out, b_sk=Hp, b_sn=1, ldc=h,
                        accumulate=1, split_k=0),
            E.make_gemm(d_out, ones, g_fcb, n_out, 1, B, a_sm=1, a_sk=n_out, b_sk=1, b_sn=1, ldc=1,
                        accumulate=1, split_k=0)])
        E.lstm_seq([E.make_seq(gates, hs, cs, w_hh, h, dh_ext=dh_last, ld_dh=h)], T, B, backward=True)
        g_wih = torch.zeros_like(w_ih); g_whh = torch.zeros_like(w_hh)
        g_bih = torch.zeros(4 * h, device=dev); g_bhh = torch.zeros(4 * h, device=dev)
        descs = [E.make_gemm(gates, xr, g_wih, h, d, T * B, a_sm=1, a_sk=4 * Hp, b_sk=ldx, b_sn=1, ldc=d,
                             batch=4, a_sz=Hp, c_sz=h * d, accumulate=1, split_k=0),
                 E.make_gemm(gates, ones, g_bih, h, 1, T * B, a_sm=1, a_sk=4 * Hp, b_sk=1, b_sn=1, ldc=1,
                             batch=4, a_sz=Hp, c_sz=h, accumulate=1, split_k=0, c2=g_bhh)]
        if T > 1:
            descs.append(E.make_gemm(gates[1:], hs, g_whh, h, h, (T - 1) * B, a_sm=1, a_sk=4 * Hp, b_sk=Hp,
                                     b_sn=1, ldc=h, batch=4, a_sz=Hp, c_sz=h * h, accumulate=1, split_k=0))
        dx = None
        if need_dx:
            dx = torch.zeros(T, B, d, device=dev)
            descs.append(E.make_gemm(gates, w_ih, dx, T * B, d, h, a_sm=4 * Hp, a_sk=1, b_sk=d, b_sn=1, ldc=d,
                                     batch=4, a_sz=Hp, b_sz=h * d, c_sz=0, accumulate=1, split_k=1))
        E.gemm_grouped(descs)
        return dx, g_wih, g_whh, g_bih, g_bhh, g_fcw, g_fcb


class encoderLSTM(nn.Module):
    def __init__(self, d, h):
        super(encoderLSTM, self).__init__()
        self.lstm = nn.LSTMCell(d, h)     # parameter container: weight_ih/hh, bias_ih/hh (torch init)
        self.fc1 = nn.Linear(h, h)
        self.h = h

    def forward(self, x):
        _require_cuda(x, "encoderLSTM.forward")
        return _EncoderSeqFn.apply(x, self.lstm.weight_ih, self.lstm.weight_hh, self.lstm.bias_ih,
                                   self.lstm.bias_hh, self.fc1.weight, self.fc1.bias)


# ----------------------------------------------------------------------------------- decoder
class _DecoderSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hT, t, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b):
        B, h = hT.shape
        T = int(t)
        Hp = _hp(h)
        d = fc_w.shape[0]
        dev = hT.device
        hT = hT.contiguous().float()
        gates = torch.empty(T, B, 4, Hp, device=dev, dtype=torch.float32)
        hs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
        cs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
        E.lstm_seq([E.make_seq(gates, hs, cs, w_hh, h, w_ih=w_ih, b_ih=b_ih, b_hh=b_hh, h_init=hT, is_dec=True)],
                   T, B)
        out = torch.empty(T, B, d, device=dev, dtype=torch.float32)
        E.gemm_grouped([E.make_gemm(hs, fc_w, out, T * B, d, h, a_sm=Hp, a_sk=1, b_sk=1, b_sn=h, ldc=d,
                                    bias=fc_b)])
        ctx.save_for_backward(hT, w_ih, w_hh, b_ih, b_hh, fc_w, gates, hs, cs)
        ctx.dims = (T, B, d, h, Hp)
        return out

    @staticmethod
    def backward(ctx, d_out):
        hT, w_ih, w_hh, b_ih, b_hh, fc_w, gates, hs, cs = ctx.saved_tensors
        T, B, d, h, Hp = ctx.dims
        dev = d_out.device
        d_out = d_out.contiguous()
        dhs = torch.empty(T, B, Hp, device=dev)
        g_fcw = torch.zeros_like(fc_w); g_fcb = torch.zeros(d, device=dev)
        ones = torch.ones(T * B, device=dev)
        E.gemm_grouped([
            E.make_gemm(d_out, fc_w, dhs, T * B, Hp, d, a_sm=d, a_sk=1, b_sk=h, b_sn=1, ldc=Hp, n_valid=h),
            E.make_gemm(d_out, hs, g_fcw, d, h, T * B, a_sm=1, a_sk=d, b_sk=Hp, b_sn=1, ldc=h,
                        accumulate=1, split_k=0),
            E.make_gemm(d_out, ones, g_fcb, d, 1, T * B, a_sm=1, a_sk=d, b_sk=1, b_sn=1, ldc=1,
                        accumulate=1, split_k=0)])
        d_hT = torch.empty(B, h, device=dev)
        E.lstm_seq([E.make_seq(gates, hs, cs, w_hh, h, w_ih=w_ih, b_ih=b_ih, b_hh=b_hh, h_init=hT, is_dec=True,
                               dh_ext=dhs, ld_dh=Hp, d_h_init=d_hT)], T, B, backward=True)
        g_wih = torch.zeros_like(w_ih); g_whh = torch.zeros_like(w_hh)
        g_bih = torch.zeros(4 * h, device=dev); g_bhh = torch.zeros(4 * h, device=dev)
        descs = [E.make_gemm(gates, hT, g_wih, h, h, B, a_sm=1, a_sk=4 * Hp, b_sk=h, b_sn=1, ldc=h,
                             batch=4, a_sz=Hp, c_sz=h * h, accumulate=1, split_k=0),
                 E.make_gemm(gates, ones, g_bih, h, 1, T * B, a_sm=1, a_sk=4 * Hp, b_sk=1, b_sn=1, ldc=1,
                             batch=4, a_sz=Hp, c_sz=h, accumulate=1, split_k=0, c2=g_bhh)]
        if T > 1:
            # steps >= 1 feed h back as the input (mfm_model.py:85): the same product goes to both
            descs.append(E.make_gemm(gates[1:], hs, g_whh, h, h, (T - 1) * B, a_sm=1, a_sk=4 * Hp, b_sk=Hp,
                                     b_sn=1, ldc=h, batch=4, a_sz=Hp, c_sz=h * h, accumulate=1, split_k=0,
                                     c2=g_wih))
        E.gemm_grouped(descs)
        return d_hT, None, g_wih, g_whh, g_bih, g_bhh, g_fcw, g_fcb


class decoderLSTM(nn.Module):
    def __init__(self, h, d):
        super(decoderLSTM, self).__init__()
        self.lstm = nn.LSTMCell(h, h)
        self.fc1 = nn.Linear(h, d)
        self.d = d
        self.h = h

    def forward(self, hT, t):
        _require_cuda(hT, "decoderLSTM.forward")
        return _DecoderSeqFn.apply(hT, t, self.lstm.weight_ih, self.lstm.weight_hh, self.lstm.bias_ih,
                                   self.lstm.bias_hh, self.fc1.weight, self.fc1.bias)


# ----------------------------------------------------------------------------------- MFM_KL_EF
class _KLEFFn(torch.autograd.Function):
    """The whole MFM_KL_EF forward as ONE plan call; backward = mfm_plan_backward_ext with the
    upstream gradients autograd hands us (any user loss)."""

    @staticmethod
    def forward(ctx, x, module, *params):
        eng = module.engine
        out = eng.forward(x, None, train=module.training, want_xhat=True)
        kld = out["losses"][4].clone()
        ctx.module = module
        ctx.save_for_backward(x)
        return out["x_l_hat"], out["x_a_hat"], out["x_v_hat"], out["y_hat"], kld

    @staticmethod
    def backward(ctx, d_xl, d_xa, d_xv, d_y, d_kld):
        (x,) = ctx.saved_tensors
        module = ctx.module
        eng = module.engine
        T, B, _ = x.shape
        d_l, d_a, d_v = eng.cfg["input_dims"]
        dev = x.device

        def z(t, shape):
            return torch.zeros(shape, device=dev) if t is None else t.contiguous().float()
        d_xl, d_xa, d_xv = z(d_xl, (T, B, d_l)), z(d_xa, (T, B, d_a)), z(d_xv, (T, B, d_v))
        d_y = z(d_y, (B, eng.cfg["output_dim"]))
        d_kld = z(d_kld, ()).reshape(1)
        eng.backward_ext(x, d_xl, d_xa, d_xv, d_y, d_kld)
        gv = eng.grad_views()
        return (None, None) + tuple(gv[n].clone() for n in module._param_names)


class MFM_KL_EF(nn.Module):
    def __init__(self, config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig):
        super(MFM_KL_EF, self).__init__()
        self._configs = [config, NN1Config, NN2Config, gamma1Config, gamma2Config, outConfig]
        [self.d_l, self.d_a, self.d_v] = config["input_dims"]
        zy, zl, za, zv = config['zy_size'], config['zl_size'], config['za_size'], config['zv_size']
        fy, fl, fa, fv = config['fy_size'], config['fl_size'], config['fa_size'], config['fv_size']
        output_dim = config['output_dim']
        self.encoder_l = encoderLSTM(self.d_l, zl)
        self.encoder_a = encoderLSTM(self.d_a, za)
        self.encoder_v = encoderLSTM(self.d_v, zv)
        self.decoder_l = decoderLSTM(fy + fl, self.d_l)
        self.decoder_a = decoderLSTM(fy + fa, self.d_a)
        self.decoder_v = decoderLSTM(fy + fv, self.d_v)
        last_ef_size = zl + za + zv
        self.ef_encoder = encoderLSTM(self.d_l + self.d_a + self.d_v, last_ef_size)
        self.last_to_zy_fc1 = nn.Linear(last_ef_size, zy)
        self.last_to_logvarzy_fc1 = nn.Linear(last_ef_size, zy)
        self.last_to_zl_fc1 = nn.Linear(zl, zl)
        self.last_to_za_fc1 = nn.Linear(za, za)
        self.last_to_zv_fc1 = nn.Linear(zv, zv)
        self.last_to_logvarzl_fc1 = nn.Linear(zl, zl)
        self.last_to_logvarza_fc1 = nn.Linear(za, za)
        self.last_to_logvarzv_fc1 = nn.Linear(zv, zv)
        self.zy_to_fy_fc1 = nn.Linear(zy, fy)
        self.zy_to_fy_fc2 = nn.Linear(fy, fy)
        self.zy_to_fy_dropout = nn.Dropout(config['zy_to_fy_dropout'])
        self.zl_to_fl_fc1 = nn.Linear(zl, fl)
        self.zl_to_fl_fc2 = nn.Linear(fl, fl)
        self.zl_to_fl_dropout = nn.Dropout(config['zl_to_fl_dropout'])
        self.za_to_fa_fc1 = nn.Linear(za, fa)
        self.za_to_fa_fc2 = nn.Linear(fa, fa)
        self.za_to_fa_dropout = nn.Dropout(config['za_to_fa_dropout'])
        self.zv_to_fv_fc1 = nn.Linear(zv, fv)
        self.zv_to_fv_fc2 = nn.Linear(fv, fv)
        self.zv_to_fv_dropout = nn.Dropout(config['zv_to_fv_dropout'])
        self.fy_to_y_fc1 = nn.Linear(fy, fy)
        self.fy_to_y_fc2 = nn.Linear(fy, output_dim)
        self.fy_to_y_dropout = nn.Dropout(config['fy_to_y_dropout'])
        self._param_names = [n for n, _ in self.named_parameters()]
        self._engine = None

    # ---- flat storage: every parameter becomes a view into the engine's flat buffer
    def _flat_ok(self):
        if self._engine is None:
            return False
        views = self._engine.param_views()
        first, last = self._param_names[0], self._param_names[-1]
        pd = dict(self.named_parameters())
        return (pd[first].data_ptr() == views[first].data_ptr()
                and pd[last].data_ptr() == views[last].data_ptr())

    def _adopt(self, device):
        cfg = dict(self._configs[0])
        for k, dflt in (("lda_xl", 1.0), ("lda_xa", 1.0), ("lda_xv", 1.0), ("lda_mmd", 1.0)):
            cfg.setdefault(k, dflt)
        eng = E.MFMEngine([cfg] + list(self._configs[1:]), device=device)
        assert list(eng.layout.shapes.keys()) == self._param_names, "parameter naming drifted from the reference"
        pd = OrderedDict(self.named_parameters())
        eng.load_weights(OrderedDict((n, p.detach()) for n, p in pd.items()))
        views = eng.param_views()
        for n, p in pd.items():
            p.data = views[n]
        self._engine = eng

    @property
    def engine(self):
        """The fused engine sharing this module's parameter storage (build on first CUDA use)."""
        if not self._flat_ok():
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise _lib.MfmError("MFM_KL_EF: parameters are on %s; move the model to the GPU first" % dev)
            self._adopt(dev)
        return self._engine

    def forward(self, x):
        _require_cuda(x, "MFM_KL_EF.forward")
        if not (x.dtype == torch.float32 and x.is_contiguous()):
            x = x.contiguous().float()
        _ = self.engine
        pd = dict(self.named_parameters())
        x_l_hat, x_a_hat, x_v_hat, y_hat, kld = _KLEFFn.apply(x, self, *[pd[n] for n in self._param_names])
        decoded = [x_l_hat, x_a_hat, x_v_hat, y_hat]
        missing_loss = 0.0
        return decoded, kld, missing_loss


class MFN(nn.Module):
    """Memory Fusion Network encoder (reference mfm_model.py:93-199): scheduled, see DESIGN.md."""

    def __init__(self, *a, **k):
        super(MFN, self).__init__()
        raise NotImplementedError("factorized_amd: MFN (and MFM / MFM_KL which embed it) is not built yet; "
                                  "MFM_KL_EF is the implemented model (SURVEY.md section 8f)")


class MFM(nn.Module):
    def __init__(self, *a, **k):
        super(MFM, self).__init__()
        raise NotImplementedError("factorized_amd: MFM needs the MFN encoder, not built yet (use MFM_KL_EF)")


class MFM_KL(nn.Module):
    def __init__(self, *a, **k):
        super(MFM_KL, self).__init__()
        raise NotImplementedError("factorized_amd: MFM_KL needs the MFN encoder, not built yet (use MFM_KL_EF)")
