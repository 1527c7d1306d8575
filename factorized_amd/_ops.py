"""Granular autograd ops of the module path on libmfm_hip.so (no PyTorch fallback): whole-sequence LSTM encoders / decoders and their
grouped forms, Linear layers on the HIP GEMM (single / grouped), the MFN memory recurrence, the MMD kernel.  `mfm_model.py` builds the
reference's classes from them; the ablation / missing-modality classes (`mfm_extra.py`) are compositions of exactly these."""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import engine as E


class _MMDFn(torch.autograd.Function):
    """loss_MMD value and gradient wrt z in one HIP kernel (mfm_mmd_fwd_bwd, csrc/mmd.hip)."""

    @staticmethod
    def forward(ctx, z, gauss):
        zc, gc = z.detach().contiguous().float(), gauss.detach().contiguous().float()
        B, dim = zc.shape
        loss = torch.zeros((), device=z.device)
        dz = torch.empty_like(zc)
        _lib.check(_lib.lib().mfm_mmd_fwd_bwd(zc.data_ptr(), gc.data_ptr(), B, dim, loss.data_ptr(), dz.data_ptr(),
                                              E._stream()), "mfm_mmd_fwd_bwd")
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, dl):
        (dz,) = ctx.saved_tensors
        return dz * dl, None


def _require_cuda(t, what):
    if not t.is_cuda:
        raise _lib.MfmError("%s: input is on %s; factorized_amd runs on the MI355X HIP library only "
                            "(no CPU fallback)" % (what, t.device))
    _lib.lib()


def _hp(h):
    return (h + 15) // 16 * 16


_ONES = {}


def _ones(n, dev):
    """Cached all-ones vector (the bias column sums are GEMMs against it): one fill per size, not per call."""
    key = (int(n), str(dev))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(int(n), device=dev)
    return t


def _zeros_many(dev, *shapes):
    """Several zero-filled tensors carved from ONE allocation (one fill launch instead of len(shapes)); each
    starts 64-float aligned."""
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    offs, cur = [], 0
    for n in sizes:
        offs.append(cur)
        cur += (n + 63) // 64 * 64
    flat = torch.zeros(max(cur, 1), device=dev)
    return [flat[o:o + n].view(sh) for o, n, sh in zip(offs, sizes, shapes)]


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
        g_fcw, g_fcb, g_wih, g_whh, g_bih, g_bhh = _zeros_many(dev, fc_w.shape, (n_out,), w_ih.shape, w_hh.shape,
                                                               (4 * h,), (4 * h,))
        dh_last = torch.empty(B, h, device=dev)
        ones = _ones(max(T * B, B), dev)
        E.gemm_grouped([
            E.make_gemm(d_out, fc_w, dh_last, B, h, n_out, a_sm=n_out, a_sk=1, b_sk=h, b_sn=1, ldc=h),
            E.make_gemm(d_out, h_last, g_fcw, n_out, h, B, a_sm=1, a_sk=n_out, b_sk=Hp, b_sn=1, ldc=h,
                        accumulate=1, split_k=0),
            E.make_gemm(d_out, ones, g_fcb, n_out, 1, B, a_sm=1, a_sk=n_out, b_sk=1, b_sn=1, ldc=1,
                        accumulate=1, split_k=0)])
        E.lstm_seq([E.make_seq(gates, hs, cs, w_hh, h, dh_ext=dh_last, ld_dh=h)], T, B, backward=True)
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
        g_fcw, g_fcb, g_wih, g_whh, g_bih, g_bhh = _zeros_many(dev, fc_w.shape, (d,), w_ih.shape, w_hh.shape,
                                                               (4 * h,), (4 * h,))
        ones = _ones(T * B, dev)
        E.gemm_grouped([
            E.make_gemm(d_out, fc_w, dhs, T * B, Hp, d, a_sm=d, a_sk=1, b_sk=h, b_sn=1, ldc=Hp, n_valid=h),
            E.make_gemm(d_out, hs, g_fcw, d, h, T * B, a_sm=1, a_sk=d, b_sk=Hp, b_sn=1, ldc=h,
                        accumulate=1, split_k=0),
            E.make_gemm(d_out, ones, g_fcb, d, 1, T * B, a_sm=1, a_sk=d, b_sk=1, b_sn=1, ldc=1,
                        accumulate=1, split_k=0)])
        d_hT = torch.empty(B, h, device=dev)
        E.lstm_seq([E.make_seq(gates, hs, cs, w_hh, h, w_ih=w_ih, b_ih=b_ih, b_hh=b_hh, h_init=hT, is_dec=True,
                               dh_ext=dhs, ld_dh=Hp, d_h_init=d_hT)], T, B, backward=True)
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


class _DecoderGroupFn(torch.autograd.Function):
    """n independent decoderLSTMs (7 tensors each: hT, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b) on the same T in
    shared launches: one mfm_lstm_seq_* call and one grouped fc1 GEMM per direction (the three modality
    decoders of MFM / MFM_KL, reference mfm_model.py:547-549)."""

    @staticmethod
    def forward(ctx, n, t, *args):
        T = int(t)
        dev = args[0].device
        seqs, heads, saved, dims, outs = [], [], [], [], []
        for i in range(n):
            hT, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b = args[7 * i:7 * i + 7]
            B, h = hT.shape
            Hp, d = _hp(h), fc_w.shape[0]
            hT = hT.contiguous().float()
            gates = torch.empty(T, B, 4, Hp, device=dev, dtype=torch.float32)
            hs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
            cs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
            out = torch.empty(T, B, d, device=dev, dtype=torch.float32)
            seqs.append(E.make_seq(gates, hs, cs, w_hh, h, w_ih=w_ih, b_ih=b_ih, b_hh=b_hh, h_init=hT, is_dec=True))
            heads.append(E.make_gemm(hs, fc_w, out, T * B, d, h, a_sm=Hp, a_sk=1, b_sk=1, b_sn=h, ldc=d, bias=fc_b))
            saved += [hT, w_ih, w_hh, b_ih, b_hh, fc_w, gates, hs, cs]
            dims.append((B, d, h, Hp))
            outs.append(out)
        for i in range(0, n, 4):
            E.lstm_seq(seqs[i:i + 4], T, dims[0][0])
        E.gemm_grouped(heads)
        ctx.save_for_backward(*saved)
        ctx.dims = (T, tuple(dims))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        saved = ctx.saved_tensors
        T, dims = ctx.dims
        dev = saved[0].device
        pre, seqs, post, grads, keep = [], [], [], [], []
        for i, (B, d, h, Hp) in enumerate(dims):
            hT, w_ih, w_hh, b_ih, b_hh, fc_w, gates, hs, cs = saved[9 * i:9 * i + 9]
            d_out = douts[i]
            d_out = torch.zeros(T, B, d, device=dev) if d_out is None else d_out.contiguous().float()
            dhs = torch.empty(T, B, Hp, device=dev)
            d_hT = torch.empty(B, h, device=dev)
            g_fcw, g_fcb, g_wih, g_whh, g_bih, g_bhh = _zeros_many(dev, fc_w.shape, (d,), w_ih.shape, w_hh.shape,
                                                                   (4 * h,), (4 * h,))
            ones = _ones(T * B, dev)
            keep += [d_out, dhs]
            pre += [E.make_gemm(d_out, fc_w, dhs, T * B, Hp, d, a_sm=d, a_sk=1, b_sk=h, b_sn=1, ldc=Hp, n_valid=h),
                    E.make_gemm(d_out, hs, g_fcw, d, h, T * B, a_sm=1, a_sk=d, b_sk=Hp, b_sn=1, ldc=h,
                                accumulate=1, split_k=0),
                    E.make_gemm(d_out, ones, g_fcb, d, 1, T * B, a_sm=1, a_sk=d, b_sk=1, b_sn=1, ldc=1,
                                accumulate=1, split_k=0)]
            seqs.append(E.make_seq(gates, hs, cs, w_hh, h, w_ih=w_ih, b_ih=b_ih, b_hh=b_hh, h_init=hT, is_dec=True,
                                   dh_ext=dhs, ld_dh=Hp, d_h_init=d_hT))
            post += [E.make_gemm(gates, hT, g_wih, h, h, B, a_sm=1, a_sk=4 * Hp, b_sk=h, b_sn=1, ldc=h,
                                 batch=4, a_sz=Hp, c_sz=h * h, accumulate=1, split_k=0),
                     E.make_gemm(gates, ones, g_bih, h, 1, T * B, a_sm=1, a_sk=4 * Hp, b_sk=1, b_sn=1, ldc=1,
                                 batch=4, a_sz=Hp, c_sz=h, accumulate=1, split_k=0, c2=g_bhh)]
            if T > 1:
                # steps >= 1 feed h back as the input (mfm_model.py:85): the same product goes to both
                post.append(E.make_gemm(gates[1:], hs, g_whh, h, h, (T - 1) * B, a_sm=1, a_sk=4 * Hp, b_sk=Hp,
                                        b_sn=1, ldc=h, batch=4, a_sz=Hp, c_sz=h * h, accumulate=1, split_k=0,
                                        c2=g_wih))
            grads += [d_hT, g_wih, g_whh, g_bih, g_bhh, g_fcw, g_fcb]
        E.gemm_grouped(pre)
        for i in range(0, len(seqs), 4):
            E.lstm_seq(seqs[i:i + 4], T, dims[0][0], backward=True)
        E.gemm_grouped(post)
        del keep
        return (None, None) + tuple(grads)


def decoder_group(pairs, t):
    """[(hT, decoderLSTM), ...] -> [x_hat, ...] in shared launches (all on the same batch size)."""
    if os.environ.get("MFM_NO_SEQ_GROUP") or len({p[0].shape[0] for p in pairs}) != 1:
        return [m.forward(hT, t) for hT, m in pairs]
    args = []
    for hT, m in pairs:
        args += [hT, m.lstm.weight_ih, m.lstm.weight_hh, m.lstm.bias_ih, m.lstm.bias_hh, m.fc1.weight, m.fc1.bias]
    return list(_DecoderGroupFn.apply(len(pairs), t, *args))



# ----------------------------------------------------------------------------------- Linear on the HIP GEMM
class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on mfm_gemm_grouped_f32 (forward NT, backward NN for dx and TN for dW, db)."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous().float()
        M, K = x2.shape
        N = w.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        E.gemm_grouped([E.make_gemm(x2, w, y, M, N, K, a_sm=K, a_sk=1, b_sk=1, b_sn=K, ldc=N, bias=b)])
        ctx.save_for_backward(x2, w)
        ctx.shp = shp
        return y.reshape(shp[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(M, N).contiguous().float()
        dev = dy.device
        dx = torch.empty(M, K, device=dev)
        dw, db = _zeros_many(dev, w.shape, (N,))
        ones = _ones(M, dev)
        E.gemm_grouped([
            E.make_gemm(dy2, w, dx, M, K, N, a_sm=N, a_sk=1, b_sk=K, b_sn=1, ldc=K),
            E.make_gemm(dy2, x2, dw, N, K, M, a_sm=1, a_sk=N, b_sk=K, b_sn=1, ldc=K, accumulate=1, split_k=0),
            E.make_gemm(dy2, ones, db, N, 1, M, a_sm=1, a_sk=N, b_sk=1, b_sn=1, ldc=1, accumulate=1, split_k=0)])
        return dx.reshape(ctx.shp), dw, db


class _MemFn(torch.autograd.Function):
    """MFN memory recurrence (reference mfm_model.py:177-181) on mfm_mfn_mem_fwd/bwd: one launch per
    direction for all T steps; weight gradients as grouped GEMMs over the saved tensors."""

    _calls = 0
    _replay_counter = {}       # device -> int64[1]: advanced inside a captured graph, added to the seed by the kernel

    @staticmethod
    def supported(M, H1, H2):
        def p2(x):
            p = 1
            while p < x:
                p <<= 1
            return p
        qa, qb = p2(-(-M // 32)), p2(-(-max(H1, H2) // 32))
        return qa <= 16 and qb <= 16 and (H1 + H2) * qa <= 1024 and M * qb <= 1024

    @staticmethod
    def _desc(T, B, M, H1, H2, a1, a2, chat, w1m, w2m, w1b, b1b, w2b, b2b, gam1, gam2, mems, p1, p2, train, seed,
              mem_out=None, dmem=None, du1=None, du2=None, dchat=None):
        d = _lib.MemDesc()
        d.a1, d.a2, d.chat = a1.data_ptr(), a2.data_ptr(), chat.data_ptr()
        d.w1m, d.w2m = w1m.data_ptr(), w2m.data_ptr()
        d.w1b, d.b1b, d.w2b, d.b2b = w1b.data_ptr(), b1b.data_ptr(), w2b.data_ptr(), b2b.data_ptr()
        d.gam1, d.gam2, d.mems = gam1.data_ptr(), gam2.data_ptr(), mems.data_ptr()
        d.mem_out = mem_out.data_ptr() if mem_out is not None else None
        d.dmem_out = dmem.data_ptr() if dmem is not None else None
        d.du1 = du1.data_ptr() if du1 is not None else None
        d.du2 = du2.data_ptr() if du2 is not None else None
        d.dchat = dchat.data_ptr() if dchat is not None else None
        d.T, d.B, d.M, d.H1, d.H2 = T, B, M, H1, H2
        d.train, d.p1, d.p2, d.seed = int(train), float(p1), float(p2), int(seed)
        return d

    @staticmethod
    def forward(ctx, g1_att, g2_att, chat, w1m, w2m, w1b, b1b, w2b, b2b, p1, p2, train):
        T, B, H1 = g1_att.shape
        H2, M = g2_att.shape[2], chat.shape[2]
        dev = chat.device
        f = lambda t: t.detach().contiguous().float()
        a1, a2 = f(g1_att).clone(), f(g2_att).clone()         # overwritten with the activations
        chat, w1m, w2m, w1b, b1b, w2b, b2b = map(f, (chat, w1m, w2m, w1b, b1b, w2b, b2b))
        gam1, gam2, mems = (torch.empty(T, B, M, device=dev) for _ in range(3))
        mem_out = torch.empty(B, M, device=dev)
        _MemFn._calls += 1
        seed = (torch.initial_seed() * 0x9E3779B1 + _MemFn._calls) & 0xFFFFFFFFFFFF
        d = _MemFn._desc(T, B, M, H1, H2, a1, a2, chat, w1m, w2m, w1b, b1b, w2b, b2b, gam1, gam2, mems, p1, p2, train,
                         seed, mem_out=mem_out)
        if train and (p1 > 0 or p2 > 0):
            # a replayed hipGraph re-runs this launch with the same host seed: under capture, add a device word
            # that the graph itself advances, so every replay draws new masks
            ctr = _MemFn._replay_counter.get(dev)
            capturing = torch.cuda.is_current_stream_capturing()
            if ctr is None:
                if capturing:
                    raise RuntimeError("MFN memory kernel: run one eager training step before capturing a graph "
                                       "(its replay counter cannot be allocated inside the capture)")
                ctr = _MemFn._replay_counter[dev] = torch.zeros(1, dtype=torch.int64, device=dev)
            if capturing:
                ctr.add_(0x1E3779B97F4A7C15)
                d.seed_dev = ctr.data_ptr()
        _lib.check(_lib.lib().mfm_mfn_mem_fwd(C.byref(d), E._stream()), "mfm_mfn_mem_fwd")
        ctx.save_for_backward(a1, a2, chat, w1m, w2m, w1b, b1b, w2b, b2b, gam1, gam2, mems)
        ctx.cfg = (T, B, M, H1, H2, p1, p2, train, seed)
        return mem_out

    @staticmethod
    def backward(ctx, dmem):
        a1, a2, chat, w1m, w2m, w1b, b1b, w2b, b2b, gam1, gam2, mems = ctx.saved_tensors
        T, B, M, H1, H2, p1, p2, train, seed = ctx.cfg
        dev = chat.device
        dz1, dz2 = gam1.clone(), gam2.clone()                 # turned into pre-sigmoid gradients in place
        du1 = torch.empty(T, B, H1, device=dev)
        du2 = torch.empty(T, B, H2, device=dev)
        dchat = torch.empty(T, B, M, device=dev)
        dm = dmem.contiguous().float()
        d = _MemFn._desc(T, B, M, H1, H2, a1, a2, chat, w1m, w2m, w1b, b1b, w2b, b2b, dz1, dz2, mems, p1, p2, train, seed,
                         dmem=dm, du1=du1, du2=du2, dchat=dchat)
        _lib.check(_lib.lib().mfm_mfn_mem_bwd(C.byref(d), E._stream()), "mfm_mfn_mem_bwd")
        TB = T * B
        dw1b, dw2b, db1b, db2b, dw1m, dw2m = _zeros_many(dev, w1b.shape, w2b.shape, b1b.shape, b2b.shape, w1m.shape,
                                                         w2m.shape)
        ones = _ones(TB, dev)
        g = [
            # gamma_n_fc2: dW[m, j] = sum_r dz[r, m] a[r, j] ; db[m] = sum_r dz[r, m]
            E.make_gemm(dz1, a1, dw1b, M, H1, TB, a_sm=1, a_sk=M, b_sk=H1, b_sn=1, ldc=H1, accumulate=1, split_k=0),
            E.make_gemm(dz2, a2, dw2b, M, H2, TB, a_sm=1, a_sk=M, b_sk=H2, b_sn=1, ldc=H2, accumulate=1, split_k=0),
            E.make_gemm(dz1, ones, db1b, M, 1, TB, a_sm=1, a_sk=M, b_sk=1, b_sn=1, ldc=1, accumulate=1, split_k=0),
            E.make_gemm(dz2, ones, db2b, M, 1, TB, a_sm=1, a_sk=M, b_sk=1, b_sn=1, ldc=1, accumulate=1, split_k=0),
        ]
        if T > 1:
            # memory columns of gamma_n_fc1: dW[j, m] = sum_{t>=1,b} du[t,b,j] mem_{t-1}[b,m]
            g += [E.make_gemm(du1[1:], mems, dw1m, H1, M, TB - B, a_sm=1, a_sk=H1, b_sk=M, b_sn=1, ldc=M, accumulate=1, split_k=0),
                  E.make_gemm(du2[1:], mems, dw2m, H2, M, TB - B, a_sm=1, a_sk=H2, b_sk=M, b_sn=1, ldc=M, accumulate=1, split_k=0)]
        E.gemm_grouped(g)
        return du1, du2, dchat, dw1m, dw2m, dw1b, db1b, dw2b, db2b, None, None, None


class _GroupLinearFn(torch.autograd.Function):
    """n independent Linears (their own inputs, weights, biases) as ONE grouped GEMM launch forward and one
    backward: the factorized model has 4-8 of them side by side at three places (mu/logvar heads, z->f fc1,
    fc2), and at these sizes a launch plus an autograd node cost more than the product."""

    @staticmethod
    def forward(ctx, n, *args):
        xs, ws, bs = args[:n], args[n:2 * n], args[2 * n:3 * n]
        x2s, ys, descs = [], [], []
        for x, w, b in zip(xs, ws, bs):
            x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
            M, K = x2.shape
            N = w.shape[0]
            y = torch.empty(M, N, device=x.device, dtype=torch.float32)
            descs.append(E.make_gemm(x2, w, y, M, N, K, a_sm=K, a_sk=1, b_sk=1, b_sn=K, ldc=N, bias=b))
            x2s.append(x2); ys.append(y)
        E.gemm_grouped(descs)
        ctx.n = n
        ctx.shapes = [tuple(x.shape) for x in xs]
        ctx.save_for_backward(*x2s, *ws)
        return tuple(y.reshape(shp[:-1] + (y.shape[1],)) for y, shp in zip(ys, ctx.shapes))

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        x2s, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dev = x2s[0].device
        zs = _zeros_many(dev, *([tuple(w.shape) for w in ws] + [(w.shape[0],) for w in ws]))
        dws, dbs = zs[:n], zs[n:]
        dxs, descs, keep = [], [], []      # `keep`: the descriptors hold raw pointers of per-iteration temporaries
        for i in range(n):
            x2, w = x2s[i], ws[i]
            M, K = x2.shape
            N = w.shape[0]
            dy2 = dys[i].reshape(M, N).contiguous().float() if dys[i] is not None else torch.zeros(M, N, device=dev)
            keep.append(dy2)
            dx = torch.empty(M, K, device=dev)
            dxs.append(dx)
            descs += [E.make_gemm(dy2, w, dx, M, K, N, a_sm=N, a_sk=1, b_sk=K, b_sn=1, ldc=K),
                      E.make_gemm(dy2, x2, dws[i], N, K, M, a_sm=1, a_sk=N, b_sk=K, b_sn=1, ldc=K, accumulate=1, split_k=0),
                      E.make_gemm(dy2, _ones(M, dev), dbs[i], N, 1, M, a_sm=1, a_sk=N, b_sk=1, b_sn=1, ldc=1, accumulate=1, split_k=0)]
        E.gemm_grouped(descs)
        del keep
        return (None,) + tuple(dx.reshape(shp) for dx, shp in zip(dxs, ctx.shapes)) + tuple(dws) + tuple(dbs)


def linear_group(pairs):
    """[(x, HipLinear), ...] -> [y, ...] in one launch."""
    if os.environ.get("MFM_NO_GROUP_LINEAR"):          # A/B timing only
        return [l(x) for x, l in pairs]
    n = len(pairs)
    xs = [x for x, _ in pairs]
    return list(_GroupLinearFn.apply(n, *xs, *[l.weight for _, l in pairs], *[l.bias for _, l in pairs]))


class HipLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose matmuls run on the HIP GEMM."""

    def forward(self, x):
        _require_cuda(x, "Linear.forward")
        return _LinearFn.apply(x, self.weight, self.bias)


# ----------------------------------------------------------------------------------- MFN
class _SeqGroupFn(torch.autograd.Function):
    """Several independent sequence encoders in the same launches: `kinds[i]` is "enc" (x -> fc1(h_T), the
    encoderLSTM contract, 7 tensors) or "states" (x -> (h_T, c_all), the MFN LSTMs, 5 tensors).  One grouped
    GEMM for all input projections, one mfm_lstm_seq_* call for all recurrences (the library packs up to
    four per launch), one grouped GEMM for the fc1 heads; backward likewise (MFM / MFM_KL run three
    encoders and the three MFN LSTMs on the same batch, reference mfm_model.py:745-756, 163-169)."""

    @staticmethod
    def forward(ctx, kinds, *args):
        specs, pos = [], 0
        for k in kinds:
            n = 7 if k == "enc" else 5
            specs.append((k,) + tuple(args[pos:pos + n]))
            pos += n
        T, B = args[0].shape[0], args[0].shape[1]
        dev = args[0].device
        proj, seqs, heads, saved, dims, outs = [], [], [], [], [], []
        for sp in specs:
            k, x, w_ih, w_hh, b_ih, b_hh = sp[:6]
            d, h = x.shape[2], w_hh.shape[1]
            Hp = _hp(h)
            xr, ldx = _rows(x)
            gates = torch.empty(T, B, 4, Hp, device=dev, dtype=torch.float32)
            hs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
            cs = torch.empty(T, B, Hp, device=dev, dtype=torch.float32)
            proj.append(E.make_gemm(xr, w_ih, gates, T * B, Hp, d, a_sm=ldx, a_sk=1, b_sk=1, b_sn=d, ldc=4 * Hp,
                                    bias=b_ih, bias2=b_hh, n_valid=h, batch=4, b_sz=h * d, c_sz=Hp, bias_sz=h))
            seqs.append(E.make_seq(gates, hs, cs, w_hh, h))
            if k == "enc":
                fc_w, fc_b = sp[6], sp[7]
                out = torch.empty(B, fc_w.shape[0], device=dev, dtype=torch.float32)
                heads.append(E.make_gemm(hs[T - 1], fc_w, out, B, fc_w.shape[0], h, a_sm=Hp, a_sk=1, b_sk=1, b_sn=h,
                                         ldc=fc_w.shape[0], bias=fc_b))
                outs.append(out)
                saved += [xr, w_ih, w_hh, gates, hs, cs, fc_w]
            else:
                saved += [xr, w_ih, w_hh, gates, hs, cs]
            dims.append((k, d, h, Hp, ldx, bool(x.requires_grad)))
        E.gemm_grouped(proj)
        for i in range(0, len(seqs), 4):                   # MFM_MAX_SEQ recurrences per launch
            E.lstm_seq(seqs[i:i + 4], T, B)
        if heads:
            E.gemm_grouped(heads)
        res, it, si = [], iter(outs), 0
        for (k, d, h, Hp, ldx, _) in dims:
            if k == "enc":
                res.append(next(it))
                si += 7
            else:
                hs, cs = saved[si + 4], saved[si + 5]
                res += [hs[T - 1, :, :h].clone(), cs[:, :, :h].clone()]
                si += 6
        ctx.save_for_backward(*saved)
        ctx.dims = (T, B, tuple(dims))
        return tuple(res)

    @staticmethod
    def backward(ctx, *douts):
        saved = ctx.saved_tensors
        T, B, dims = ctx.dims
        dev = saved[0].device
        ones = _ones(max(T * B, B), dev)
        pre, seqs, post, grads = [], [], [], []
        keep = []          # the descriptors hold raw pointers: temporaries must outlive the launches below
        si, gi = 0, 0
        for (k, d, h, Hp, ldx, need_dx) in dims:
            if k == "enc":
                xr, w_ih, w_hh, gates, hs, cs, fc_w = saved[si:si + 7]
                si += 7
                d_out = douts[gi]
                gi += 1
                n_out = fc_w.shape[0]
                d_out = torch.zeros(B, n_out, device=dev) if d_out is None else d_out.contiguous().float()
                g_fcw, g_fcb, g_wih, g_whh, g_bih, g_bhh = _zeros_many(dev, fc_w.shape, (n_out,), w_ih.shape, w_hh.shape,
                                                                       (4 * h,), (4 * h,))
                dh = torch.empty(B, h, device=dev)
                pre += [E.make_gemm(d_out, fc_w, dh, B, h, n_out, a_sm=n_out, a_sk=1, b_sk=h, b_sn=1, ldc=h),
                        E.make_gemm(d_out, hs[T - 1], g_fcw, n_out, h, B, a_sm=1, a_sk=n_out, b_sk=Hp, b_sn=1, ldc=h,
                                    accumulate=1, split_k=0),
                        E.make_gemm(d_out, ones, g_fcb, n_out, 1, B, a_sm=1, a_sk=n_out, b_sk=1, b_sn=1, ldc=1,
                                    accumulate=1, split_k=0)]
                seqs.append(E.make_seq(gates, hs, cs, w_hh, h, dh_ext=dh, ld_dh=h))
                keep += [d_out, dh]
                tail = [g_fcw, g_fcb]
            else:
                xr, w_ih, w_hh, gates, hs, cs = saved[si:si + 6]
                si += 6
                d_hT, d_cs = douts[gi], douts[gi + 1]
                gi += 2
                g_wih, g_whh, g_bih, g_bhh = _zeros_many(dev, w_ih.shape, w_hh.shape, (4 * h,), (4 * h,))
                dh = torch.zeros(B, h, device=dev) if d_hT is None else d_hT.contiguous().float()
                dc = torch.zeros(T, B, Hp, device=dev)
                if d_cs is not None:
                    dc[:, :, :h] = d_cs
                seqs.append(E.make_seq(gates, hs, cs, w_hh, h, dh_ext=dh, ld_dh=h, dc_ext=dc))
                keep += [dh, dc]
                tail = []
            post += [E.make_gemm(gates, xr, g_wih, h, d, T * B, a_sm=1, a_sk=4 * Hp, b_sk=ldx, b_sn=1, ldc=d,
                                 batch=4, a_sz=Hp, c_sz=h * d, accumulate=1, split_k=0),
                     E.make_gemm(gates, ones, g_bih, h, 1, T * B, a_sm=1, a_sk=4 * Hp, b_sk=1, b_sn=1, ldc=1,
                                 batch=4, a_sz=Hp, c_sz=h, accumulate=1, split_k=0, c2=g_bhh)]
            if T > 1:
                post.append(E.make_gemm(gates[1:], hs, g_whh, h, h, (T - 1) * B, a_sm=1, a_sk=4 * Hp, b_sk=Hp,
                                        b_sn=1, ldc=h, batch=4, a_sz=Hp, c_sz=h * h, accumulate=1, split_k=0))
            dx = None
            if need_dx:
                # dx_t = dA_t W_ih, summed over the four gates (same product as _EncoderSeqFn.backward)
                dx = torch.zeros(T, B, d, device=dev)
                post.append(E.make_gemm(gates, w_ih, dx, T * B, d, h, a_sm=4 * Hp, a_sk=1, b_sk=d, b_sn=1, ldc=d,
                                        batch=4, a_sz=Hp, b_sz=h * d, c_sz=0, accumulate=1, split_k=1))
            grads += [dx, g_wih, g_whh, g_bih, g_bhh] + tail
        if pre:
            E.gemm_grouped(pre)
        for i in range(0, len(seqs), 4):
            E.lstm_seq(seqs[i:i + 4], T, B, backward=True)
        E.gemm_grouped(post)
        del keep
        return (None,) + tuple(grads)


def seq_group(encoders, state_lstms):
    """[(x, encoderLSTM)], [(x, nn.LSTMCell)] -> ([fc1(h_T)], [(h_T, c_all)]) through _SeqGroupFn."""
    kinds, args = [], []
    for x, m in encoders:
        kinds.append("enc")
        args += [x, m.lstm.weight_ih, m.lstm.weight_hh, m.lstm.bias_ih, m.lstm.bias_hh, m.fc1.weight, m.fc1.bias]
    for x, c in state_lstms:
        kinds.append("states")
        args += [x, c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh]
    res = _SeqGroupFn.apply(tuple(kinds), *args)
    ne = len(encoders)
    return list(res[:ne]), [(res[ne + 2 * i], res[ne + 2 * i + 1]) for i in range(len(state_lstms))]
