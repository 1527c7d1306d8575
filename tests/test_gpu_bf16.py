"""bf16 compute path (BASELINE.json configs 2-4; SURVEY.md section 8d: "bf16 compute, fp32 master weights / cell state
/ loss; gate: matched loss curve vs fp32, not 1e-4").

Two layers of evidence:
  * kernels against an EMULATION that rounds exactly what the kernels round (GEMM operands; W, h_{t-1}, dA_t of the
    recurrences) to bf16 with nearest-even and does everything else in fp64 -- a wrong fragment layout, a missing
    rounding or a transposed operand fails these at the 1e-3 level, far below bf16's own 4e-3 spacing;
  * the whole step against the fp32 reference: forward losses and gradients within stated bounds of the fp32 golden /
    oracle, and N-step loss curves that track the reference's own fp32 trace (goldens klef_b32_t20, klef_you_b32_t50,
    klef_mosei_b1024_t20)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import mfm_oracle as O
from factorized_amd import configs, synth
from tests import cases
from tests.cases import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    return engine


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def bf(a):
    """round to bf16 (nearest even) and back, as float64"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy().astype(np.float64)


def gemm_bf16(descs):
    from factorized_amd import _lib
    arr = (_lib.GemmDesc * len(descs))(*descs)
    _lib.check(_lib.lib().mfm_gemm_grouped_bf16(arr, len(descs), None), "mfm_gemm_grouped_bf16")


def seq_bf16(descs, T, B, backward=False):
    from factorized_amd import _lib
    arr = (_lib.SeqDesc * len(descs))(*descs)
    fn = _lib.lib().mfm_lstm_seq_bwd_bf16 if backward else _lib.lib().mfm_lstm_seq_fwd_bf16
    _lib.check(fn(arr, len(descs), T, B, None), "mfm_lstm_seq_*_bf16")


# ---------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("fr", ["1", "2", "4"])
@pytest.mark.parametrize("m,n,k", [(640, 128, 300), (37, 5, 11), (64, 64, 64), (1, 1, 1), (130, 70, 325), (2048, 96, 410)])
def test_bf16_gemm_nt_bias(eng, m, n, k, fr, monkeypatch):
    """forward product layout (x W^T + b: both operands k-contiguous), strided A rows, pad columns"""
    monkeypatch.setenv("MFM_GEMM_FR", fr)
    rs = np.random.RandomState(m + n + k)
    A = rs.normal(size=(m, k + 3)).astype(np.float32)
    W = rs.normal(size=(n, k)).astype(np.float32)
    b1 = rs.normal(size=n).astype(np.float32)
    a_d, w_d, b1_d = dev(A), dev(W), dev(b1)
    npad = n + 3
    c_d = torch.full((m, npad), 7.0, device="cuda")
    d = eng.make_gemm(a_d, w_d, c_d, m, npad, k, a_sm=k + 3, a_sk=1, b_sk=1, b_sn=k, ldc=npad, bias=b1_d, n_valid=n)
    gemm_bf16([d])
    ref = bf(A[:, :k]) @ bf(W).T + b1
    out = c_d.cpu().numpy()
    assert rel_err(out[:, :n], ref) < 1e-5
    assert np.all(out[:, n:] == 0.0)
    # and it really is bf16 arithmetic: the fp32 product differs at the 1e-3 level
    if k >= 64:
        full = A[:, :k].astype(np.float64) @ W.T.astype(np.float64) + b1
        assert 1e-4 < rel_err(out[:, :n], full) < 3e-2


@pytest.mark.parametrize("fr", ["1", "2", "4"])
def test_bf16_gemm_tn_splitk_accumulate_dual_output(eng, fr, monkeypatch):
    """weight-gradient layout dW = dA^T X: BOTH operands contiguous along m / n (transposed in registers), batched
    over the four gates, split-K with atomics, second output"""
    monkeypatch.setenv("MFM_GEMM_FR", fr)
    rs = np.random.RandomState(5)
    R, M, N = 650, 120, 325
    dA = rs.normal(size=(R, 4, 128)).astype(np.float32)
    X = rs.normal(size=(R, N)).astype(np.float32)
    da_d, x_d = dev(dA), dev(X)
    c1 = torch.zeros(4, M, N, device="cuda")
    c2 = torch.zeros(4, M, N, device="cuda")
    d = eng.make_gemm(da_d, x_d, c1, M, N, R, a_sm=1, a_sk=4 * 128, b_sk=N, b_sn=1, ldc=N, batch=4,
                      a_sz=128, c_sz=M * N, accumulate=1, split_k=0, c2=c2)
    gemm_bf16([d])
    ref = np.einsum("rgm,rn->gmn", bf(dA[:, :, :M]), bf(X))
    assert rel_err(c1.cpu().numpy(), ref) < 2e-5
    assert rel_err(c2.cpu().numpy(), ref) < 2e-5


def test_bf16_gemm_nn_group_and_column_sums(eng):
    """dH = dX W (B n-contiguous), a group of differently shaped problems in one launch, bias column sums against
    the ones vector (exact in bf16)"""
    rs = np.random.RandomState(9)
    descs, checks, keep = [], [], []
    for (m, n, k) in [(640, 104, 300), (640, 24, 5), (33, 16, 20), (7, 128, 64)]:
        A = rs.normal(size=(m, k)).astype(np.float32)
        W = rs.normal(size=(k, n)).astype(np.float32)
        a_d, w_d = dev(A), dev(W)
        c = torch.empty(m, n, device="cuda")
        descs.append(eng.make_gemm(a_d, w_d, c, m, n, k, a_sm=k, a_sk=1, b_sk=n, b_sn=1, ldc=n))
        checks.append((c, bf(A) @ bf(W)))
        keep += [a_d, w_d]
    R, M = 1500, 96
    G = rs.normal(size=(R, M)).astype(np.float32)
    g_d, ones = dev(G), torch.ones(R, device="cuda")
    cs = torch.zeros(M, device="cuda")
    descs.append(eng.make_gemm(g_d, ones, cs, M, 1, R, a_sm=1, a_sk=M, b_sk=1, b_sn=1, ldc=1, accumulate=1, split_k=0))
    gemm_bf16(descs)
    for c, ref in checks:
        assert rel_err(c.cpu().numpy(), ref) < 1e-5
    assert rel_err(cs.cpu().numpy(), bf(G).sum(0)) < 2e-5


# ---------------------------------------------------------------------------------- LSTM recurrences
def _sig(v):
    return 1.0 / (1.0 + np.exp(-v))


def _emulate_fwd(gx, Wb, h, T, B, dec_init=None, W0b=None, bias=None):
    """recurrence with h_{t-1} rounded to bf16 before the product; fp64 otherwise.  gx [T,B,4h] (encoder) or None
    (decoder: step 0 consumes dec_init through W0b, later steps the hidden state through Wb, plus bias)."""
    hh, cc = np.zeros((B, h)), np.zeros((B, h))
    gates, hs, cs = [], [], []
    for t in range(T):
        if dec_init is None:
            g = gx[t] + (bf(hh) @ Wb.T if t > 0 else 0.0)
        else:
            g = bias + (bf(dec_init) @ W0b.T if t == 0 else bf(hh) @ Wb.T)
        i, f, gg, o = _sig(g[:, :h]), _sig(g[:, h:2 * h]), np.tanh(g[:, 2 * h:3 * h]), _sig(g[:, 3 * h:])
        cc = f * cc + i * gg
        hh = o * np.tanh(cc)
        gates.append(np.stack([i, f, gg, o], 1)); hs.append(hh); cs.append(cc)
    return np.stack(gates), np.stack(hs), np.stack(cs)


def _emulate_bwd(gates, cs, Wb, T, B, h, dh_ext_all=None, dh_last=None, W0b=None):
    """BPTT with dA_t rounded to bf16 before dh_{t-1} = dA_t W; returns dA [T,B,4,h] and (decoder) d h_init."""
    dA = np.zeros((T, B, 4, h))
    dh_rec, dc = np.zeros((B, h)), np.zeros((B, h))
    for t in range(T - 1, -1, -1):
        dh = dh_rec.copy()
        if dh_ext_all is not None:
            dh += dh_ext_all[t]
        elif t == T - 1:
            dh += dh_last
        i, f, gg, o = gates[t, :, 0], gates[t, :, 1], gates[t, :, 2], gates[t, :, 3]
        tc = np.tanh(cs[t])
        cp = cs[t - 1] if t > 0 else np.zeros((B, h))
        dct = dh * o * (1 - tc * tc) + dc
        dA[t, :, 0] = dct * gg * i * (1 - i)
        dA[t, :, 1] = dct * cp * f * (1 - f)
        dA[t, :, 2] = dct * i * (1 - gg * gg)
        dA[t, :, 3] = dh * tc * o * (1 - o)
        dc = dct * f
        W = W0b if (t == 0 and W0b is not None) else Wb
        dh_rec = bf(dA[t].reshape(B, 4 * h)) @ W
    return dA, dh_rec


SEQ_SHAPES = [(8, 5, 1, 3), (8, 5, 32, 20), (24, 7, 33, 4), (32, 300, 32, 20), (80, 20, 17, 6),
              (104, 9, 16, 5), (120, 325, 32, 20), (20, 6, 5, 1), (128, 4, 3, 2), (36, 10, 40, 3), (120, 30, 300, 8)]


@pytest.mark.parametrize("h,d,B,T", SEQ_SHAPES)
def test_bf16_lstm_seq_encoder_fwd_bwd(eng, h, d, B, T):
    rs = np.random.RandomState(h * 7 + B)
    k = 1.0 / np.sqrt(h)
    w_ih = rs.uniform(-k, k, size=(4 * h, d)).astype(np.float32)
    w_hh = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    b = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    x = rs.normal(size=(T, B, d)).astype(np.float32)
    gx = (x.astype(np.float64) @ w_ih.T.astype(np.float64) + b).astype(np.float32)
    Hp = (h + 15) // 16 * 16
    gp = np.zeros((T, B, 4, Hp), dtype=np.float32)
    gp[:, :, :, :h] = gx.reshape(T, B, 4, h)
    gates = dev(gp)
    hs = torch.full((T, B, Hp), 9.0, device="cuda")
    cs = torch.full((T, B, Hp), 9.0, device="cuda")
    w_d = dev(w_hh)
    seq_bf16([eng.make_seq(gates, hs, cs, w_d, h)], T, B)
    g_ref, hs_ref, cs_ref = _emulate_fwd(gx.astype(np.float64), bf(w_hh), h, T, B)
    hs_o, cs_o = hs.cpu().numpy(), cs.cpu().numpy()
    assert rel_err(hs_o[:, :, :h], hs_ref) < 2e-3
    assert rel_err(cs_o[:, :, :h], cs_ref) < 2e-3
    assert rel_err(gates.cpu().numpy()[:, :, :, :h], g_ref) < 2e-3
    assert np.all(hs_o[:, :, h:] == 0.0) and np.all(cs_o[:, :, h:] == 0.0)
    # backward from the kernel's OWN saved activations (what the emulation is fed too)
    g_sav, c_sav = gates.cpu().numpy().astype(np.float64)[:, :, :, :h], cs_o.astype(np.float64)[:, :, :h]
    dh_last = rs.normal(size=(B, h)).astype(np.float32)
    dh_d = dev(dh_last)
    seq_bf16([eng.make_seq(gates, hs, cs, w_d, h, dh_ext=dh_d, ld_dh=h)], T, B, backward=True)
    dA_ref, _ = _emulate_bwd(g_sav, c_sav, bf(w_hh), T, B, h, dh_last=dh_last.astype(np.float64))
    dA = gates.cpu().numpy()
    assert rel_err(dA[:, :, :, :h], dA_ref) < 2e-3
    assert np.all(dA[:, :, :, h:] == 0.0)


@pytest.mark.parametrize("h,B,T", [(24, 32, 20), (104, 32, 20), (24, 5, 1), (40, 19, 3), (112, 33, 7), (104, 200, 6)])
def test_bf16_lstm_seq_decoder_fwd_bwd(eng, h, B, T):
    rs = np.random.RandomState(h + B + T)
    k = 1.0 / np.sqrt(h)
    w_ih = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    w_hh = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    b_ih = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    b_hh = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    init = rs.normal(size=(B, h)).astype(np.float32)
    Hp = (h + 15) // 16 * 16
    gates = torch.full((T, B, 4, Hp), 3.0, device="cuda")
    hs = torch.full((T, B, Hp), 9.0, device="cuda")
    cs = torch.full((T, B, Hp), 9.0, device="cuda")
    wi, wh, bi, bh, init_d = dev(w_ih), dev(w_hh), dev(b_ih), dev(b_hh), dev(init)
    seq_bf16([eng.make_seq(gates, hs, cs, wh, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init_d, is_dec=True)], T, B)
    Wsum = bf(w_ih + w_hh)                       # fp32 sum, ONE rounding -- as the kernel does
    g_ref, hs_ref, cs_ref = _emulate_fwd(None, Wsum, h, T, B, dec_init=init, W0b=bf(w_ih),
                                         bias=(b_ih + b_hh).astype(np.float64))
    hs_o, cs_o = hs.cpu().numpy(), cs.cpu().numpy()
    assert rel_err(hs_o[:, :, :h], hs_ref) < 2e-3
    assert rel_err(cs_o[:, :, :h], cs_ref) < 2e-3
    assert np.all(hs_o[:, :, h:] == 0.0)
    g_sav, c_sav = gates.cpu().numpy().astype(np.float64)[:, :, :, :h], cs_o.astype(np.float64)[:, :, :h]
    dH = rs.normal(size=(T, B, h)).astype(np.float32)
    dH_p = np.zeros((T, B, Hp), dtype=np.float32)
    dH_p[:, :, :h] = dH
    dh_d = dev(dH_p)
    dinit = torch.full((B, h), 5.0, device="cuda")
    seq_bf16([eng.make_seq(gates, hs, cs, wh, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init_d, is_dec=True,
                           dh_ext=dh_d, ld_dh=Hp, d_h_init=dinit)], T, B, backward=True)
    dA_ref, dinit_ref = _emulate_bwd(g_sav, c_sav, Wsum, T, B, h, dh_ext_all=dH.astype(np.float64), W0b=bf(w_ih))
    assert rel_err(gates.cpu().numpy()[:, :, :, :h], dA_ref) < 2e-3
    assert rel_err(dinit.cpu().numpy(), dinit_ref) < 2e-3


# ---------------------------------------------------------------------------------- bf16-RESIDENT saved activations
def bfd(a):
    """bf16 device tensor from an fp32 array"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().bfloat16()


def b2n(t):
    return t.float().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("h,d,B,T", [(8, 5, 32, 20), (24, 7, 33, 4), (32, 300, 32, 20), (80, 20, 17, 6), (120, 325, 40, 5), (36, 10, 300, 3)])
def test_bf16_resident_encoder_fwd_bwd(eng, h, d, B, T):
    """store_bf16: the x-projection is READ as bf16, activated gates / h are WRITTEN as bf16 (c stays fp32), h_{T-1} also
    as fp32; BPTT reads the bf16 gates and leaves dA as bf16 in place.  Reference: the emulation fed with exactly the
    stored (rounded) values -- what is left is the rounding of each output element (2^-9 relative)."""
    rs = np.random.RandomState(h * 7 + B)
    k = 1.0 / np.sqrt(h)
    w_hh = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    Hp = (h + 15) // 16 * 16
    gp = np.zeros((T, B, 4, Hp), dtype=np.float32)
    gp[:, :, :, :h] = rs.normal(size=(T, B, 4, h))
    gates = bfd(gp)
    gx = b2n(gates)[:, :, :, :h].reshape(T, B, 4 * h)          # what the kernel reads
    hs = torch.full((T, B, Hp), 9.0, device="cuda", dtype=torch.bfloat16)
    cs = torch.full((T, B, Hp), 9.0, device="cuda")
    h_last = torch.full((B, Hp), 7.0, device="cuda")
    w_d = dev(w_hh)
    seq_bf16([eng.make_seq(gates, hs, cs, w_d, h, store_bf16=True, h_last=h_last)], T, B)
    g_ref, hs_ref, cs_ref = _emulate_fwd(gx, bf(w_hh), h, T, B)
    hs_o, cs_o, g_o = b2n(hs), cs.cpu().numpy(), b2n(gates)
    assert rel_err(hs_o[:, :, :h], hs_ref) < 6e-3
    assert rel_err(cs_o[:, :, :h], cs_ref) < 2e-3
    assert rel_err(g_o[:, :, :, :h], g_ref) < 6e-3
    assert np.all(hs_o[:, :, h:] == 0.0) and np.all(cs_o[:, :, h:] == 0.0)
    hl = h_last.cpu().numpy()
    assert rel_err(hl[:, :h], hs_ref[-1]) < 2e-3 and np.all(hl[:, h:] == 0.0)
    assert np.array_equal(bf(hl), hs_o[-1])                     # hs[T-1] is the rounding of the fp32 copy
    # backward from the stored activations
    dh_last = rs.normal(size=(B, h)).astype(np.float32)
    seq_bf16([eng.make_seq(gates, hs, cs, w_d, h, dh_ext=dev(dh_last), ld_dh=h, store_bf16=True)], T, B, backward=True)
    dA_ref, _ = _emulate_bwd(g_o[:, :, :, :h], cs_o.astype(np.float64)[:, :, :h], bf(w_hh), T, B, h, dh_last=dh_last.astype(np.float64))
    dA = b2n(gates)
    assert rel_err(dA[:, :, :, :h], dA_ref) < 8e-3
    assert np.all(dA[:, :, :, h:] == 0.0)


@pytest.mark.parametrize("h,B,T", [(24, 32, 20), (104, 32, 20), (24, 5, 1), (112, 33, 7), (104, 200, 6)])
def test_bf16_resident_decoder_fwd_bwd(eng, h, B, T):
    """decoders: hs (the fc1 operand) bf16, and in the BPTT the per-step external gradient dh_ext [T,B,Hp] is a bf16
    buffer (written by the fc1-backward GEMM with c_bf16); d h_init stays fp32"""
    rs = np.random.RandomState(h + B + T)
    k = 1.0 / np.sqrt(h)
    w_ih = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    w_hh = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    b_ih = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    b_hh = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    init = rs.normal(size=(B, h)).astype(np.float32)
    Hp = (h + 15) // 16 * 16
    gates = torch.full((T, B, 4, Hp), 3.0, device="cuda", dtype=torch.bfloat16)
    hs = torch.full((T, B, Hp), 9.0, device="cuda", dtype=torch.bfloat16)
    cs = torch.full((T, B, Hp), 9.0, device="cuda")
    wi, wh, bi, bh, init_d = dev(w_ih), dev(w_hh), dev(b_ih), dev(b_hh), dev(init)
    seq_bf16([eng.make_seq(gates, hs, cs, wh, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init_d, is_dec=True, store_bf16=True)], T, B)
    Wsum = bf(w_ih + w_hh)
    g_ref, hs_ref, cs_ref = _emulate_fwd(None, Wsum, h, T, B, dec_init=init, W0b=bf(w_ih), bias=(b_ih + b_hh).astype(np.float64))
    hs_o, cs_o, g_o = b2n(hs), cs.cpu().numpy(), b2n(gates)
    assert rel_err(hs_o[:, :, :h], hs_ref) < 6e-3
    assert rel_err(cs_o[:, :, :h], cs_ref) < 2e-3
    assert np.all(hs_o[:, :, h:] == 0.0)
    dH = rs.normal(size=(T, B, h)).astype(np.float32)
    dH_p = np.zeros((T, B, Hp), dtype=np.float32)
    dH_p[:, :, :h] = dH
    dh_d = bfd(dH_p)
    dinit = torch.full((B, h), 5.0, device="cuda")
    seq_bf16([eng.make_seq(gates, hs, cs, wh, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init_d, is_dec=True,
                           dh_ext=dh_d, ld_dh=Hp, d_h_init=dinit, store_bf16=True)], T, B, backward=True)
    dA_ref, dinit_ref = _emulate_bwd(g_o[:, :, :, :h], cs_o.astype(np.float64)[:, :, :h], Wsum, T, B, h,
                                     dh_ext_all=b2n(dh_d)[:, :, :h], W0b=bf(w_ih))
    assert rel_err(b2n(gates)[:, :, :, :h], dA_ref) < 8e-3
    assert rel_err(dinit.cpu().numpy(), dinit_ref) < 8e-3


@pytest.mark.parametrize("fr", ["1", "2", "4"])
@pytest.mark.parametrize("m,n,k", [(640, 300, 104), (37, 5, 24), (2048, 20, 24), (130, 70, 128), (1, 1, 8), (5, 24, 5), (33, 104, 300)])
def test_bf16_resident_gemm_a_kcontig_and_c_out(eng, m, n, k, fr, monkeypatch):
    """A = bf16-resident hidden states [m, Kp] (k-contiguous, a_bf16): no rounding pass, bits go straight to LDS; the
    same product once with an fp32 C and once with a bf16 C (c_bf16: x-projection / dH of a bf16-resident plan)"""
    monkeypatch.setenv("MFM_GEMM_FR", fr)
    rs = np.random.RandomState(m + n + k)
    Kp = (k + 7) // 8 * 8          # rows padded to 16 bytes (d x_hat [rows, round_up(d, 8)]; K = 5: an odd element count)
    A = np.zeros((m, Kp), dtype=np.float32)
    A[:, :k] = rs.normal(size=(m, k))
    W = rs.normal(size=(n, k)).astype(np.float32)
    b1 = rs.normal(size=n).astype(np.float32)
    a_d, w_d, b1_d = bfd(A), dev(W), dev(b1)
    npad = n + 3
    c32 = torch.full((m, npad), 7.0, device="cuda")
    c16 = torch.full((m, npad), 7.0, device="cuda", dtype=torch.bfloat16)
    gemm_bf16([eng.make_gemm(a_d, w_d, c32, m, npad, k, a_sm=Kp, a_sk=1, b_sk=1, b_sn=k, ldc=npad, bias=b1_d, n_valid=n, a_bf16=True),
               eng.make_gemm(a_d, w_d, c16, m, npad, k, a_sm=Kp, a_sk=1, b_sk=1, b_sn=k, ldc=npad, bias=b1_d, n_valid=n, a_bf16=True, c_bf16=True)])
    ref = b2n(a_d)[:, :k] @ bf(W).T + b1
    out = c32.cpu().numpy()
    assert rel_err(out[:, :n], ref) < 1e-5
    assert np.all(out[:, n:] == 0.0)
    o16 = b2n(c16)
    assert np.array_equal(o16[:, :n], bf(out[:, :n]))            # the bf16 C is the rounding of the fp32 one
    assert np.all(o16[:, n:] == 0.0)


def test_bf16_resident_gemm_a_mcontig_small_product(eng):
    """dW_ih += dA_0^T h_init of the decoders (K = B rows only): dA is a bf16-resident, m-contiguous operand batched over
    the four gates; the generic kernel takes it on its slow path (8 two-byte LDS writes per load)"""
    rs = np.random.RandomState(11)
    R, h, Hp = 37, 24, 32
    dA = np.zeros((R, 4, Hp), dtype=np.float32)
    dA[:, :, :h] = rs.normal(size=(R, 4, h))
    X = rs.normal(size=(R, h)).astype(np.float32)
    da_d, x_d = bfd(dA), dev(X)
    c1 = torch.zeros(4, h, h, device="cuda")
    gemm_bf16([eng.make_gemm(da_d, x_d, c1, h, h, R, a_sm=1, a_sk=4 * Hp, b_sk=h, b_sn=1, ldc=h, batch=4,
                             a_sz=Hp, c_sz=h * h, accumulate=1, split_k=0, a_bf16=True)])
    ref = np.einsum("rgm,rn->gmn", b2n(da_d)[:, :, :h], bf(X))
    assert rel_err(c1.cpu().numpy(), ref) < 2e-5


def test_bf16_resident_flags_refused_by_fp32_entry_point(eng):
    from factorized_amd import _lib
    a, w = bfd(np.ones((16, 16))), dev(np.ones((16, 16)))
    c = torch.zeros(16, 16, device="cuda")
    d = eng.make_gemm(a, w, c, 16, 16, 16, a_sm=16, a_sk=1, b_sk=1, b_sn=16, ldc=16, a_bf16=True)
    arr = (_lib.GemmDesc * 1)(d)
    assert _lib.lib().mfm_gemm_grouped_f32(arr, 1, None) != 0


@pytest.mark.parametrize("h,dx,rows,shift", [(120, 325, 640, 32), (120, 410, 640, 32), (32, 300, 640, 32), (8, 5, 100, 5), (80, 20, 4580, 229),
                                             (104, 0, 640, 32), (24, 0, 37, 37), (120, 325, 20480, 1024), (36, 37, 171, 19),
                                             (8, 5, 2100, 5), (104, 0, 1500, 37)])
@pytest.mark.parametrize("mf", [3, 4])
def test_bf16_resident_weight_gradients_one_pass(eng, h, dx, rows, shift, mf, monkeypatch):
    """(mf: 96- / 128-column M-tiles, dw_stream_kernel<false, 3, 9> / <false, 4, 8>; dx = 410: a right-hand side of 544 columns,
    <false, 4, 9>.  mf = 4 also lets the launcher pick per item: h = 120 / 104 / 80 / 36 run 256-column tiles -- h = 120, dx = 325
    as two column parts -- in dw_stream_mixed_kernel, and the narrow items take 64 / 128 rows per chunk: rows = 2100 and 1500 end
    in a ragged chunk of that size)
    dw_bf16_kernel: dW_ih, dW_hh (+ the decoders' second target), db from bf16-resident dA / x / h in ONE pass: LDS-DMA
    slabs in memory order, transposing LDS reads, ragged row ranges and the t = 0 rows of h_{t-1} as zero-filled DMA lanes.
    Products of bf16 values are exact, so the fp64 reference over the stored values must match to fp32 summation error."""
    import ctypes as C
    from factorized_amd import _lib
    monkeypatch.setenv("MFM_DWB_MF", str(mf))
    rs = np.random.RandomState(h + dx + rows)
    Hp = (h + 15) // 16 * 16
    dA = np.zeros((rows, 4, Hp), dtype=np.float32)
    dA[:, :, :h] = rs.normal(size=(rows, 4, h))
    hs = np.zeros((rows, Hp), dtype=np.float32)
    hs[:, :h] = rs.normal(size=(rows, h))
    da_d, hs_d = bfd(dA), bfd(hs)
    ldx = (dx + 15) // 16 * 16 + 8 if dx else 0
    xb_d = None
    if dx:
        xb = rs.normal(size=(rows, ldx)).astype(np.float32)        # columns >= dx hold junk on purpose
        xb_d = bfd(xb)
    dw_ih = torch.full((4 * h, max(dx, 1)), 0.5, device="cuda")
    dw_hh = torch.full((4 * h, h), 0.25, device="cuda")
    dw_hh2 = torch.zeros(4 * h, h, device="cuda")
    db1, db2 = torch.zeros(4 * h, device="cuda"), torch.full((4 * h,), 1.0, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    _lib.check(_lib.lib().mfm_dw_bf16_lstm(p(da_d), rows, h, p(xb_d), ldx, dx, p(hs_d), shift, p(dw_ih) if dx else C.c_void_p(0),
                                           p(dw_hh), p(dw_hh2), p(db1), p(db2), None), "mfm_dw_bf16_lstm")
    A = b2n(da_d)[:, :, :h].reshape(rows, 4 * h)
    H = b2n(hs_d)[:, :h]
    ref_hh = A[shift:].T @ H[:rows - shift] if rows > shift else np.zeros((4 * h, h))
    scale = max(np.abs(ref_hh).max(), 1.0)
    assert np.abs(dw_hh.cpu().numpy() - 0.25 - ref_hh).max() < 2e-5 * scale
    assert np.abs(dw_hh2.cpu().numpy() - ref_hh).max() < 2e-5 * scale
    ref_b = A.sum(0)
    assert np.abs(db1.cpu().numpy() - ref_b).max() < 2e-5 * max(np.abs(ref_b).max(), 1.0)
    assert np.abs(db2.cpu().numpy() - 1.0 - ref_b).max() < 2e-5 * max(np.abs(ref_b).max(), 1.0)
    if dx:
        ref_ih = A.T @ b2n(xb_d)[:, :dx]
        assert np.abs(dw_ih.cpu().numpy() - 0.5 - ref_ih).max() < 2e-5 * max(np.abs(ref_ih).max(), 1.0)


def test_bf16_lstm_seq_four_in_one_launch_matches_single_launches(eng):
    """the four encoders of the plan (h = 32, 8, 80, 120) in one call == each alone, bit for bit"""
    rs = np.random.RandomState(3)
    T, B = 7, 37
    single, group, keep = [], [], []
    for h in (32, 8, 80, 120):
        Hp = (h + 15) // 16 * 16
        k = 1.0 / np.sqrt(h)
        w = dev(rs.uniform(-k, k, size=(4 * h, h)))
        gp = np.zeros((T, B, 4, Hp), dtype=np.float32)
        gp[:, :, :, :h] = rs.normal(size=(T, B, 4, h))
        outs = []
        for _ in range(2):
            g = dev(gp)
            hs, cs = torch.zeros(T, B, Hp, device="cuda"), torch.zeros(T, B, Hp, device="cuda")
            outs.append((g, hs, cs, eng.make_seq(g, hs, cs, w, h)))
        keep.append(w)
        single.append(outs[0]); group.append(outs[1])
    for g, hs, cs, d in single:
        seq_bf16([d], T, B)
    seq_bf16([d for _, _, _, d in group], T, B)
    for (g0, h0, c0, _), (g1, h1, c1, _) in zip(single, group):
        assert torch.equal(h0, h1) and torch.equal(c0, c1) and torch.equal(g0, g1)


@pytest.mark.parametrize("h,is_dec", [(120, False), (104, True), (24, True), (36, False), (8, False)])
def test_bf16_packed_weight_fragments_equal_in_kernel_gather(eng, h, is_dec):
    """mfm_lstm_pack_bf16 (what the plan runs once per step) must hand the recurrences exactly the fragments they
    would gather themselves: forward and backward results bit-identical with and without the pack."""
    from factorized_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(h)
    T, B = 5, 21
    Hp = (h + 15) // 16 * 16
    k = 1.0 / np.sqrt(h)
    w_ih, w_hh = dev(rs.uniform(-k, k, size=(4 * h, h))), dev(rs.uniform(-k, k, size=(4 * h, h)))
    b_ih, b_hh = dev(rs.uniform(-k, k, size=4 * h)), dev(rs.uniform(-k, k, size=4 * h))
    init = dev(rs.normal(size=(B, h)))
    gp = np.zeros((T, B, 4, Hp), dtype=np.float32)
    gp[:, :, :, :h] = rs.normal(size=(T, B, 4, h))
    dh_all = np.zeros((T, B, Hp), dtype=np.float32)
    dh_all[:, :, :h] = rs.normal(size=(T, B, h))
    nbytes = L.mfm_lstm_pack_bytes(h, int(is_dec))
    assert nbytes > 0 and L.mfm_lstm_pack_bytes(129, 0) == 0
    pack = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    res = []
    for use_pack in (False, True):
        gates = dev(gp)
        hs, cs = torch.zeros(T, B, Hp, device="cuda"), torch.zeros(T, B, Hp, device="cuda")
        kw = dict(w_pack=pack if use_pack else None)
        if is_dec:
            kw.update(w_ih=w_ih, b_ih=b_ih, b_hh=b_hh, h_init=init, is_dec=True)
        d = eng.make_seq(gates, hs, cs, w_hh, h, **kw)
        if use_pack:
            arr = (_lib.SeqDesc * 1)(d)
            _lib.check(L.mfm_lstm_pack_bf16(arr, 1, None), "mfm_lstm_pack_bf16")
        seq_bf16([d], T, B)
        fwd = (gates.clone(), hs.clone(), cs.clone())
        dinit = torch.zeros(B, h, device="cuda")
        if is_dec:
            d = eng.make_seq(gates, hs, cs, w_hh, h, dh_ext=dev(dh_all), ld_dh=Hp, d_h_init=dinit, **kw)
        else:
            d = eng.make_seq(gates, hs, cs, w_hh, h, dh_ext=dev(dh_all[-1, :, :h]), ld_dh=h, **kw)
        seq_bf16([d], T, B, backward=True)
        res.append(fwd + (gates.clone(), dinit.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------- the whole step
def _bf16_engine(cfgs):
    from factorized_amd import engine
    e = engine.MFMEngine(cfgs, precision="bf16")
    w = synth.make_weights(e.layout.shapes, seed=1234)
    e.load_weights(w)
    return e, w


@pytest.fixture(params=["all-bf16", "default", "default+dot2", "all-bf16+panel", "all-bf16+panel160", "all-bf16+panel96", "all-bf16+proj128", "all-bf16+proj64", "all-bf16+fc1r16", "all-bf16+tnsplit", "all-bf16+dwmf4"])
def seq_policy(request, monkeypatch):
    """a bf16 plan runs its recurrences on the bf16 MFMA kernels from B = 128 on and on the fp32 VALU kernels below
    (lstm_seq.hip::bf16_seq_pays); 'all-bf16' forces the bf16 kernels at every batch size."""
    # "dot2": below B = 128 the decoders' one-row forward recurrence takes its product on v_dot2c_f32_bf16 (plan option
    # "bf16_dot"; opt-in: measured no faster than the fp32 FMAs, profiles/r04_bf16_onerow.txt)
    if "dot2" in request.param:
        monkeypatch.setenv("MFM_BF16_DOT", "1")
    else:
        monkeypatch.delenv("MFM_BF16_DOT", raising=False)
    if request.param.startswith("all-bf16"):
        monkeypatch.setenv("MFM_BF16_SEQ_MINB", "1")
        monkeypatch.setenv("MFM_BF16_STORE", "1")       # and the bf16-RESIDENT saved activations (default from T*B = 3840)
    else:
        monkeypatch.delenv("MFM_BF16_SEQ_MINB", raising=False)
        monkeypatch.delenv("MFM_BF16_STORE", raising=False)
    # bf16-resident plans project on proj_bf16_kernel (proj_bf16.hip; it also writes the bf16 image of x the one-pass weight
    # gradients stream); "projNN" forces its panel height, "panel*" switches it off: gemm_panel_kernel<true> + x_to_bf16_kernel
    # one-pass weight gradients: 96-column M-tiles; from T*B = 65536 rows 128-column ones (dw_stream_kernel<false, 4, 8>) when
    # every item has <= 512 right-hand columns: "dwmf4" forces those
    if "dwmf4" in request.param:
        monkeypatch.setenv("MFM_DWB_MF", "4")
    else:
        monkeypatch.delenv("MFM_DWB_MF", raising=False)
    # "tnsplit": the B-row products of fp32 operands (the latent stack's weight gradients) on gemm_tn_kernel, the rest of
    # the last launch on the grouped bf16 GEMM (default from B = 192)
    if "tnsplit" in request.param:
        monkeypatch.setenv("MFM_GEMM_TN_BF16_MINB", "1")
    else:
        monkeypatch.delenv("MFM_GEMM_TN_BF16_MINB", raising=False)
    # decoder fc1 of a bf16-resident plan: dec_fc1_large64_kernel (64-row tiles); "fc1r16" = the 16-row kernel it replaced
    if "fc1r16" in request.param:
        monkeypatch.setenv("MFM_FC1_LARGE_ROWS", "16")
    else:
        monkeypatch.delenv("MFM_FC1_LARGE_ROWS", raising=False)
    monkeypatch.delenv("MFM_PROJ16", raising=False)
    monkeypatch.delenv("MFM_PROJ16_BM", raising=False)
    if "proj" in request.param:
        monkeypatch.setenv("MFM_PROJ16_BM", request.param.split("proj")[1])
    if "panel" in request.param:
        monkeypatch.setenv("MFM_PROJ16", "0")
        monkeypatch.setenv("MFM_PANEL_MINROWS", "1")    # gemm_panel_kernel<true> for the input projections
        bm = request.param.split("panel")[1]                # forced panel height (default: the launcher's pick, 128 here)
        if bm:
            monkeypatch.setenv("MFM_PANEL_BM", bm)
        else:
            monkeypatch.delenv("MFM_PANEL_BM", raising=False)
    else:
        monkeypatch.delenv("MFM_PANEL_BM", raising=False)
        monkeypatch.delenv("MFM_PANEL_MINROWS", raising=False)
    return request.param


# Per-tensor bounds of the bf16 gradients (round 6; the round-5 review: one `< 8e-2` for every tensor with 6.4e-2 measured).
#
# WHY some tensors are noisy.  The discriminative loss seeds d y_hat = +-1/B per row (L1), so the weight gradients of the y path
# (z_y -> f_y MLP, classifier) are sums over the batch rows of SIGNED terms that cancel: at B = 229 the net gradient of
# zy_to_fy_fc1 / fc2 is 1/17 / 1/47 of the sum of the rows' norms (~sqrt(B) .. 3 sqrt(B); oracle, per-row autograd).  Rounding the
# shared operands to bf16 moves every row's term coherently by ~2^-9, and the cancellation amplifies that relative to what is
# left.  It is a property of the PROBLEM at bf16 operand precision, not of the kernels' accumulation: the fp32 oracle evaluated
# with nothing but its weights, its batch, the hidden states between LSTM steps and the Linears' outputs rounded to bf16 -- fp32 arithmetic
# throughout -- is already off by ~6e-2 on zy_to_fy_fc1 at B = 229 (the kernels: 6.4e-2) and by ~1e-2 on everything else
# (_conditioning below computes exactly that).
#
# THE BOUND, per tensor n:   rel_L2(n) <= 2 * cond(n) + 1.5e-2,   cond(n) = that ablation's relative L2 error of tensor n.
# The factor 2 covers what the ablation leaves out (hidden states, gate gradients and d x_hat rounded at every step); the
# additive term is the bf16 level of a well-conditioned tensor (measured worst 2.2e-2 against 2 * 0.4e-2 + 1.5e-2).  Measured
# ratio rel / bound: worst 0.49 (profiles/*parity_worst.jsonl, bf16_grad_bound_ratio_*): a margin of ~2x everywhere, and a tensor
# that gets worse for a reason other than conditioning shows up against ITS bound instead of hiding under the noisiest one's.
NOISY_BF16_TENSORS = ("zy_to_fy_fc1.", "zy_to_fy_fc2.")          # (kept for the reports: the two the old common bound was set by)


def _conditioning(cfgs, w, x, y, cfg, loss_kind, variant="kl_ef"):
    """name -> relative L2 distance between the fp32 oracle's gradient and the SAME fp32 computation with weights and batch
    rounded to bf16 first: how far bf16 operands alone move each tensor."""
    def grads(rounded):
        m = O.build(variant, cfgs)
        ww = {k: (torch.from_numpy(np.asarray(v)).to(torch.bfloat16).float().numpy() if rounded else v) for k, v in w.items()}
        O.load_numpy_weights(m, ww)
        m.train()
        if rounded:
            # the hidden state of every LSTM step passes through bf16 as well (what a bf16 plan exchanges between steps and, when
            # bf16-resident, saves): the cast's backward rounds d h the same way
            for mod in m.modules():
                if isinstance(mod, torch.nn.LSTMCell):
                    mod.register_forward_hook(lambda _m, _i, out: (out[0].to(torch.bfloat16).float(), out[1]))
                elif isinstance(mod, torch.nn.Linear):          # ... and so does every Linear's output (the next product's operand)
                    mod.register_forward_hook(lambda _m, _i, out: out.to(torch.bfloat16).float())
        xx = x.to(torch.bfloat16).float() if rounded else x
        O.loss_terms(m, xx, y, cfg, loss_kind)["loss"].backward()
        return {n: p.grad.numpy().astype(np.float64).ravel() for n, p in m.named_parameters() if p.grad is not None}
    g0, g1 = grads(False), grads(True)
    return {n: float(np.linalg.norm(g1[n] - g0[n]) / max(np.linalg.norm(g0[n]), 1e-30)) for n in g0}, g0


def _check_bf16_gradients(gv, ref, cond, tag):
    """every tensor against ITS bound; returns (worst rel, worst rel of the well-conditioned rest, worst cosine, worst rel / bound)"""
    worst_g, worst_c, worst_rest, worst_ratio = ("", 0.0), ("", 1.0), ("", 0.0), ("", 0.0)
    for n, r in ref.items():
        g = gv[n].cpu().numpy().astype(np.float64).ravel()
        nr = np.linalg.norm(r)
        if nr < 1e-9:
            continue
        rel = float(np.linalg.norm(g - r) / nr)
        cos = float(g @ r / (np.linalg.norm(g) * nr + 1e-300))
        bound = 2.0 * cond[n] + 1.5e-2
        if rel > worst_g[1]:
            worst_g = (n, rel)
        if cos < worst_c[1]:
            worst_c = (n, cos)
        if not n.startswith(NOISY_BF16_TENSORS) and rel > worst_rest[1]:
            worst_rest = (n, rel)
        if rel / bound > worst_ratio[1]:
            worst_ratio = (n, rel / bound)
        assert rel <= bound, (tag, n, rel, bound, cond[n])
    cases.report("bf16_grad_relL2_%s" % tag, worst_g[1])
    cases.report("bf16_grad_relL2_rest_%s" % tag, worst_rest[1])
    cases.report("bf16_grad_one_minus_cos_%s" % tag, 1.0 - worst_c[1])
    cases.report("bf16_grad_bound_ratio_%s" % tag, worst_ratio[1])
    assert worst_g[1] < 0.2, worst_g              # (whatever the conditioning says: a gradient 20 % off is not a gradient)
    return worst_g, worst_rest, worst_c, worst_ratio


@pytest.mark.parametrize("name", ["klef_b32_t20", "klef_b33_t7", "klef_b1_t20", "klef_b5_t1", "klef_b229_t20",
                                  "klef_you_b32_t50", "klef_mosei_b64_t20", "klef_odd_b19_t9"])
def test_bf16_forward_and_gradients_near_fp32_reference(name, seq_policy):
    """bounds (measured worst case in brackets): loss terms within 1e-3 relative of the
    reference's fp32 golden [7e-5]; every parameter gradient within ITS bound 2 cond(n) + 1.5e-2 of the oracle's in relative L2
    norm (cond: what bf16 operands alone do to that tensor, see above) [worst rel / bound 0.49]; cosine > 0.995 [1 - 7.4e-4]."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cs = cases.load_case(name)
    e, w = _bf16_engine(cs["cfgs"])
    cfg, gold = cs["cfg"], cs["gold"]
    x, y = torch.from_numpy(cs["x"]), torch.from_numpy(cs["y"])
    xd, yd = x.cuda(), y.cuda()
    out = e.forward(xd, yd, train=True)
    ld = e.loss_dict(out["losses"])
    worst_l = 0.0
    for k in ("disc", "gen_l", "gen_a", "gen_v", "reg", "loss"):
        ref = float(gold["fwd_" + k])
        worst_l = max(worst_l, abs(ld[k] - ref) / max(abs(ref), 1e-3))
    cases.report("bf16_loss_terms_rel_%s_%s" % (name, seq_policy), worst_l)
    assert worst_l < 1e-3, (ld, worst_l)
    assert rel_err(out["y_hat"].cpu().numpy(), gold["y_hat"]) < 5e-2
    assert rel_err(out["x_a_hat"].cpu().numpy(), gold["x_a_hat"]) < 5e-2
    torch.set_num_threads(4)
    cond, ref = _conditioning(cs["cfgs"], w, x, y, cfg, cs["loss_kind"])
    e.backward(xd, yd, stage=0)
    worst_g, worst_rest, worst_c, _ = _check_bf16_gradients(e.grad_views(), ref, cond, "%s_%s" % (name, seq_policy))
    assert worst_c[1] > 0.995, worst_c
    # not accidentally the fp32 path
    assert worst_g[1] > 1e-5


@pytest.mark.parametrize("name", ["klef_b32_t20", "klef_you_b32_t50", "klef_b33_t7"])
def test_bf16_loss_curve_tracks_fp32_reference(name, seq_policy):
    """'matched loss curve' gate: N fused bf16 steps against the reference's own fp32
    loss trace (golden); every term within 2e-3 relative at every step [measured 1.5e-4 over 20 steps]."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cs = cases.load_case(name)
    e, _ = _bf16_engine(cs["cfgs"])
    x, y = torch.from_numpy(cs["x"]).cuda(), torch.from_numpy(cs["y"]).cuda()
    trace = []
    for _ in range(cs["steps"]):
        ld = e.loss_dict(e.train_step(x, y, lr=1e-3))
        trace.append([ld["loss"], ld["disc"], ld["gen"], ld["reg"]])
    trace, ref = np.array(trace), cs["gold"]["trace"]
    dev_ = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
    cases.report("bf16_trace_rel_%s_%s" % (name, seq_policy), dev_)
    assert dev_ < 2e-3, (trace[-1], ref[-1])
    assert trace[-1, 0] < trace[0, 0]                  # and it trains


def test_bf16_resident_plan_selection_and_stored_dtypes(monkeypatch):
    """which bf16 plans keep their saved activations as bf16 (default: from T*B = 2560 rows), and that the buffers really
    hold bf16: hs[T-1] is the bf16 rounding of the fp32 h_{T-1} copy the latent stack reads, the gates are activations"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    monkeypatch.delenv("MFM_BF16_STORE", raising=False)
    monkeypatch.delenv("MFM_BF16_SEQ_MINB", raising=False)
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    for prec, B, T, want in (("bf16", 32, 20, False), ("bf16", 112, 20, False), ("bf16", 128, 20, True), ("bf16", 192, 20, True), ("bf16", 1024, 20, True), ("fp32", 1024, 20, False)):
        e = engine.MFMEngine(cfgs, precision=prec)
        e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
        xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7)
        x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
        e.forward(x, y, train=True, want_xhat=False)
        sb = e.seq_buffers(T, B, 3)                      # the early-fusion encoder
        assert sb["bf16_resident"] == want, (prec, B)
        assert sb["bf16_recurrence"] == (prec == "bf16" and B >= 128)
        if want:
            assert sb["hs"].dtype == torch.bfloat16 and sb["gates"].dtype == torch.bfloat16 and sb["cs"].dtype == torch.float32
            h = sb["h"]
            hl = sb["h_last"].cpu().numpy()
            assert np.array_equal(bf(hl), b2n(sb["hs"])[T - 1])
            g = b2n(sb["gates"])[:, :, :, :h]
            assert np.all((g[:, :, [0, 1, 3]] >= 0.0) & (g[:, :, [0, 1, 3]] <= 1.0)) and np.all(np.abs(g[:, :, 2]) <= 1.0)
            dec = e.seq_buffers(T, B, 4)                  # decoder l
            assert dec["dxhat"].dtype == torch.bfloat16 and dec["dhs"].dtype == torch.bfloat16


@pytest.mark.parametrize("gname", ["klef_mosei_b1024_t20", "klef_mosei_b256_t50", "klef_you_b256_t50"])
def test_bf16_large_batch_mosei_loss_curve(gname):
    """BASELINE config 4's shape at a large batch (7 regression outputs; B=1024, T=20 and -- the sequence length the config
    names, SURVEY section 8d -- B=256, T=50): bf16 (bf16-resident plan) and fp32 loss curves against the reference's fp32 trace
    (goldens klef_mosei_b1024_t20 / klef_mosei_b256_t50, light: summaries only)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    gold = np.load(cases.GOLDEN + "/%s.npz" % gname)
    B, T, steps = (int(v) for v in gold["meta"])
    cfgs = (configs.you_configs if "_you_" in gname else configs.mosei_configs)(dropout=False)      # (round 6: the YouTube shape too)
    cfg = cfgs[0]
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7, output_dim=cfg["output_dim"],
                              classes=cfg["output_dim"] if cfg.get("loss", "l1") == "ce" else 0)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    for prec, bound in (("fp32", 2e-4), ("bf16", 1e-3)):          # measured: bf16 3.2e-5
        from factorized_amd import engine
        e = engine.MFMEngine(cfgs, precision=prec)
        e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
        trace = []
        for _ in range(steps):
            ld = e.loss_dict(e.train_step(x, y, lr=1e-3))
            trace.append([ld["loss"], ld["disc"], ld["gen"], ld["reg"]])
        trace, ref = np.array(trace), gold["trace"]
        dev_ = float(np.max(np.abs(trace - ref) / np.maximum(np.abs(ref), 1e-2)))
        cases.report("%s_trace_rel_%s" % (gname, prec), dev_)
        assert dev_ < bound, (prec, trace[-1], ref[-1])
        if prec == "bf16":
            assert e.seq_buffers(T, B, 0)["bf16_resident"]


@pytest.mark.parametrize("shape", ["mosei", "you"])
def test_bf16_resident_t50_gradients_against_the_oracle(shape):
    """T=50 x large B on the bf16-RESIDENT path (B=256: 12,800 rows), MOSEI shape (7 regression outputs) and -- round 6, BASELINE
    config 3 -- YouTube shape (cross-entropy head, D = 410): losses against the reference's golden (klef_mosei_b256_t50 /
    klef_you_b256_t50) and the oracle, all 78 gradients of one step against the fp32 CPU oracle on the same batch, every tensor
    against its own bound (see _conditioning)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    B, T = 256, 50
    cfgs = (configs.mosei_configs if shape == "mosei" else configs.you_configs)(dropout=False)
    cfg = cfgs[0]
    loss_kind = cfg.get("loss", "l1")
    gold = np.load(cases.GOLDEN + "/klef_%s_b256_t50.npz" % shape)
    xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=7, output_dim=cfg["output_dim"],
                              classes=cfg["output_dim"] if loss_kind == "ce" else 0)
    e = engine.MFMEngine(cfgs, precision="bf16")
    w = synth.make_weights(e.layout.shapes, seed=1234)
    e.load_weights(w)
    torch.set_num_threads(8)
    cond, ref = _conditioning(cfgs, w, torch.from_numpy(xn), torch.from_numpy(yn), cfg, loss_kind)
    x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
    out = e.forward(x, y, train=True, want_xhat=False)
    ld = e.loss_dict(out["losses"])
    assert e.seq_buffers(T, B, 0)["bf16_resident"]
    for k in ("disc", "gen", "reg", "loss"):
        r = float(gold["fwd_" + k])                     # the REFERENCE's own fp32 value of the same step
        assert abs(ld[k] - r) <= 2e-3 * max(abs(r), 1e-2), (k, ld[k], r)
    e.backward(x, y, stage=0)
    worst_g, _, _, _ = _check_bf16_gradients(e.grad_views(), ref, cond, "resident_%s_b256_t50" % shape)
    cases.report("bf16_resident_%s_b256_t50_grad_relL2" % shape, worst_g[1])
    assert worst_g[1] > 1e-5          # not accidentally the fp32 path
