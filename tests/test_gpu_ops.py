"""GPU parity of the individual C-ABI ops against CPU references (numpy fp64 / the torch-CPU
oracle recurrences).  Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest
import torch

from tests.cases import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star: 1e-4 relative fp32 tolerance


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from factorized_amd import engine
    return engine


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


# ---------------------------------------------------------------------------------- GEMM
@pytest.fixture(params=["auto", "1", "2", "4"])
def gemm_fr(request, monkeypatch):
    """tile size of the grouped GEMM: 32x32, 64x64, 128x128 (gemm_f32_kernel<1|2|4, *>) or the launcher's own choice"""
    if request.param == "auto":
        monkeypatch.delenv("MFM_GEMM_FR", raising=False)
    else:
        monkeypatch.setenv("MFM_GEMM_FR", request.param)
    return request.param


@pytest.mark.parametrize("m,n,k", [(640, 128, 300), (37, 5, 11), (64, 64, 16), (1, 1, 1), (130, 70, 325)])
def test_gemm_nt_bias(eng, m, n, k, gemm_fr):
    rs = np.random.RandomState(m + n + k)
    A = rs.normal(size=(m, k + 3)).astype(np.float32)     # row stride k+3: a strided slice
    W = rs.normal(size=(n, k)).astype(np.float32)
    b1 = rs.normal(size=n).astype(np.float32)
    b2 = rs.normal(size=n).astype(np.float32)
    a_d, w_d, b1_d, b2_d = dev(A), dev(W), dev(b1), dev(b2)
    npad = n + 3
    c_d = torch.full((m, npad), 7.0, device="cuda")
    d = eng.make_gemm(a_d, w_d, c_d, m, npad, k, a_sm=k + 3, a_sk=1, b_sk=1, b_sn=k, ldc=npad,
                      bias=b1_d, bias2=b2_d, n_valid=n)
    eng.gemm_grouped([d])
    ref = A[:, :k].astype(np.float64) @ W.T.astype(np.float64) + b1 + b2
    out = c_d.cpu().numpy()
    assert rel_err(out[:, :n], ref) < 2e-6
    assert np.all(out[:, n:] == 0.0)            # pad columns are exact zeros


def test_gemm_tn_splitk_accumulate_dual_output(eng, gemm_fr):
    rs = np.random.RandomState(5)
    R, M, N = 640, 120, 325
    dA = rs.normal(size=(R, 4, 128)).astype(np.float32)      # [rows, gate, Hp]
    X = rs.normal(size=(R, N)).astype(np.float32)
    da_d, x_d = dev(dA), dev(X)
    c1 = torch.zeros(4, M, N, device="cuda")
    c2 = torch.zeros(4, M, N, device="cuda")
    d = eng.make_gemm(da_d, x_d, c1, M, N, R, a_sm=1, a_sk=4 * 128, b_sk=N, b_sn=1, ldc=N, batch=4,
                      a_sz=128, c_sz=M * N, accumulate=1, split_k=0, c2=c2)
    eng.gemm_grouped([d])
    ref = np.einsum("rgm,rn->gmn", dA[:, :, :M].astype(np.float64), X.astype(np.float64))
    assert rel_err(c1.cpu().numpy(), ref) < 5e-6
    assert rel_err(c2.cpu().numpy(), ref) < 5e-6


def test_gemm_nn_and_group_of_many(eng, gemm_fr):
    rs = np.random.RandomState(9)
    descs, refs, outs = [], [], []
    for i in range(7):
        m, n, k = 50 + 13 * i, 20 + 9 * i, 5 + 31 * i
        A = rs.normal(size=(m, k)).astype(np.float32)
        B = rs.normal(size=(k, n)).astype(np.float32)
        a_d, b_d = dev(A), dev(B)
        c_d = torch.empty(m, n, device="cuda")
        descs.append(eng.make_gemm(a_d, b_d, c_d, m, n, k, a_sm=k, a_sk=1, b_sk=n, b_sn=1, ldc=n, alpha=0.5))
        refs.append(0.5 * A.astype(np.float64) @ B.astype(np.float64))
        outs.append((a_d, b_d, c_d))
    eng.gemm_grouped(descs)
    for (_, _, c_d), ref in zip(outs, refs):
        assert rel_err(c_d.cpu().numpy(), ref) < 2e-6


@pytest.mark.parametrize("count", [56, 57, 130])
def test_gemm_group_larger_than_one_launch(eng, count):
    """A group at and beyond the per-launch problem limit (56): the ABI call splits it into launches; every
    problem, including the last one of a full launch and the first of the next, must come out right."""
    rs = np.random.RandomState(count)
    descs, refs, outs = [], [], []
    for i in range(count):
        m, n, k = 1 + (i * 7) % 70, 1 + (i * 11) % 45, 1 + (i * 5) % 90
        A = rs.normal(size=(m, k)).astype(np.float32)
        B = rs.normal(size=(k, n)).astype(np.float32)
        a_d, b_d = dev(A), dev(B)
        c_d = torch.full((m, n), 7.0, device="cuda")
        descs.append(eng.make_gemm(a_d, b_d, c_d, m, n, k, a_sm=k, a_sk=1, b_sk=n, b_sn=1, ldc=n))
        refs.append(A.astype(np.float64) @ B.astype(np.float64))
        outs.append((a_d, b_d, c_d))
    eng.gemm_grouped(descs)
    for i, ((_, _, c_d), ref) in enumerate(zip(outs, refs)):
        assert rel_err(c_d.cpu().numpy(), ref) < 2e-6, i


def test_gemm_column_sums_with_ones(eng):
    rs = np.random.RandomState(3)
    R, M = 333, 96
    dA = rs.normal(size=(R, M)).astype(np.float32)
    da_d = dev(dA)
    ones = torch.ones(R, device="cuda")
    c = torch.zeros(M, device="cuda")
    d = eng.make_gemm(da_d, ones, c, M, 1, R, a_sm=1, a_sk=M, b_sk=1, b_sn=1, ldc=1, accumulate=1, split_k=0)
    eng.gemm_grouped([d])
    assert rel_err(c.cpu().numpy(), dA.astype(np.float64).sum(0)) < 5e-6


# ---------------------------------------------------------------------------------- LSTM sequences
def _cpu_lstm(x, w_ih, w_hh, b_ih, b_hh, dec_init=None, T=None):
    """torch-CPU recurrence with autograd (reference semantics mfm_model.py:47-58 / 72-88)."""
    h = w_hh.shape[1]
    cell = torch.nn.LSTMCell(w_ih.shape[1], h)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(w_ih)); cell.weight_hh.copy_(torch.from_numpy(w_hh))
        cell.bias_ih.copy_(torch.from_numpy(b_ih)); cell.bias_hh.copy_(torch.from_numpy(b_hh))
    hs, cs = [], []
    if dec_init is None:
        xt = torch.from_numpy(x)
        B = xt.shape[1]
        hx, cx = torch.zeros(B, h), torch.zeros(B, h)
        for t in range(xt.shape[0]):
            hx, cx = cell(xt[t], (hx, cx))
            hs.append(hx); cs.append(cx)
        return cell, None, torch.stack(hs), torch.stack(cs)
    init = torch.from_numpy(dec_init).clone().requires_grad_(True)
    B = init.shape[0]
    hx, cx = torch.zeros(B, h), torch.zeros(B, h)
    inp = init
    for t in range(T):
        hx, cx = cell(inp, (hx, cx))
        inp = hx
        hs.append(hx); cs.append(cx)
    return cell, init, torch.stack(hs), torch.stack(cs)


def _pad_gates(g, h, Hp):
    """[T,B,4h] -> [T,B,4,Hp] zero padded."""
    T, B = g.shape[:2]
    out = np.zeros((T, B, 4, Hp), dtype=np.float32)
    out[:, :, :, :h] = g.reshape(T, B, 4, h)
    return out


SEQ_SHAPES = [(8, 5, 1, 3), (8, 5, 32, 20), (24, 7, 33, 4), (32, 300, 32, 20), (80, 20, 17, 6),
              (104, 9, 16, 5), (120, 325, 32, 20), (20, 6, 5, 1), (128, 4, 3, 2), (36, 10, 40, 3)]


@pytest.fixture(params=["mfma", "small", "small:1", "small:2", "small:4", "stepwise", "small:1:ks8"])
def seq_path(request, monkeypatch):
    """Both recurrent kernel families: MFMA (16 rows/workgroup) and VALU small-tile (4 rows); "ks8": the one-row BPTT
    with 8 k-slices per unit pair (small_bwd_body<.., 1, 8>, opt-in)."""
    monkeypatch.delenv("MFM_SEQ_KS", raising=False)
    if request.param.endswith(":ks8"):
        monkeypatch.setenv("MFM_SEQ_KS", "8")
    path, _, rows = request.param.replace(":ks8", "").partition(":")
    if path == "stepwise":                            # recurrent GEMM + cell kernel per step (h > 128 path)
        monkeypatch.setenv("MFM_SEQ_STEPWISE", "1")
        path = "small"
    else:
        monkeypatch.delenv("MFM_SEQ_STEPWISE", raising=False)
    monkeypatch.setenv("MFM_SEQ_PATH", path)
    if rows:
        monkeypatch.setenv("MFM_SEQ_ROWS", rows)      # force the row-tile size of the VALU kernels
    else:
        monkeypatch.delenv("MFM_SEQ_ROWS", raising=False)
    return request.param


@pytest.mark.parametrize("h,d,B,T", SEQ_SHAPES)
def test_lstm_seq_encoder_fwd_bwd(eng, seq_path, h, d, B, T):
    rs = np.random.RandomState(h * 7 + B)
    k = 1.0 / np.sqrt(h)
    w_ih = rs.uniform(-k, k, size=(4 * h, d)).astype(np.float32)
    w_hh = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    b_ih = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    b_hh = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    x = rs.normal(size=(T, B, d)).astype(np.float32)
    cell, _, hs_ref, cs_ref = _cpu_lstm(x, w_ih, w_hh, b_ih, b_hh)
    Hp = (h + 15) // 16 * 16
    gx = x.astype(np.float64) @ w_ih.T.astype(np.float64) + b_ih + b_hh
    gates = dev(_pad_gates(gx.astype(np.float32), h, Hp))
    hs = torch.full((T, B, Hp), 9.0, device="cuda")
    cs = torch.full((T, B, Hp), 9.0, device="cuda")
    w_d = dev(w_hh)
    desc = eng.make_seq(gates, hs, cs, w_d, h)
    eng.lstm_seq([desc], T, B)
    hs_o, cs_o = hs.cpu().numpy(), cs.cpu().numpy()
    assert rel_err(hs_o[:, :, :h], hs_ref.detach().numpy()) < TOL
    assert rel_err(cs_o[:, :, :h], cs_ref.detach().numpy()) < TOL
    assert np.all(hs_o[:, :, h:] == 0.0) and np.all(cs_o[:, :, h:] == 0.0)

    # ---- backward: external grad on the last hidden state only (encoder form)
    dh_last = rs.normal(size=(B, h)).astype(np.float32)
    (hs_ref[-1] * torch.from_numpy(dh_last)).sum().backward()
    dh_d = dev(dh_last)
    desc = eng.make_seq(gates, hs, cs, w_d, h, dh_ext=dh_d, ld_dh=h)
    eng.lstm_seq([desc], T, B, backward=True)
    dA = gates.cpu().numpy().astype(np.float64)[:, :, :, :h].reshape(T, B, 4 * h)
    db = dA.sum((0, 1))
    dW_ih = np.einsum("tbg,tbd->gd", dA, x.astype(np.float64))
    hprev = np.concatenate([np.zeros((1, B, h)), hs_ref.detach().numpy()[:-1]], 0)
    dW_hh = np.einsum("tbg,tbh->gh", dA, hprev)
    assert rel_err(db, cell.bias_ih.grad.numpy()) < TOL
    assert rel_err(dW_ih, cell.weight_ih.grad.numpy()) < TOL
    assert rel_err(dW_hh, cell.weight_hh.grad.numpy()) < TOL


@pytest.mark.parametrize("h,B,T", [(24, 32, 20), (104, 32, 20), (24, 5, 1), (40, 19, 3), (112, 33, 7)])
def test_lstm_seq_decoder_fwd_bwd(eng, seq_path, h, B, T):
    rs = np.random.RandomState(h + B + T)
    k = 1.0 / np.sqrt(h)
    w_ih = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    w_hh = rs.uniform(-k, k, size=(4 * h, h)).astype(np.float32)
    b_ih = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    b_hh = rs.uniform(-k, k, size=4 * h).astype(np.float32)
    init = rs.normal(size=(B, h)).astype(np.float32)
    cell, init_t, hs_ref, cs_ref = _cpu_lstm(None, w_ih, w_hh, b_ih, b_hh, dec_init=init, T=T)
    Hp = (h + 15) // 16 * 16
    gates = torch.full((T, B, 4, Hp), 3.0, device="cuda")
    hs = torch.full((T, B, Hp), 9.0, device="cuda")
    cs = torch.full((T, B, Hp), 9.0, device="cuda")
    wi, wh, bi, bh, init_d = dev(w_ih), dev(w_hh), dev(b_ih), dev(b_hh), dev(init)
    desc = eng.make_seq(gates, hs, cs, wh, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init_d, is_dec=True)
    eng.lstm_seq([desc], T, B)
    hs_o = hs.cpu().numpy()
    assert rel_err(hs_o[:, :, :h], hs_ref.detach().numpy()) < TOL
    assert rel_err(cs.cpu().numpy()[:, :, :h], cs_ref.detach().numpy()) < TOL
    assert np.all(hs_o[:, :, h:] == 0.0)

    dH = rs.normal(size=(T, B, h)).astype(np.float32)
    (hs_ref * torch.from_numpy(dH)).sum().backward()
    dH_p = np.zeros((T, B, Hp), dtype=np.float32)
    dH_p[:, :, :h] = dH
    dh_d = dev(dH_p)
    dinit = torch.full((B, h), 5.0, device="cuda")
    desc = eng.make_seq(gates, hs, cs, wh, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init_d, is_dec=True,
                        dh_ext=dh_d, ld_dh=Hp, d_h_init=dinit)
    eng.lstm_seq([desc], T, B, backward=True)
    dA = gates.cpu().numpy().astype(np.float64)[:, :, :, :h].reshape(T, B, 4 * h)
    hsr = hs_ref.detach().numpy().astype(np.float64)
    S = np.einsum("tbg,tbh->gh", dA[1:], hsr[:-1]) if T > 1 else np.zeros((4 * h, h))
    dW_hh = S
    dW_ih = S + np.einsum("bg,bh->gh", dA[0], init.astype(np.float64))
    assert rel_err(dinit.cpu().numpy(), init_t.grad.numpy()) < TOL
    assert rel_err(dA.sum((0, 1)), cell.bias_hh.grad.numpy()) < TOL
    assert rel_err(dW_ih, cell.weight_ih.grad.numpy()) < TOL
    if T > 1:
        assert rel_err(dW_hh, cell.weight_hh.grad.numpy()) < TOL


def test_lstm_seq_four_in_one_launch(eng, seq_path):
    """The 4 encoders of the canonical model share one launch; results must equal solo launches (to rounding:
    a solo launch of an uncommon size takes the generic 4-row tile whose k-order of partial sums differs
    from the 1-row tile of the shared launch)."""
    rs = np.random.RandomState(0)
    T, B = 20, 32
    keep, solo, descs = [], [], []
    for h in (32, 8, 80, 120):
        Hp = (h + 15) // 16 * 16
        k = 1.0 / np.sqrt(h)
        w = dev(rs.uniform(-k, k, size=(4 * h, h)))
        g0 = _pad_gates(rs.normal(size=(T, B, 4 * h)).astype(np.float32), h, Hp)
        bufs = []
        for _ in range(2):
            gates = dev(g0)
            hs = torch.zeros(T, B, Hp, device="cuda")
            cs = torch.zeros(T, B, Hp, device="cuda")
            bufs.append((gates, hs, cs))
        keep.append((w, bufs))
        descs.append(eng.make_seq(*bufs[0], w, h))
        solo.append(eng.make_seq(*bufs[1], w, h))
    eng.lstm_seq(descs, T, B)
    for s in solo:
        eng.lstm_seq([s], T, B)
    torch.cuda.synchronize()
    for w, bufs in keep:
        for a, b in zip(bufs[0], bufs[1]):
            assert torch.allclose(a, b, rtol=0.0, atol=2e-6)


# ---------------------------------------------------------------------------------- MSE / Adam
def test_mse_fwd_bwd(eng):
    import ctypes as C
    from factorized_amd import _lib
    rs = np.random.RandomState(1)
    rows, d, D = 640, 20, 325
    X = rs.normal(size=(rows, D)).astype(np.float32)
    xh = rs.normal(size=(rows, d)).astype(np.float32)
    x_d, xh_d = dev(X), dev(xh)
    dx = torch.empty(rows, d, device="cuda")
    slot = torch.zeros(1, device="cuda")
    off = 305
    xs = x_d.view(-1)[off:]
    _lib.check(_lib.lib().mfm_mse_fwd_bwd(xh_d.data_ptr(), xs.data_ptr(), D, rows, d, 1.0 / (rows * d),
                                          2.0 * 0.5 / (rows * d), dx.data_ptr(), slot.data_ptr(),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    diff = xh.astype(np.float64) - X[:, off:off + d]
    assert abs(slot.item() - (diff ** 2).mean()) < 1e-5 * (diff ** 2).mean()
    assert rel_err(dx.cpu().numpy(), diff / (rows * d)) < 1e-6


def test_adam_matches_torch(eng):
    import ctypes as C
    from factorized_amd import _lib
    rs = np.random.RandomState(2)
    n = 4099
    p0 = rs.normal(size=n).astype(np.float32)
    p_ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([p_ref])
    p = dev(np.concatenate([p0, np.zeros(1, np.float32)]))[:n]
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 6):
        g = rs.normal(size=n).astype(np.float32) * (10.0 ** rs.randint(-4, 2))
        p_ref.grad = torch.from_numpy(g.copy())
        opt.step()
        g_d = dev(g)
        _lib.check(_lib.lib().mfm_adam_flat(p.data_ptr(), g_d.data_ptr(), m.data_ptr(), v.data_ptr(), n, step,
                                            1e-3, 0.9, 0.999, 1e-8, 1.0,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert rel_err(p.cpu().numpy(), p_ref.detach().numpy()) < 1e-6


# ---------------------------------------------------------------------------------- MFN memory recurrence
@pytest.mark.parametrize("T,B,M,H1,H2", [(20, 32, 64, 128, 128), (7, 5, 24, 40, 72), (1, 3, 64, 128, 128), (9, 19, 100, 100, 60)])
def test_mfn_memory_recurrence(eng, T, B, M, H1, H2):
    """mfm_mfn_mem_fwd/bwd against the step-by-step statement of reference mfm_model.py:177-181 in torch
    (float64 on the CPU): last memory and every input / weight gradient."""
    from factorized_amd.mfm_model import _MemFn
    assert _MemFn.supported(M, H1, H2)
    rs = np.random.RandomState(5)
    f = lambda *shape, s=1.0: (rs.normal(size=shape) * s).astype(np.float32)
    arrs = dict(g1=f(T, B, H1), g2=f(T, B, H2), ch=np.tanh(f(T, B, M)),
                w1m=f(H1, M, s=0.2), w2m=f(H2, M, s=0.2), w1b=f(M, H1, s=0.15), b1b=f(M, s=0.1),
                w2b=f(M, H2, s=0.15), b2b=f(M, s=0.1))
    dmem = f(B, M)
    # reference
    r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in arrs.items()}
    mem = torch.zeros(B, M, dtype=torch.float64)
    for t in range(T):
        a1 = torch.relu(r["g1"][t] + mem @ r["w1m"].T)
        a2 = torch.relu(r["g2"][t] + mem @ r["w2m"].T)
        gamma1 = torch.sigmoid(a1 @ r["w1b"].T + r["b1b"])
        gamma2 = torch.sigmoid(a2 @ r["w2b"].T + r["b2b"])
        mem = gamma1 * mem + gamma2 * r["ch"][t]
    (mem * torch.tensor(dmem, dtype=torch.float64)).sum().backward()
    # HIP
    d = {k: torch.tensor(v, device="cuda", requires_grad=True) for k, v in arrs.items()}
    out = _MemFn.apply(d["g1"], d["g2"], d["ch"], d["w1m"], d["w2m"], d["w1b"], d["b1b"], d["w2b"], d["b2b"], 0.0, 0.0, True)
    (out * torch.tensor(dmem, device="cuda")).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu().numpy(), mem.detach().numpy()) < TOL
    for k in arrs:
        assert rel_err(d[k].grad.cpu().numpy(), r[k].grad.numpy()) < TOL, k
    # inputs are not modified (the kernel works on private copies)
    assert np.array_equal(d["g1"].detach().cpu().numpy(), arrs["g1"])


def test_mfn_memory_dropout_statistics(eng):
    """Train-mode dropout inside the gamma nets: kept fraction ~ 1-p, scaled by 1/(1-p), and the backward
    uses the same mask (a zeroed activation passes no gradient)."""
    from factorized_amd.mfm_model import _MemFn
    T, B, M, H = 6, 64, 64, 128
    g = torch.full((T, B, H), 1.0, device="cuda", requires_grad=True)      # relu input > 0 everywhere
    z = lambda *s: torch.zeros(*s, device="cuda")
    ch = torch.ones(T, B, M, device="cuda")
    out = _MemFn.apply(g, g.detach().clone().requires_grad_(True), ch, z(H, M), z(H, M), torch.full((M, H), 0.01, device="cuda"),
                       z(M), z(M, H), z(M), 0.5, 0.0, True)
    out.sum().backward()
    torch.cuda.synchronize()
    kept = (g.grad[1:] != 0).float().mean().item()       # t = 0 passes no gradient to gamma1 (mem_{-1} = 0)
    assert 0.45 < kept < 0.55


# ---------------------------------------------------------------------------------- MMD
def test_mmd_wide_features_device_ops(eng):
    """dim > 256 is outside mmd_kernel's register budget: loss_MMD then composes the reference's formula from device ops
    (never the CPU); value and gradient against float64.  A CPU tensor raises like every other op of the package."""
    from factorized_amd.mfm_model import loss_MMD
    from factorized_amd import _lib
    rs = np.random.RandomState(77)
    B, dim = 24, 300
    zn = rs.normal(size=(B, dim)).astype(np.float32) * 1.1
    gn = rs.normal(size=(B, dim)).astype(np.float32)

    def ck(x, y):
        return torch.exp(-((x.unsqueeze(1) - y.unsqueeze(0)) ** 2).mean(2) / float(dim))
    zr = torch.tensor(zn, dtype=torch.float64, requires_grad=True)
    gr = torch.tensor(gn, dtype=torch.float64)
    ref = ck(gr, gr).mean() + ck(zr, zr).mean() - 2.0 * ck(gr, zr).mean()
    ref.backward()
    zd = torch.tensor(zn, device="cuda", requires_grad=True)
    out = loss_MMD(zd, torch.tensor(gn, device="cuda"))
    assert out.is_cuda
    out.backward()
    assert abs(out.item() - ref.item()) <= TOL * max(abs(ref.item()), 1e-3)
    assert rel_err(zd.grad.cpu().numpy(), zr.grad.numpy()) < 10 * TOL
    with pytest.raises(_lib.MfmError):
        loss_MMD(torch.tensor(zn), torch.tensor(gn))


@pytest.mark.parametrize("rows", ["auto", "8", "32"])
@pytest.mark.parametrize("B,dim", [(32, 32), (19, 80), (100, 8), (64, 256), (1, 16), (33, 5), (300, 24)])
def test_mmd_matches_reference_formula(eng, B, dim, rows, monkeypatch):
    """mfm_mmd_fwd_bwd against the reference statement of loss_MMD / compute_kernel (mfm_model.py:14-34) in
    float64 on the CPU: value and gradient wrt z.  rows: mmd_kernel<8> (default for B <= 256) / mmd_kernel<32> forced."""
    from factorized_amd.mfm_model import loss_MMD
    if rows == "auto":
        monkeypatch.delenv("MFM_MMD_ROWS", raising=False)
    else:
        monkeypatch.setenv("MFM_MMD_ROWS", rows)
    rs = np.random.RandomState(B * 1000 + dim)
    zn = rs.normal(size=(B, dim)).astype(np.float32) * 1.3
    gn = rs.normal(size=(B, dim)).astype(np.float32)

    def ck(x, y):
        d = x.shape[1]
        return torch.exp(-((x.unsqueeze(1) - y.unsqueeze(0)) ** 2).mean(2) / float(d))
    zr = torch.tensor(zn, dtype=torch.float64, requires_grad=True)
    gr = torch.tensor(gn, dtype=torch.float64)
    ref = ck(gr, gr).mean() + ck(zr, zr).mean() - 2.0 * ck(gr, zr).mean()
    (3.0 * ref).backward()
    zd = torch.tensor(zn, device="cuda", requires_grad=True)
    out = loss_MMD(zd, torch.tensor(gn, device="cuda"))
    (3.0 * out).backward()
    assert abs(out.item() - ref.item()) <= TOL * max(abs(ref.item()), 1e-3)
    assert rel_err(zd.grad.cpu().numpy(), zr.grad.numpy()) < TOL
