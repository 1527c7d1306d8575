"""dw_bf16_kernel alone, one LSTM shape per launch (mfm_dw_bf16_lstm): us per launch and cycles per (M-tile, 32-row chunk)
on one CU when the item fills the chip -- calibrates the launcher's cost model (1 + N / 128 per chunk)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from factorized_amd import _lib
L = _lib.lib()
rows, shift = 40960, 2048
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
for name, h, dx, ldx in (("ef", 120, 325, 352), ("enc_l", 32, 300, 304), ("enc_v", 80, 20, 32), ("enc_a", 8, 5, 16),
                         ("dec_l", 104, 0, 0), ("dec_a", 24, 0, 0)):
    Hp = (h + 15) // 16 * 16
    dA = torch.randn(rows, 4 * Hp, device="cuda").bfloat16()
    hs = torch.randn(rows, Hp, device="cuda").bfloat16()
    xb = torch.randn(rows, ldx, device="cuda").bfloat16() if dx else None
    dw_ih = torch.zeros(4 * h, max(dx, 1), device="cuda"); dw_hh = torch.zeros(4 * h, h, device="cuda")
    db1 = torch.zeros(4 * h, device="cuda"); db2 = torch.zeros(4 * h, device="cuda")
    def run():
        _lib.check(L.mfm_dw_bf16_lstm(p(dA), rows, h, p(xb), ldx, dx, p(hs), shift, p(dw_ih) if dx else C.c_void_p(0), p(dw_hh),
                                      C.c_void_p(0), p(db1), p(db2), None), "dw")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    mt = (4 * Hp + 95) // 96
    N = (ldx if dx else 0) + Hp
    byt = rows * (4 * Hp + N) * 2
    print("%-6s h=%3d N=%3d m_tiles=%d  %7.1f us  %6.0f cycles/(tile,chunk)/CU  operands %5.1f MB -> %5.0f GB/s (once-through)" %
          (name, h, N, mt, us, us * 1e-6 * 2.4e9 * 256 / (mt * rows / 32), byt / 1e6, byt / us / 1e3))
