#!/usr/bin/env python3
"""The four bf16 recurrence launches of one MFM_KL_EF step (csrc/lstm_seq_bf16.hip) on their own, at the MOSI shapes:
encoders l/a/v/ef (h = 32, 8, 80, 120) forward and BPTT, decoders l/a/v (h = 104, 24, 24) forward and BPTT, bf16-resident
saved activations, packed weights.  Prints us per launch, us per time step and the HBM rate of the algorithmic bytes.

    python scripts/bench_seq_bf16.py [B=2048] [T=20]          (MFM_LIB_PATH=... selects an experimental build)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from factorized_amd import _lib
from factorized_amd import engine as E

ENC = [32, 8, 80, 120]
DEC = [104, 24, 24]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    dev = "cuda"
    torch.manual_seed(0)
    keep = []

    def mk(h, dec):
        Hp = (h + 15) // 16 * 16
        k = 1.0 / np.sqrt(h)
        gates = (torch.randn(T, B, 4, Hp, device=dev) * 0.5).to(torch.bfloat16)
        hs = torch.zeros(T, B, Hp, device=dev, dtype=torch.bfloat16)
        cs = torch.zeros(T, B, Hp, device=dev)
        w = (torch.rand(4 * h, h, device=dev) * 2 - 1) * k
        pack = torch.zeros(_lib.lib().mfm_lstm_pack_bytes(h, int(dec)), dtype=torch.uint8, device=dev)
        hl = torch.zeros(B, Hp, device=dev)
        if dec:
            wi = (torch.rand(4 * h, h, device=dev) * 2 - 1) * k
            bi = torch.zeros(4 * h, device=dev)
            bh = torch.zeros(4 * h, device=dev)
            init = torch.randn(B, h, device=dev)
            dh = (torch.randn(T, B, Hp, device=dev) * 0.1).to(torch.bfloat16)
            dinit = torch.zeros(B, h, device=dev)
            keep.append((gates, hs, cs, w, pack, wi, bi, bh, init, dh, dinit))
            f = E.make_seq(gates, hs, cs, w, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init, is_dec=True, w_pack=pack, store_bf16=True)
            b = E.make_seq(gates, hs, cs, w, h, w_ih=wi, b_ih=bi, b_hh=bh, h_init=init, is_dec=True, w_pack=pack, store_bf16=True,
                           dh_ext=dh, ld_dh=Hp, d_h_init=dinit)
        else:
            dh = torch.randn(B, h, device=dev) * 0.1
            keep.append((gates, hs, cs, w, pack, dh, hl))
            f = E.make_seq(gates, hs, cs, w, h, w_pack=pack, store_bf16=True, h_last=hl)
            b = E.make_seq(gates, hs, cs, w, h, w_pack=pack, store_bf16=True, dh_ext=dh, ld_dh=h)
        return f, b, Hp

    enc = [mk(h, False) for h in ENC]
    dec = [mk(h, True) for h in DEC]
    L = _lib.lib()

    def arr(ds):
        return (_lib.SeqDesc * len(ds))(*ds)

    all_f = arr([x[0] for x in enc + dec])
    _lib.check(L.mfm_lstm_pack_bf16(all_f, len(enc + dec), None), "pack")
    cases = [("enc_seq_fwd", arr([x[0] for x in enc]), len(enc), L.mfm_lstm_seq_fwd_bf16, sum(x[2] for x in enc) * (14 + 8)),
             ("dec_seq_fwd", arr([x[0] for x in dec]), len(dec), L.mfm_lstm_seq_fwd_bf16, sum(x[2] for x in dec) * 14),
             ("dec_seq_bwd", arr([x[1] for x in dec]), len(dec), L.mfm_lstm_seq_bwd_bf16, sum(x[2] for x in dec) * (8 + 8 + 2 + 8)),
             ("enc_seq_bwd", arr([x[1] for x in enc]), len(enc), L.mfm_lstm_seq_bwd_bf16, sum(x[2] for x in enc) * (8 + 8 + 8))]
    tot = 0.0
    for name, a, n, fn, bytes_per_row_step in cases:
        if only and only not in name:
            continue
        for _ in range(3):
            _lib.check(fn(a, n, T, B, None), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fn(a, n, T, B, None)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        tot += us
        mb = bytes_per_row_step * B * T / 1e6
        print("%-12s %8.2f us/launch  %6.3f us/step  %7.1f MB algorithmic  %6.0f GB/s" % (name, us, us / T, mb, mb / us * 1e3))
    print("total %.2f us" % tot)


if __name__ == "__main__":
    main()
