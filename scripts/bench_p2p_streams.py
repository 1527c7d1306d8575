"""The P2P all-reduce kernel with all W ranks in ONE process: W handles connected by base pointers, one stream
per rank on cuda:0.  Kernels of one process do run concurrently, so this times the kernel's own cost (two flag
rounds + three passes over the 1.9 MB buffer) without xGMI and without cross-process scheduling."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from factorized_amd import _lib  # noqa: E402


def run(W, n=477294, iters=200):
    L = _lib.lib()
    hs = []
    for r in range(W):
        h = C.c_void_p()
        _lib.check(L.mfm_p2p_create(W, r, n, C.byref(h)), "create")
        hs.append(h)
    bases = (C.c_void_p * W)(*[L.mfm_p2p_local_base(h) for h in hs])
    for h in hs:
        _lib.check(L.mfm_p2p_connect_bases(h, bases), "connect")
    streams = [torch.cuda.Stream() for _ in range(W)]
    g = torch.Generator().manual_seed(3)
    src = [torch.randn(n, generator=g).cuda() for _ in range(W)]
    bufs = [s.clone() for s in src]
    ref = torch.stack(src).sum(0)

    def once():
        for r in range(W):
            _lib.check(L.mfm_p2p_allreduce(hs[r], C.c_void_p(bufs[r].data_ptr()), n, C.c_void_p(streams[r].cuda_stream)), "ar")

    torch.cuda.synchronize()
    once()
    torch.cuda.synchronize()
    err = max(float((b - ref).abs().max() / ref.abs().max()) for b in bufs)
    same = all(torch.equal(bufs[0], b) for b in bufs)
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(W)]
    for r in range(W):
        ev[r][0].record(streams[r])
    for _ in range(iters):
        once()
    for r in range(W):
        ev[r][1].record(streams[r])
    torch.cuda.synchronize()
    us = max(1e3 * a.elapsed_time(b) / iters for a, b in ev)
    to = C.c_int32(0)
    L.mfm_p2p_status(hs[0], C.byref(to))
    print("in-process W=%d  n=%d floats  %.1f us/call  rel err %.1e  identical on all ranks %s  timed_out %d"
          % (W, n, us, err, same, to.value))
    for h in hs:
        L.mfm_p2p_destroy(h)


if __name__ == "__main__":
    for W in (2, 4, 8):
        run(W)
