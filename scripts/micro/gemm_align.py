import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from factorized_amd import engine as E

def run(M, N, K, lda, iters=20):
    a = torch.randn(M, lda, device="cuda"); b = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
    d = [E.make_gemm(a, b, c, M, N, K, a_sm=lda, a_sk=1, b_sk=1, b_sn=K, ldc=N)]
    for _ in range(3): E.gemm_grouped(d)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): E.gemm_grouped(d)
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / iters

for (M, N, K) in [(40960, 960, 325), (40960, 960, 324), (40960, 960, 328), (640, 960, 325)]:
    for lda in (325, 328, 336):
        if lda < K: continue
        us = run(M, N, K, lda)
        print("M=%d N=%d K=%d lda=%d  %8.1f us  %5.1f TF/s" % (M, N, K, lda, us, 2.0 * M * N * K / us / 1e6))
