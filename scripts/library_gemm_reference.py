"""What the vendor GEMM (rocBLAS / hipBLASLt behind torch.mm) reaches on the large-batch products of the MFM step.

A yardstick for the hand-written grouped GEMM kernels only: nothing in the product path calls it.
Usage: python scripts/library_gemm_reference.py [B] [T]   (defaults 2048 20; MOSI canonical sizes)
"""
import sys
import torch


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    R = T * B
    dev = "cuda"
    torch.backends.cuda.matmul.allow_tf32 = False
    print(f"# torch {torch.__version__}, rows T*B = {R}, fp32 and bf16 operands, fp32 peak 157.3 TF/s")
    print(f"{'product':34s} {'M':>6s} {'N':>6s} {'K':>6s} {'fp32 us':>9s} {'TF/s':>7s} {'bf16 us':>9s} {'TF/s':>7s}")
    # weight gradients dW[M,N] = dA[R,M]^T . X[R,N]
    dw = [("dW ef   dA^T [x|h]", 480, 445), ("dW dec_l dA^T [h|h]", 416, 208), ("dW dec_a", 96, 48), ("dW dec_v", 96, 48),
          ("dW fc1_l dx^T h", 300, 104), ("dW fc1_v", 20, 24)]
    tot32 = tot16 = 0.0
    for name, M, N in dw:
        a = torch.randn(R, M, device=dev)
        x = torch.randn(R, N, device=dev)
        t32 = bench(lambda: torch.mm(a.t(), x))
        ab, xb = a.bfloat16(), x.bfloat16()
        t16 = bench(lambda: torch.mm(ab.t(), xb))
        fl = 2.0 * M * N * R
        tot32 += t32; tot16 += t16
        print(f"{name:34s} {M:6d} {N:6d} {R:6d} {t32:9.1f} {fl / t32 / 1e6:7.1f} {t16:9.1f} {fl / t16 / 1e6:7.1f}")
    print(f"{'sum of the weight-gradient products':55s} {tot32:9.1f} {'':7s} {tot16:9.1f}")
    # projections pre[R,M] = X[R,K] . W[M,K]^T
    pr = [("proj ef  x W_ih^T", 480, 325), ("fc1_l  h W^T", 300, 104), ("dH_l   dx W", 104, 300)]
    tot32 = tot16 = 0.0
    for name, M, K in pr:
        x = torch.randn(R, K, device=dev)
        w = torch.randn(M, K, device=dev)
        t32 = bench(lambda: torch.mm(x, w.t()))
        xb, wb = x.bfloat16(), w.bfloat16()
        t16 = bench(lambda: torch.mm(xb, wb.t()))
        fl = 2.0 * M * K * R
        tot32 += t32; tot16 += t16
        print(f"{name:34s} {R:6d} {M:6d} {K:6d} {t32:9.1f} {fl / t32 / 1e6:7.1f} {t16:9.1f} {fl / t16 / 1e6:7.1f}")
    print(f"{'sum of the row-major products':55s} {tot32:9.1f} {'':7s} {tot16:9.1f}")


if __name__ == "__main__":
    main()
