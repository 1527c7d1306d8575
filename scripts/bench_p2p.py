"""Latency of the P2P gradient all-reduce kernel (csrc/p2p.hip) next to torch.distributed.all_reduce.

  python -m torch.distributed.run --nproc-per-node W --master-addr 127.0.0.1 scripts/bench_p2p.py [--one-device]

--one-device: every rank uses cuda:0 (gloo control plane): times the kernel's flag rounds and copies with local
HBM under the peer pointers, NOT xGMI -- the only form a 1-GPU box can run.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from factorized_amd import comm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--n", type=int, default=477294)
    ap.add_argument("--iters", type=int, default=300)
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.one_device:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    ar = comm.P2PAllReduce(world, rank, a.n)
    ok, worst = comm.validate(ar, world, rank, a.n, dev)
    buf = torch.zeros(a.n, device=dev)

    def timeit(fn):
        for _ in range(20):
            fn(buf)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn(buf)
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / a.iters

    t_p2p = timeit(ar)
    t_ref = timeit(comm.TorchAllReduce()) if not a.one_device else float("nan")
    if rank == 0:
        print("world %d  n %d floats (%.2f MB)  p2p-two-shot %.1f us/call (validated %s, worst rel err %.1e)  "
              "torch.distributed %.1f us/call%s"
              % (world, a.n, a.n * 4 / 1e6, t_p2p, ok, worst, t_ref,
                 "  [one device: local HBM under the peer pointers]" if a.one_device else ""))
    ar.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
