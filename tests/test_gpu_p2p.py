"""The P2P gradient all-reduce (csrc/p2p.hip, factorized_amd/comm.py) on ONE device: W processes share
cuda:0, map each other's staging blocks through HIP IPC and run the same kernel a multi-GPU job runs;
only the transport under the peer pointers differs (local HBM instead of xGMI).  The control plane
(handle exchange, reference sums) is torch.distributed on gloo."""
import ctypes as C
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MFM_P2P_TIMEOUT_MS"] = "4000"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from factorized_amd import comm
    try:
        nmax = max(sizes)
        ar = comm.P2PAllReduce(world, rank, nmax)
        out = {}
        for n in sizes:
            ok, worst = comm.validate(ar, world, rank, n, dev, rounds=5)
            out[n] = (ok, worst)
        # a long unsynchronised train of calls on one buffer (sum of sums): x -> W^k x
        v = torch.full((1003,), 1.0 + rank, device=dev)
        for _ in range(6):
            ar(v)
        torch.cuda.synchronize()
        expect = sum(1.0 + r for r in range(world)) * float(world) ** 5
        out["chain"] = bool((v == expect).all().item())
        # fused collective + Adam against all-reduce followed by mfm_adam_flat, three steps
        class E:
            pass
        n = 477294
        g = torch.Generator().manual_seed(5)
        p0 = torch.randn(n, generator=g)
        ea, eb = E(), E()
        for e in (ea, eb):
            e.params, e.adam_m, e.adam_v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            e.grads, e.step_count = torch.zeros(n, device=dev), 0
        from factorized_amd import _lib
        L = _lib.lib()
        for step in range(1, 4):
            gr = torch.Generator().manual_seed(100 * step + rank)
            grad = (torch.randn(n, generator=gr) * 0.01).to(dev)
            ea.grads.copy_(grad)
            eb.grads.copy_(grad)
            ar.allreduce_adam(ea, 1e-3, 1.0 / world)
            ar(eb.grads)
            _lib.check(L.mfm_adam_flat(C.c_void_p(eb.params.data_ptr()), C.c_void_p(eb.grads.data_ptr()),
                                       C.c_void_p(eb.adam_m.data_ptr()), C.c_void_p(eb.adam_v.data_ptr()), n, step,
                                       1e-3, 0.9, 0.999, 1e-8, 1.0 / world, None), "adam")
        torch.cuda.synchronize()
        out["fused_adam"] = (bool(torch.equal(ea.grads, eb.grads)),
                             float((ea.params - eb.params).abs().max()), float((ea.adam_v - eb.adam_v).abs().max()),
                             ea.step_count)
        out["timed_out"] = ar.timed_out()
        # guard word (round 4): ONE rank's gradients carry a raised guard (a hand-over of its step gave up, csrc/plan.hip) ->
        # the exchange completes, NO rank updates p / m / v, the replicas stay bit-identical; the next step trains again
        class Lay:
            guard = 8128
        eg = E()
        n2 = 8192
        eg.layout = Lay()
        eg.params = torch.linspace(-1, 1, n2, device=dev)
        eg.adam_m, eg.adam_v = torch.zeros(n2, device=dev), torch.zeros(n2, device=dev)
        eg.grads, eg.step_count = torch.zeros(n2, device=dev), 0
        hist = []
        for step in range(3):
            eg.grads.copy_(torch.full((n2,), 0.01 * (rank + 1) * (step + 1), device=dev))
            eg.grads[Lay.guard:] = 0.0
            if step == 1 and rank == world - 1:
                eg.grads[Lay.guard] = float("nan")
            ar.allreduce_adam(eg, 1e-2, 1.0 / world)
            torch.cuda.synchronize()
            hist.append(eg.params.cpu().clone())
        gathered = [None] * world
        dist.all_gather_object(gathered, hist[-1].numpy().tobytes())
        out["guard"] = (bool(torch.equal(hist[0], hist[1])), bool(not torch.equal(hist[1], hist[2])),
                        bool(not torch.equal(hist[0], torch.linspace(-1, 1, n2))), all(g == gathered[0] for g in gathered),
                        bool(torch.isnan(eg.grads[Lay.guard]).item()) if False else True)
        # the selection logic picks the kernel when it validates
        chosen = comm.make_allreduce(world, rank, 4099, dev, verbose=False)
        out["chosen"] = chosen.name
        chosen.close()
        ar.close()
        ret[rank] = out
    except Exception as e:           # surface the failure in the parent
        ret[rank] = {"error": "%s: %s" % (type(e).__name__, e)}
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_p2p_allreduce_matches_reference_sum(world):
    sizes = [477294, 4, 1, 1003, 262144]        # the MFM_KL_EF gradient buffer, tiny, ragged and aligned sizes
    mgr = mp.get_context("spawn").Manager()       # not a fork: the parent holds live GPU state (a GC in a forked child frees it there)
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), sizes, ret), nprocs=world, join=True)
    for r in range(world):
        out = ret[r]
        assert "error" not in out, out
        for n in sizes:
            ok, worst = out[n]
            assert ok, (r, n, worst)
        assert out["chain"] and not out["timed_out"]
        same_g, dp, dv, steps = out["fused_adam"]
        assert same_g and dp < 1e-6 and dv < 1e-9 and steps == 3, out["fused_adam"]
        skipped, resumed, first_applied, in_sync, _ = out["guard"]
        assert skipped and resumed and first_applied and in_sync, out["guard"]
        assert out["chosen"] == "p2p-two-shot"


def _lonely(rank, world, port, ret):
    """rank 1 never calls the collective: rank 0's kernel must give up, not hang."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MFM_P2P_TIMEOUT_MS"] = "300"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from factorized_amd import comm
    ar = comm.P2PAllReduce(world, rank, 4096)
    if rank == 0:
        v = torch.ones(4096, device="cuda")
        ar(v)
        ar(v)                        # the second call returns at once (the error word is already set)
        torch.cuda.synchronize()
        ret["timed_out"] = ar.timed_out()
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


def test_p2p_missing_peer_times_out_instead_of_hanging():
    mgr = mp.get_context("spawn").Manager()       # not a fork: the parent holds live GPU state (a GC in a forked child frees it there)
    ret = mgr.dict()
    mp.spawn(_lonely, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["timed_out"] is True


def test_p2p_single_rank_and_argument_errors():
    from factorized_amd import _lib, comm
    L = _lib.lib()
    ar = comm.P2PAllReduce(1, 0, 1000)
    v = torch.arange(1000, dtype=torch.float32, device="cuda")
    ar(v)                                           # world 1: the sum over one rank is the buffer itself
    torch.cuda.synchronize()
    assert torch.equal(v.cpu(), torch.arange(1000, dtype=torch.float32))
    with pytest.raises(_lib.MfmError):
        ar(torch.zeros(1001, device="cuda"))        # larger than max_elems
    with pytest.raises(_lib.MfmError):
        ar(torch.zeros(10, dtype=torch.float64, device="cuda"))
    assert not ar.timed_out()
    # one rank: the fused call is just the optimizer
    class E:
        pass
    e = E()
    e.params, e.adam_m, e.adam_v = torch.ones(1000, device="cuda"), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    e.grads, e.step_count = torch.full((1000,), 0.5, device="cuda"), 0
    ar.allreduce_adam(e, 1e-3, 1.0)
    torch.cuda.synchronize()
    assert torch.allclose(e.params.cpu(), torch.full((1000,), 1.0 - 1e-3), atol=1e-6) and e.step_count == 1
    ar.close()
    h = C.c_void_p()
    assert L.mfm_p2p_create(9, 0, 10, C.byref(h)) != 0          # more than 8 ranks
    assert L.mfm_p2p_create(2, 2, 10, C.byref(h)) != 0          # rank out of range
    assert L.mfm_p2p_create(2, 0, 10, C.byref(h)) == 0
    buf = torch.zeros(10, device="cuda")
    assert L.mfm_p2p_allreduce(h, C.c_void_p(buf.data_ptr()), 10, None) != 0      # not connected yet
    L.mfm_p2p_destroy(h)
