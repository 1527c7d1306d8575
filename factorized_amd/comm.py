"""Gradient collectives of the data-parallel step (the reference is single-GPU).

Two implementations behind one call, `allreduce(tensor)` = in-place fp32 sum over the ranks:

* `P2PAllReduce` -- libmfm_hip's two-shot exchange over peer-mapped staging buffers (csrc/p2p.hip):
  one kernel on the compute stream, no host synchronisation, no stream hand-over.
* `TorchAllReduce` -- `torch.distributed.all_reduce` (backend nccl = RCCL ring over xGMI).

`make_allreduce` sets the P2P path up, PROVES it on this job's devices against torch.distributed on
pseudo-random buffers (several back-to-back calls, so that a stale staging row or a lost flag shows up),
and lets every rank agree on the outcome; anything short of an exact set-up and a matching result on all
ranks selects the RCCL path.  Which one runs is reported (`.name`) and printed by bench.py.
"""
import ctypes as C
import os
import sys

import torch

from . import _lib


class TorchAllReduce:
    name = "rccl"

    def __call__(self, t):
        import torch.distributed as dist
        dist.all_reduce(t)

    def timed_out(self):
        return False

    def close(self):
        pass


class P2PAllReduce:
    """One staging block per rank, mapped by every peer (HIP IPC).  `exchange(bytes) -> [bytes]*W` is the
    out-of-band channel for the 64-byte handles (torch.distributed.all_gather_object by default)."""
    name = "p2p-two-shot"

    def __init__(self, world, rank, max_elems, exchange=None):
        L = _lib.lib()
        self._L = L
        self._h = None
        self.world, self.rank, self.max_elems = world, rank, int(max_elems)
        if exchange is None and world > 1:
            import torch.distributed as dist

            def exchange(b):
                out = [None] * world
                dist.all_gather_object(out, b)
                return out
        # Every rank takes part in both exchanges whatever happened locally, so that a failure on one rank
        # (IPC refused, allocation failed) raises on all of them instead of leaving the others waiting.
        err, mine = None, b""
        try:
            h = C.c_void_p()
            _lib.check(L.mfm_p2p_create(world, rank, self.max_elems, C.byref(h)), "mfm_p2p_create")
            self._h = h
            self._pid = os.getpid()
            if world > 1:
                nb = L.mfm_p2p_handle_bytes()
                buf = C.create_string_buffer(nb)
                _lib.check(L.mfm_p2p_export(h, buf), "mfm_p2p_export")
                mine = bytes(buf.raw)
        except _lib.MfmError as e:
            err = e
        if world == 1:
            if err is not None:
                raise err
            return
        handles = exchange(mine)
        status = b"ok"
        if err is not None:
            status = ("rank %d: %s" % (rank, err)).encode()
        elif len(handles) != world or any(len(x) != len(mine) for x in handles):
            status = ("rank %d: a peer exported no handle" % rank).encode()
        else:
            try:
                blob = C.create_string_buffer(b"".join(handles), len(mine) * world)
                _lib.check(L.mfm_p2p_connect(self._h, blob), "mfm_p2p_connect")
            except _lib.MfmError as e:
                status = ("rank %d: %s" % (rank, e)).encode()
        # second exchange = agreement + barrier: nobody raises a flag before every rank has mapped and
        # cleared its block
        bad = [x for x in exchange(status) if x != b"ok"]
        if bad:
            self.close()
            raise _lib.MfmError("P2PAllReduce set-up failed: " + b"; ".join(bad).decode(errors="replace"))

    def __call__(self, t):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise _lib.MfmError("P2PAllReduce: contiguous fp32 device tensor expected")
        stream = torch._C._cuda_getCurrentRawStream(t.device.index)
        _lib.check(self._L.mfm_p2p_allreduce(self._h, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(stream)),
                   "mfm_p2p_allreduce")

    def allreduce_adam(self, e, lr, grad_scale):
        """All-reduce of e.grads and the flat Adam update of e.params in the same launch."""
        check = getattr(e, "_require_uniform_steps", None)
        if check is not None:
            check("P2PAllReduce.allreduce_adam")      # one bias-correction step for all tensors: refuse after staged steps
        e.step_count += 1
        gs = getattr(e, "group_steps", None)
        if gs:
            for g in gs:
                gs[g] = e.step_count
        stream = torch._C._cuda_getCurrentRawStream(e.grads.device.index)
        # guard word of the gradient buffer (engine.FlatLayout.guard): a rank whose step lost a hand-over tells the others in
        # its first-push flags, and then NO rank applies the update (csrc/p2p.hip)
        guard = getattr(getattr(e, "layout", None), "guard", -1)
        _lib.check(self._L.mfm_p2p_allreduce_adam_guarded(self._h, C.c_void_p(e.grads.data_ptr()), C.c_void_p(e.params.data_ptr()),
                                                          C.c_void_p(e.adam_m.data_ptr()), C.c_void_p(e.adam_v.data_ptr()),
                                                          e.grads.numel(), e.step_count, lr, 0.9, 0.999, 1e-8, grad_scale,
                                                          int(guard), C.c_void_p(stream)), "mfm_p2p_allreduce_adam_guarded")

    def timed_out(self):
        v = C.c_int32(0)
        _lib.check(self._L.mfm_p2p_status(self._h, C.byref(v)), "mfm_p2p_status")
        return bool(v.value)

    def wait_stats(self, reset=True):
        """How long workgroup 0 of THIS rank spun in the two flag rounds since the last reset: dict(round1_us, round2_us per
        call, calls).  Round 1 waits for the slowest peer's first push (= how late that rank entered the exchange), round 2 for
        the slowest owner's result (synchronises; diagnosis of a scaling run)."""
        v = (C.c_int64 * 3)()
        _lib.check(self._L.mfm_p2p_wait_stats(self._h, v, 1 if reset else 0), "mfm_p2p_wait_stats")
        n = max(int(v[2]), 1)
        return dict(round1_us=round(v[0] / 100.0 / n, 2), round2_us=round(v[1] / 100.0 / n, 2), calls=int(v[2]))

    def close(self):
        if self._h is not None:
            if getattr(self, "_pid", None) == os.getpid():     # (not from a forked child: see engine._Plan.__del__)
                self._L.mfm_p2p_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _agree(ok, device):
    """True only if `ok` holds on every rank."""
    import torch.distributed as dist
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def validate(ar, world, rank, n, device, rounds=6):
    """Run `rounds` back-to-back all-reduces of rank- and round-dependent data through `ar` and compare with
    torch.distributed on the same data.  Returns (ok, worst relative error)."""
    import torch.distributed as dist
    gen = torch.Generator(device="cpu")
    bufs, refs = [], []
    for it in range(rounds):
        gen.manual_seed(1000 * it + rank)
        v = torch.randn(n, generator=gen).to(device)
        bufs.append(v)
        refs.append(v.clone())
    torch.cuda.synchronize(device)
    dist.barrier()
    fail = None
    try:
        for it in range(rounds):             # no synchronisation between the calls
            ar(bufs[it])
    except Exception as e:                   # keep going: the collectives below must stay aligned across ranks
        fail = e
    torch.cuda.synchronize(device)
    worst = 0.0
    for it in range(rounds):
        dist.all_reduce(refs[it])
        err = float((bufs[it] - refs[it]).abs().max() / refs[it].abs().max().clamp_min(1e-30))
        worst = max(worst, err)
    # bit-identical results on all ranks (every slice is reduced once, by its owner)
    chk = bufs[-1].clone()
    dist.all_reduce(chk, op=dist.ReduceOp.MAX)
    same = bool((chk == bufs[-1]).all().item())
    try:
        timed_out = ar.timed_out()
    except Exception as e:
        fail, timed_out = e, True
    return (worst < 1e-5 and same and not timed_out and fail is None), worst


def make_allreduce(world, rank, n, device, verbose=True):
    """The collective the data-parallel step uses: the P2P kernel when it can be set up and validated on
    every rank, torch.distributed (RCCL) otherwise.  MFM_ALLREDUCE=rccl|p2p forces the choice (p2p still
    has to pass the validation)."""
    want = os.environ.get("MFM_ALLREDUCE", "p2p")
    if world == 1 or want == "rccl" or world > 8:
        return TorchAllReduce()
    ar, why = None, ""
    try:
        ar = P2PAllReduce(world, rank, n)
        ok = True
    except Exception as e:       # set-up failed on this rank (IPC refused, no peer access ...)
        ok, why = False, "%s: %s" % (type(e).__name__, e)
    ok = _agree(ok, device)
    worst = float("nan")
    if ok:
        try:
            ok, worst = validate(ar, world, rank, n, device)
        except Exception as e:
            ok, why = False, "%s: %s" % (type(e).__name__, e)
        ok = _agree(ok, device)
    if verbose and rank == 0:
        print("[factorized_amd.comm] gradient all-reduce: %s (p2p set-up/validation %s, worst rel err %.2e%s)"
              % ("p2p-two-shot" if ok else "rccl", "ok" if ok else "FAILED", worst, (", " + why) if why else ""),
              file=sys.stderr, flush=True)
    if ok:
        return ar
    if ar is not None:
        ar.close()
    return TorchAllReduce()
