"""Training-loop pieces around MFMEngine: device-resident batching, the data-parallel step and
the reference's epoch loop (train_mfm, reference mfm_mosi.py:386-503) restated for Python 3.

Data parallelism (the reference has none): one process per GPU, each rank
draws its own B-sample shard, ONE all-reduce of the flat gradient buffer per step over RCCL/xGMI.
The joint loss mixes batch-MEAN terms (L1/CE, MSE) with a batch-SUM term (KLD, mfm_model.py:37),
so to equal a single-process step at the global batch W*B each rank back-propagates
`mean-terms + W * KLD` (reg_scale = W inside the plan) and the summed gradients are divided by W
(grad_scale = 1/W inside the fused Adam).  tests/test_dp_gloo.py checks that algebra on CPU.
"""
import os

import numpy as np
import torch

from . import synth


def dp_env():
    import os
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_batches(n_batches, rank, world):
    """Rank r takes batches r, r+W, r+2W, ... (equal count on every rank: the tail that does not
    fill a round is dropped, like the reference drops the tail samples, mfm_mosi.py:423)."""
    per = n_batches // world
    return [rank + i * world for i in range(per)]


class DeviceDataset:
    """Synthetic MOSI-shape split kept resident in HBM as contiguous [nb, T, B, D] batches
    (the reference slices a host array and copies H2D every step, mfm_mosi.py:428-429)."""

    def __init__(self, cfg, n_samples, T, batchsize, device, seed=11):
        loss = cfg.get("loss", "l1")
        classes = cfg["output_dim"] if loss == "ce" else 0
        X, y = synth.make_dataset(cfg["input_dims"], n_samples, T, seed=seed, output_dim=cfg["output_dim"],
                                  classes=classes)
        nb = n_samples // batchsize                       # floor: tail dropped (mfm_mosi.py:423)
        X = X[:, :nb * batchsize].reshape(T, nb, batchsize, -1).transpose(1, 0, 2, 3)
        y = y[:nb * batchsize].reshape((nb, batchsize) + y.shape[1:])
        self.X = torch.from_numpy(np.ascontiguousarray(X)).to(device)
        self.y = torch.from_numpy(np.ascontiguousarray(y)).to(device)
        self.nb = nb

    @classmethod
    def from_arrays(cls, X, y, batchsize, device):
        """X [T, N, D] time-major float32 (what `data.assemble` / the reference's `swapaxes(0,1)` produce), y [N] or
        [N, k] -> the same HBM-resident `[nb, T, B, D]` batch layout; the tail that does not fill a batch is dropped
        (mfm_mosi.py:423)."""
        self = cls.__new__(cls)
        T, N = X.shape[0], X.shape[1]
        nb = N // batchsize
        Xb = np.ascontiguousarray(X[:, :nb * batchsize].reshape(T, nb, batchsize, -1).transpose(1, 0, 2, 3))
        yb = np.ascontiguousarray(np.asarray(y)[:nb * batchsize].reshape((nb, batchsize) + tuple(np.asarray(y).shape[1:])))
        self.X = torch.from_numpy(Xb).to(device)
        self.y = torch.from_numpy(yb).to(device)
        self.nb = nb
        return self

    def batch(self, i):
        return self.X[i], self.y[i]


def _mark_shared_device(engine):
    """Ranks that drive the SAME GPU (one-device test set-ups) must not use the in-launch hand-overs of the small-batch step
    (plan option "handover": two queues' launches can block each other's producers): compare (host, device) across the ranks and,
    when two ranks collide, switch the hand-overs of this engine off (and export MFM_SHARED_DEVICE=1, the default for engines
    created later).  One process per GPU -- the deployment -- is unaffected."""
    try:
        import socket
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
            return
        dev = torch.device(getattr(engine, "device", "cuda"))
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        try:
            ident = str(torch.cuda.get_device_properties(idx).uuid)
        except Exception:
            ident = "%s:%d:%s" % (socket.gethostname(), idx, os.environ.get("HIP_VISIBLE_DEVICES", ""))
        mine = (socket.gethostname(), ident)
        allv = [None] * dist.get_world_size()
        dist.all_gather_object(allv, mine)
        if len(set(allv)) < len(allv):
            os.environ["MFM_SHARED_DEVICE"] = "1"          # (engines / plans created from now on)
            if hasattr(engine, "set_handover"):
                engine.set_handover(False)                 # this engine and its existing plans
    except Exception:
        pass


class DataParallelStep:
    """step(x, y): fused single-GPU step, or fwd/bwd + all-reduce + Adam when world > 1."""

    def __init__(self, engine, world=1, lr=1e-3, allreduce=None, rank=None):
        self.e = engine
        self.world = world
        self.lr = lr
        if world > 1:
            # per-rank dropout streams: the latent kernels key their masks by
            # (seed, call counter, op, row, column), and every rank numbers its rows 0..B-1 -- with one
            # shared seed the W shards of a global batch would all draw the same masks
            r = dp_env()[0] if rank is None else int(rank)
            if hasattr(engine, "seed"):
                engine.seed = int(engine.seed) + 7919 * r
        self.allreduce = allreduce            # comm.make_allreduce(...); default torch.distributed (RCCL)
        if world > 1:
            # the KLD is a batch SUM (scaled by W so that the averaged gradient equals the global-batch one); the MMD
            # regulariser of `MFM` is a statistic of the shard it is computed on: the data-parallel objective uses the
            # mean over ranks of the shard-local MMDs, i.e. no extra factor
            engine.reg_scale = 1.0 if getattr(engine, "variant", "kl_ef") == "mmd" else float(world)
            if allreduce is None:
                from . import comm
                self.allreduce = comm.TorchAllReduce()
            _mark_shared_device(engine)
        self.preflight_report = None         # dict once the first step ran the pre-flight (world > 1)

    # C1 of the latent kernels' mask stream (csrc/plan.hip: seed * C1 + calls * C2): with C1 odd, the seed that makes call n + 1
    # draw the masks of call n is seed - C2 / C1 (mod 2^64)
    _C1, _C2 = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03

    def _preflight(self, x, y):
        """The deployment composition -- role-workgroup grad_step (in-launch hand-overs) next to a live P2P mapping -- cannot run
        on a 1-GPU box (ranks that share a device switch the hand-overs off), so the first multi-GPU job is the first time it
        runs.  Before it is trusted: three role-workgroup gradient steps against the separate launches on the same batch and
        the same dropout masks, on every rank; all ranks must agree, else every rank uses the separate launches."""
        e = self.e
        rep = dict(ran=False, ok=True, worst=0.0, roles=False)
        import torch.distributed as dist
        from . import comm
        want = bool(getattr(e, "handover", False)) and hasattr(e, "set_handover")
        ok, worst, roles = True, 0.0, False
        if want:
            seed0 = e.seed
            inv = pow(self._C1, -1, 1 << 64)
            try:
                for it in range(3):
                    e.seed = (seed0 + 1000003 * it) & ((1 << 64) - 1)
                    e.set_handover(False)
                    e.grad_step(x, y, check=False)
                    ref = e.grads.clone()
                    e.seed = (e.seed - self._C2 * inv) & ((1 << 64) - 1)          # the next call draws the same masks
                    e.set_handover(True)
                    e.grad_step(x, y, check=False)
                    p = e.plan(x.shape[0], x.shape[1])
                    roles = roles or bool(p.get_option("dw_roles_active")) or bool(p.get_option("proj_roles_active"))
                    d = float((e.grads - ref).abs().max())
                    sc = float(ref.abs().max())
                    worst = max(worst, d / max(sc, 1e-30))
                    ok = ok and (d <= 1e-5 * sc + 1e-7) and bool(torch.isfinite(e.grads).all())
                ok = ok and e.check_status(raise_on_error=False) == 0
            except Exception as ex:          # (keep the collective below aligned across ranks)
                ok, rep["error"] = False, "%s: %s" % (type(ex).__name__, ex)
            e.seed = seed0
        # every rank takes part in the agreement, whatever it did locally
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            ok = comm._agree(ok, getattr(e, "device", "cpu"))
        if want and not ok:
            e.set_handover(False)
            if dp_env()[0] == 0:
                import sys
                sys.stderr.write("[factorized_amd.train] pre-flight: role-workgroup grad_step disagreed with the separate launches "
                                 "on some rank (worst rel %.2e here): every rank uses the separate launches\n" % worst)
        rep.update(ran=want, ok=ok, worst=worst, roles=roles)
        self.preflight_report = rep

    def step(self, x, y):
        e = self.e
        if self.world == 1:
            return e.train_step(x, y, lr=self.lr, check=False)
        if self.preflight_report is None:
            self._preflight(x, y)
        losses = e.grad_step(x, y, check=False)           # fwd + bwd: one enqueue, 10 launches
        fused = getattr(self.allreduce, "allreduce_adam", None)
        if fused is not None and os.environ.get("MFM_DP_FUSED_ADAM", "1") == "1":
            fused(e, self.lr, 1.0 / self.world)           # P2P kernel: collective + Adam in one launch
        else:
            self.allreduce(e.grads)                       # one flat buffer, one collective
            e.adam(lr=self.lr, grad_scale=1.0 / self.world)
        return losses


def broadcast_params(engine, world):
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(engine.params, src=0)


class GraphedModuleStep:
    """The reference's training step for the module-path classes (`MFM`, `MFM_KL`; train() of
    mfm_mosi.py:419-443: zero_grad, model.forward, L1 + sum lambda MSE + lambda reg + missing, backward,
    optim.Adam.step) captured ONCE into a hipGraph and replayed per batch.

    The module path issues ~400 small launches per step and is bound by the host's enqueue rate; replaying
    the captured graph halves the step (MFM_KL 5.9 -> 2.7 ms, MFM 5.0 -> 2.6 ms at B=32, T=20).  Inputs are
    copied into static buffers, losses stay on the device (the reference's per-step `.item()` is the caller's
    choice), the learning rate is a device scalar (`set_lr`) so ReduceLROnPlateau keeps working.
    Round 4: `MFM_KL_EF` and the fused `MFM_KL` are captured too -- the reference's UNCHANGED loop around ONE plan call per
    direction.  What a captured launch freezes (kernel arguments) is split from what must change per replay: the plan's
    dropout streams and the epochs of its in-launch hand-overs add device words that a tick node inside the graph advances
    (csrc/plan.hip), and the optimizer is `factorized_amd.optim.Adam(capturable=True)` -- one flat launch whose step count and
    learning rate are device words.  (`model.engine.train_step(x, y)`, no torch loss ops at all, is still the fastest form.)

    Dropout stays random under replay: torch's generators advance a device offset, the MFN memory kernel and the fused plans
    add a device word that the graph itself advances to their host seeds (MfmMemDesc.seed_dev, mfm_plan_state_layout).
    """

    def __init__(self, model, cfg, B, T, lr=1e-3, warmup=3):
        dev = next(model.parameters()).device
        self.model, self.cfg = model, cfg
        fused = hasattr(model, "_fast_ok") and model._fast_ok() and (
            type(model).__name__ == "MFM_KL_EF" or getattr(model, "fused_forward", False))
        self.fused = fused
        d = cfg["input_dims"]
        self.x = torch.zeros(T, B, sum(d), device=dev)
        ce = cfg.get("loss", "l1") == "ce"
        self.y = torch.zeros(B, dtype=torch.int64, device=dev) if ce else \
            torch.zeros((B,) if cfg["output_dim"] == 1 else (B, cfg["output_dim"]), device=dev)
        self.lr = torch.tensor([float(lr)], device=dev) if fused else torch.tensor(float(lr), device=dev)
        if fused:
            from . import optim as our_optim
            self.opt = our_optim.Adam(model.parameters(), lr=self.lr, capturable=True)
        else:
            self.opt = torch.optim.Adam(model.parameters(), lr=self.lr, capturable=True)
        disc_fn = torch.nn.CrossEntropyLoss() if ce else torch.nn.L1Loss()
        mse = torch.nn.MSELoss()
        self._snap = self._snap_step = None

        def step(apply=True):
            # (fused models: zero_grad() costs no launch -- the forward's first launch clears the flat gradient buffer)
            self.opt.zero_grad(set_to_none=fused)
            (xl, xa, xv, yh), reg, miss = model.forward(self.x)
            x = self.x
            yhat = yh.squeeze(1) if (not ce and cfg["output_dim"] == 1) else yh
            disc = disc_fn(yhat, self.y)
            gen = cfg["lda_xl"] * mse(xl, x[:, :, :d[0]]) + cfg["lda_xa"] * mse(xa, x[:, :, d[0]:d[0] + d[1]]) \
                + cfg["lda_xv"] * mse(xv, x[:, :, d[0] + d[1]:])
            loss = disc + gen + cfg["lda_mmd"] * reg + miss
            loss.backward()
            if apply:
                self.opt.step()
            from .lazy import LossExpr, SnapshotStep
            if isinstance(loss, LossExpr):
                # symbolic losses: ONE copy node of the plan's 64-byte state block; the returned expressions read the copy
                # (valid until the next replay, whatever else runs on the plan in between)
                plan = model.engine.plan(T, B)
                if self._snap is None:
                    self._snap = torch.zeros_like(plan.state)
                    self._snap_step = SnapshotStep(self._snap, self.x)
                self._snap.copy_(plan.state)
                return LossExpr(self._snap_step, loss._coef, loss._const), LossExpr(self._snap_step, disc._coef, disc._const)
            return loss.detach(), disc.detach()

        # the warm-up steps below run on the (zero) static batch: keep them from training the model
        saved = [p.detach().clone() for p in model.parameters()]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator warm-up + lazily built caches, outside the capture
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for p, q in zip(model.parameters(), saved):
                p.copy_(q)
            for st in self.opt.state.values():          # Adam moments and step counter back to zero, in place
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            if fused:
                self.opt.reset_state()
        self._step_fn = step
        self.recaptures = 0
        self._capture()

    def _capture(self):
        torch.cuda.synchronize(self.x.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.disc = self._step_fn()

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def step(self, x, y):
        """One training step on batch (x [T,B,D], y); returns (loss, disc_loss) device scalars (no sync)."""
        if self.fused and self.model.engine.poll_status():
            # a hand-over inside a replayed launch gave up (another process / stream kept its producers off the GPU): the guarded
            # Adam skipped that step.  The captured launches would go on using the hand-overs -- plan options do not reach into a
            # captured graph -- so: clear the status, switch the engine to separate launches and capture the step again.
            import warnings
            eng = self.model.engine
            eng.check_status(raise_on_error=False)
            warnings.warn(eng.status_message() + "  (GraphedModuleStep: step re-captured on separate launches)", RuntimeWarning,
                          stacklevel=2)
            self.recaptures += 1
            # one forward / backward on the separate launches outside the capture (whatever they build lazily is built now;
            # no optimizer step: the parameters stay where they are), then the new graph
            self._step_fn(False)
            self._capture()
        self.x.copy_(x)
        self.y.copy_(y)
        self.graph.replay()
        return self.loss, self.disc


class GraphedStep:
    """Any training step of a module-path model -- `loss = loss_fn(model, *inputs)`, backward, Adam -- captured once into a
    hipGraph and replayed: the generic form of GraphedModuleStep for the classes whose outputs do not have the
    (decoded, reg, missing) shape of the reference's main loop (the ablations M_A-M_D, MFM_missing, seq2seq, basic_missing:
    reference mfm_model.py:201-467, 766-1017).  `inputs`: example device tensors; step(*tensors) copies into static buffers.
    These classes issue 150-900 small launches per step from Python; replay removes the host from the step."""

    def __init__(self, model, loss_fn, inputs, lr=1e-3, warmup=3):
        dev = next(model.parameters()).device
        self.model = model
        self.static = [t.detach().clone() for t in inputs]
        self.lr = torch.tensor(float(lr), device=dev)
        self.opt = torch.optim.Adam(model.parameters(), lr=self.lr, capturable=True)

        def step():
            self.opt.zero_grad(set_to_none=False)
            loss = loss_fn(model, *self.static)
            loss.backward()
            self.opt.step()
            return loss.detach()

        saved = [p.detach().clone() for p in model.parameters()]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for p, q in zip(model.parameters(), saved):
                p.copy_(q)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step()

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def step(self, *tensors):
        for s, t in zip(self.static, tensors):
            s.copy_(t)
        self.graph.replay()
        return self.loss
