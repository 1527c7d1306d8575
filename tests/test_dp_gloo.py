"""Data-parallel path on CPU (gloo, world_size 2): the sharding, the single flat all-reduce and
the KLD-aware scaling of factorized_amd.train.DataParallelStep must reproduce ONE process stepping
on the global batch.  The HIP engine cannot run here, so the step is driven through an adapter
with the engine's interface whose arithmetic is the CPU oracle (test infrastructure only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from factorized_amd import configs, synth, train
from oracle import mfm_oracle as O


class OracleEngine:
    """Engine-shaped adapter (forward/backward/adam on flat buffers) around the CPU oracle."""

    def __init__(self, cfgs, weights):
        self.cfg = cfgs[0]
        self.model = O.build("kl_ef", cfgs)
        O.load_numpy_weights(self.model, weights)
        self.model.train()
        self.reg_scale = 1.0
        self.params = torch.nn.utils.parameters_to_vector(self.model.parameters()).detach().clone()
        self.grads = torch.zeros_like(self.params)
        self.opt_m = torch.zeros_like(self.params)
        self.opt_v = torch.zeros_like(self.params)
        self.t = 0

    def forward(self, x, y, train=True, want_xhat=False):
        self._terms = O.loss_terms(self.model, x, y, self.cfg)
        return {"losses": self._terms["loss"].detach()}

    def backward(self, x, y, stage=0):
        t = self._terms
        loss = t["disc"] + t["gen"] + self.cfg["lda_mmd"] * self.reg_scale * t["reg"]
        self.model.zero_grad()
        loss.backward()
        self.grads.copy_(torch.nn.utils.parameters_to_vector([p.grad for p in self.model.parameters()]))

    def grad_step(self, x, y, check=False):
        out = self.forward(x, y)
        self.backward(x, y)
        return out["losses"]

    def adam(self, lr=1e-3, grad_scale=1.0):
        self.t += 1
        g = self.grads * grad_scale
        self.opt_m.mul_(0.9).add_(g, alpha=0.1)
        self.opt_v.mul_(0.999).addcmul_(g, g, value=0.001)
        bc1, bc2 = 1 - 0.9 ** self.t, 1 - 0.999 ** self.t
        self.params -= (lr / bc1) * self.opt_m / (self.opt_v.sqrt() / np.sqrt(bc2) + 1e-8)
        torch.nn.utils.vector_to_parameters(self.params, self.model.parameters())

    def train_step(self, x, y, lr=1e-3, check=False):
        out = self.forward(x, y)
        self.backward(x, y)
        self.adam(lr=lr)
        return out["losses"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    B, T, steps = 8, 5, 3
    shapes = O.state_shapes(O.build("kl_ef", cfgs))
    w = synth.make_weights(shapes, seed=1234)
    X, Y = synth.make_dataset(cfg["input_dims"], B * world * steps, T, seed=11)
    e = OracleEngine(cfgs, w)
    train.broadcast_params(e, world)
    stepper = train.DataParallelStep(e, world, lr=1e-3)
    assert e.reg_scale == world
    nb = X.shape[1] // B
    mine = train.shard_batches(nb, rank, world)
    assert len(mine) == steps
    first_grad = None
    for i in mine:
        x = torch.from_numpy(np.ascontiguousarray(X[:, i * B:(i + 1) * B]))
        y = torch.from_numpy(Y[i * B:(i + 1) * B])
        stepper.step(x, y)
        if first_grad is None:
            first_grad = e.grads.clone() / world      # all-reduced sum, scaled as the fused Adam does
    ret[rank] = e.params.clone()
    ret["g%d" % rank] = first_grad
    dist.destroy_process_group()


def test_dp2_equals_single_process_global_batch():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert torch.equal(ret[0], ret[1])                      # replicas stay identical
    # single process, global batch = the two shards side by side, same step order
    torch.set_num_threads(1)
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    B, T, steps = 8, 5, 3
    shapes = O.state_shapes(O.build("kl_ef", cfgs))
    w = synth.make_weights(shapes, seed=1234)
    X, Y = synth.make_dataset(cfg["input_dims"], B * world * steps, T, seed=11)
    e = OracleEngine(cfgs, w)
    for s in range(steps):
        lo = s * world * B
        x = torch.from_numpy(np.ascontiguousarray(X[:, lo:lo + world * B]))
        y = torch.from_numpy(Y[lo:lo + world * B])
        e.train_step(x, y)
        if s == 0:
            # the gradient algebra itself: sum_r grad(mean_r + W*KLD_r) / W == grad(global loss)
            gerr = (ret["g0"] - e.grads).abs().max().item() / e.grads.abs().max().item()
            assert gerr < 1e-5, gerr
    # parameters after 3 Adam steps (Adam normalises tiny gradients, so allow a few 1e-4)
    err = (ret[0] - e.params).abs().max().item() / e.params.abs().max().item()
    assert err < 5e-4, err


def test_shard_batches_partition():
    for nb, W in ((40, 8), (41, 8), (7, 2), (3, 4)):
        seen = []
        for r in range(W):
            b = train.shard_batches(nb, r, W)
            assert len(b) == nb // W
            seen += b
        assert len(set(seen)) == len(seen) and all(0 <= i < nb for i in seen)


def test_device_dataset_layout_cpu():
    cfg = configs.canonical_configs()[0]
    ds = train.DeviceDataset(cfg, 70, 20, 32, "cpu", seed=11)
    assert ds.nb == 2 and tuple(ds.X.shape) == (2, 20, 32, 325) and ds.X.is_contiguous()
    X, _ = synth.make_dataset(cfg["input_dims"], 70, 20, seed=11)
    assert np.array_equal(ds.batch(1)[0].numpy(), X[:, 32:64])     # contiguous column slice (mfm_mosi.py:425-429)


def _comm_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from factorized_amd import comm
    dev = torch.device("cpu")
    # no GPU here: the P2P set-up fails on every rank, all ranks must learn it together (no rank is left
    # waiting in a collective) and the RCCL/torch.distributed path must be chosen
    ar = comm.make_allreduce(world, rank, 1000, dev, verbose=False)
    v = torch.full((1000,), float(rank + 1))
    ar(v)
    os.environ["MFM_ALLREDUCE"] = "rccl"
    forced = comm.make_allreduce(world, rank, 1000, dev, verbose=False)
    ret[rank] = (ar.name, forced.name, float(v[0]), float(v[-1]), ar.timed_out())
    dist.destroy_process_group()


def test_allreduce_selection_falls_back_together():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_comm_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == ("rccl", "rccl", 3.0, 3.0, False), ret[r]


def test_allreduce_single_rank_is_torch_path():
    from factorized_amd import comm
    assert comm.make_allreduce(1, 0, 10, torch.device("cpu")).name == "rccl"
