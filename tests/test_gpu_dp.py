"""The data-parallel step END TO END on the HIP engine (round-1 review: the chain reg_scale = W inside the plan
-> mfm_plan_grad_step -> P2P all-reduce (+ fused Adam) -> grad_scale = 1/W had never run against the oracle).

W processes share cuda:0 (the 1-GPU box): each runs the real MFMEngine on its own shard of a W*B global batch and
the real P2P kernel through HIP IPC; the control plane is torch.distributed on gloo.  Expected result: the
parameters of ONE oracle process stepping on the concatenated global batch (SURVEY.md section 8e semantics)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mfm_oracle as O
from factorized_amd import configs, synth
from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, STEPS = 16, 7, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fused, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MFM_P2P_TIMEOUT_MS"] = "5000"
    os.environ["MFM_DP_FUSED_ADAM"] = "1" if fused else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    try:
        from factorized_amd import comm, engine, train
        cfgs = configs.canonical_configs(dropout=False)
        cfg = cfgs[0]
        e = engine.MFMEngine(cfgs, device="cuda:0")
        e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
        ar = comm.P2PAllReduce(world, rank, e.grads.numel())
        stepper = train.DataParallelStep(e, world, lr=1e-3, allreduce=ar, rank=rank)
        assert e.reg_scale == float(world) and e.seed == 1234 + 7919 * rank
        xg, yg = synth.make_batch(cfg["input_dims"], world * B, T, seed=17)
        x = torch.from_numpy(np.ascontiguousarray(xg[:, rank * B:(rank + 1) * B])).cuda()
        y = torch.from_numpy(np.ascontiguousarray(yg[rank * B:(rank + 1) * B])).cuda()
        losses = []
        for _ in range(STEPS):
            l = stepper.step(x, y)
            losses.append(e.loss_dict(l))
        torch.cuda.synchronize()
        p = e.params.cpu()
        lo, hi = p.clone(), p.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out = {"in_sync": bool(torch.equal(lo, hi)), "timed_out": ar.timed_out(), "steps": e.step_count,
               "disc0": losses[0]["disc"], "reg0": losses[0]["reg"], "gen0": losses[0]["gen"]}
        if rank == 0:
            out["params"] = {n: v.cpu().numpy() for n, v in e.param_views().items()}
        dist.barrier()
        ar.close()
        ret[rank] = out
    except Exception as ex:
        import traceback
        ret[rank] = {"error": "%s: %s\n%s" % (type(ex).__name__, ex, traceback.format_exc())}
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fused", [(2, True), (2, False), (4, True)])
def test_hip_data_parallel_step_equals_oracle_on_global_batch(world, fused):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fused, ret), nprocs=world, join=True)
    for r in range(world):
        assert "error" not in ret[r], ret[r].get("error")
        assert ret[r]["in_sync"] and not ret[r]["timed_out"] and ret[r]["steps"] == STEPS
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    torch.set_num_threads(4)
    m = O.build("kl_ef", cfgs)
    O.load_numpy_weights(m, synth.make_weights(O.state_shapes(m), seed=1234))
    m.train()
    xg, yg = synth.make_batch(cfg["input_dims"], world * B, T, seed=17)
    x, y = torch.from_numpy(xg), torch.from_numpy(yg)
    opt = torch.optim.Adam(m.parameters())
    first = None
    for _ in range(STEPS):
        opt.zero_grad()
        terms = O.loss_terms(m, x, y, cfg)
        if first is None:
            first = {k: float(terms[k].detach()) for k in ("disc", "gen", "reg")}
        terms["loss"].backward()
        opt.step()
    # step-0 loss terms: batch MEANS average over ranks, the batch-SUM KLD adds up
    disc = np.mean([ret[r]["disc0"] for r in range(world)])
    gen = np.mean([ret[r]["gen0"] for r in range(world)])
    reg = np.sum([ret[r]["reg0"] for r in range(world)])
    assert abs(disc - first["disc"]) < TOL * abs(first["disc"])
    assert abs(gen - first["gen"]) < TOL * abs(first["gen"])
    assert abs(reg - first["reg"]) < TOL * abs(first["reg"])
    got = ret[0]["params"]
    worst, wabs = 0.0, 0.0
    for n, p in m.named_parameters():
        worst = max(worst, cases.rel_err(got[n], p.detach().numpy()))
        wabs = max(wabs, float(np.max(np.abs(got[n] - p.detach().numpy()))))
    cases.report("dp_hip_params_rel_W%d_%s" % (world, "fused" if fused else "unfused"), worst)
    cases.report("dp_hip_params_abs_W%d_%s" % (world, "fused" if fused else "unfused"), wabs)
    # Adam's normalised update moves an element whose gradient is ~0 by up to lr per step whatever the rounding
    # says: bound the trajectory by a fraction of the STEPS*lr any parameter can move
    assert wabs < 1e-4 and worst < 1e-3, (worst, wabs)      # measured 1.6e-5 / 1.7e-4


def test_bench_two_ranks_on_one_device_stay_in_sync():
    """`torchrun --nproc-per-node 2 bench.py --gpus 2` under MFM_BENCH_ONE_DEVICE=1: the driver's multi-GPU command
    line, both ranks on cuda:0, must produce one JSON line with replicas in sync."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env.update(MFM_BENCH_ONE_DEVICE="1", MFM_P2P_TIMEOUT_MS="8000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "20", "--warmup", "5", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["scaling"] == "weak"
    assert out["config"]["replicas_in_sync"] is True
    assert out["config"]["global_batch"] == 64
    assert out["config"]["collective"] in ("p2p-two-shot", "rccl")
    assert out["value"] > 0
