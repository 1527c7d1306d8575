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


def _worker(rank, world, port, fused, ret, variant="kl_ef", precision="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MFM_P2P_TIMEOUT_MS"] = "5000"
    os.environ["MFM_DP_FUSED_ADAM"] = "1" if fused else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    os.environ["MFM_SHARED_DEVICE"] = "1"       # the ranks share one GPU: no in-launch hand-overs (lstm_seq_small.hip)
    try:
        from factorized_amd import comm, engine, train
        cfgs = configs.canonical_configs(dropout=False)
        cfg = cfgs[0]
        e = engine.MFMEngine(cfgs, device="cuda:0", variant=variant, precision=precision)
        e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
        ar = comm.P2PAllReduce(world, rank, e.grads.numel())
        stepper = train.DataParallelStep(e, world, lr=1e-3, allreduce=ar, rank=rank)
        assert e.seed == 1234 + 7919 * rank
        # KLD (a batch SUM) is scaled by W inside the plan; the MMD of `MFM` is a batch statistic of the SHARD and is not
        assert e.reg_scale == (1.0 if variant == "mmd" else float(world))
        xg, yg = synth.make_batch(cfg["input_dims"], world * B, T, seed=17)
        x = torch.from_numpy(np.ascontiguousarray(xg[:, rank * B:(rank + 1) * B])).cuda()
        y = torch.from_numpy(np.ascontiguousarray(yg[rank * B:(rank + 1) * B])).cuda()
        if variant == "mmd":
            e.gauss = torch.from_numpy(_global_gauss(cfg, world)[rank * B:(rank + 1) * B].copy()).cuda()
        losses, g0 = [], None
        for s in range(STEPS):
            l = stepper.step(x, y)
            losses.append(e.loss_dict(l))
            if s == 0:
                g0 = (e.grads / world).cpu()           # the collective leaves the SUM over ranks in e.grads
        torch.cuda.synchronize()
        p = e.params.cpu()
        lo, hi = p.clone(), p.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out = {"in_sync": bool(torch.equal(lo, hi)), "timed_out": ar.timed_out(), "steps": e.step_count,
               "disc0": losses[0]["disc"], "reg0": losses[0]["reg"], "gen0": losses[0]["gen"],
               "trace": [[q["disc"], q["gen"], q["reg"]] for q in losses]}
        if rank == 0:
            out["params"] = {n: v.cpu().numpy() for n, v in e.param_views().items()}
            out["grads0"] = {n: v.numpy().copy() for n, v in e.layout.views(g0).items()}
        dist.barrier()
        ar.close()
        ret[rank] = out
    except Exception as ex:
        import traceback
        ret[rank] = {"error": "%s: %s\n%s" % (type(ex).__name__, ex, traceback.format_exc())}
    dist.destroy_process_group()


def _global_gauss(cfg, world):
    gl = cfg["zl_size"] + cfg["za_size"] + cfg["zv_size"] + cfg["zy_size"]
    return np.random.RandomState(41).normal(size=(world * B, gl)).astype(np.float32)


def _oracle_global(variant, world):
    """ONE oracle process on the concatenated global batch: per step (disc, gen, reg) and the step-0 gradients, final
    parameters.  `MFM` (variant "mmd"): the regulariser is the MEAN over the W shards of each shard's own MMD -- the
    data-parallel semantics stated in DESIGN.md section 6 (an MMD is a statistic of the batch it is computed on; the
    shard-local one is what every rank can form without exchanging latent codes)."""
    cfgs = configs.canonical_configs(dropout=False)
    cfg = cfgs[0]
    torch.set_num_threads(4)
    m = O.build(variant, cfgs)
    O.load_numpy_weights(m, synth.make_weights(O.state_shapes(m), seed=1234))
    m.train()
    xg, yg = synth.make_batch(cfg["input_dims"], world * B, T, seed=17)
    x, y = torch.from_numpy(xg), torch.from_numpy(yg)
    sizes = [cfg["zl_size"], cfg["za_size"], cfg["zv_size"], cfg["zy_size"]]
    gg = torch.from_numpy(_global_gauss(cfg, world))
    opt = torch.optim.Adam(m.parameters())
    trace, g0 = [], None
    for s in range(STEPS):
        opt.zero_grad()
        if variant == "mmd":
            loss, acc = 0.0, np.zeros(3)
            for r in range(world):
                m.mmd_gauss = list(torch.split(gg[r * B:(r + 1) * B], sizes, dim=1))
                t = O.loss_terms(m, x[:, r * B:(r + 1) * B], y[r * B:(r + 1) * B], cfg)
                loss = loss + t["loss"] / world
                acc += np.array([float(t[k].detach()) for k in ("disc", "gen", "reg")]) / world
            trace.append(list(acc))
        else:
            t = O.loss_terms(m, x, y, cfg)
            loss = t["loss"]
            trace.append([float(t[k].detach()) for k in ("disc", "gen", "reg")])
        loss.backward()
        if s == 0:
            g0 = {n: (p.grad.detach().clone().numpy() if p.grad is not None else None) for n, p in m.named_parameters()}
        opt.step()
    return m, np.array(trace), g0


def _dp_trace(ret, world, variant):
    """per step: batch MEANS average over ranks; the batch-SUM KLD adds up (the shard-local MMD of `MFM` averages)"""
    tr = np.array([ret[r]["trace"] for r in range(world)])          # [W, steps, 3]
    out = tr.mean(0)
    if variant != "mmd":
        out[:, 2] = tr[:, :, 2].sum(0)
    return out


@pytest.mark.parametrize("world,fused", [(2, True), (2, False), (4, True)])
def test_hip_data_parallel_step_equals_oracle_on_global_batch(world, fused):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mgr = mp.get_context("spawn").Manager()       # not a fork: the parent holds live GPU state (a GC in a forked child frees it there)
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


@pytest.mark.parametrize("variant,world,fused", [("kl", 2, True), ("kl", 2, False), ("mmd", 2, True), ("mmd", 4, False)])
def test_hip_data_parallel_step_mfn_variants(variant, world, fused):
    """round-2 review: the data-parallel step had only run for fp32 MFM_KL_EF.  MFM_KL: the KLD is a batch sum ->
    reg_scale = W as for MFM_KL_EF, result = one oracle process on the global batch.  MFM: the MMD regulariser is a
    statistic of the SHARD; the data-parallel objective is mean-terms + lda * mean_r MMD(shard r, gauss_r) (DESIGN.md
    section 6), which is what the oracle forms here."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mgr = mp.get_context("spawn").Manager()       # not a fork: the parent holds live GPU state (a GC in a forked child frees it there)
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fused, ret, variant, "fp32"), nprocs=world, join=True)
    for r in range(world):
        assert "error" not in ret[r], ret[r].get("error")
        assert ret[r]["in_sync"] and not ret[r]["timed_out"] and ret[r]["steps"] == STEPS
    m, trace, g0 = _oracle_global(variant, world)
    got_tr = _dp_trace(ret, world, variant)
    assert np.max(np.abs(got_tr[0] - trace[0]) / np.maximum(np.abs(trace[0]), 1e-3)) < TOL, (got_tr[0], trace[0])
    worst_g = 0.0
    for n, r in g0.items():
        g = ret[0]["grads0"][n]
        if r is None:
            assert np.all(g == 0.0), n
            continue
        worst_g = max(worst_g, cases.grad_err(g, r))
    cases.report("dp_hip_grad0_%s_W%d" % (variant, world), worst_g)
    assert worst_g < TOL, worst_g
    got = ret[0]["params"]
    worst, wabs = 0.0, 0.0
    for n, p in m.named_parameters():
        worst = max(worst, cases.rel_err(got[n], p.detach().numpy()))
        wabs = max(wabs, float(np.max(np.abs(got[n] - p.detach().numpy()))))
    cases.report("dp_hip_params_abs_%s_W%d" % (variant, world), wabs)
    # Adam normalises by sqrt(v): an element whose gradient is rounding noise moves by up to lr per step in a direction the
    # noise decides (measured 2.3e-4 on such elements of `MFM`, W = 2).  Elements with a significant step-0 gradient are
    # held to the tight bound, the rest to "cannot have moved further than Adam moves" (the criterion of DESIGN.md section 2)
    wsig = 0.0
    for n, p in m.named_parameters():
        d = np.abs(got[n] - p.detach().numpy())
        if g0[n] is not None:
            # (and above the noise floor of the atomic summation order, ~1e-8: fy_to_y_fc1.bias is ~0 by construction here --
            # cancelling L1 signs, max |g| 1.9e-9 -- and every one of its elements would otherwise count as significant)
            sig = (np.abs(g0[n]) > 1e-3 * np.abs(g0[n]).max()) & (np.abs(g0[n]) > 1e-6)
            if sig.any():
                wsig = max(wsig, float(d[sig].max()))
        assert d.max() < 1.01 * STEPS * 1e-3, n
    cases.report("dp_hip_params_abs_significant_%s_W%d" % (variant, world), wsig)
    assert wsig < 5e-5, wsig
    assert np.max(np.abs(got_tr - trace) / np.maximum(np.abs(trace), 1e-2)) < 5e-5, (got_tr[-1], trace[-1])


@pytest.mark.parametrize("variant,world,fused", [("kl_ef", 2, True), ("kl_ef", 4, False), ("kl", 2, True)])
def test_hip_data_parallel_step_bf16(variant, world, fused):
    """BASELINE config 2 is bf16 data parallel: a bf16 plan through DataParallelStep (reg_scale = W in the plan, P2P
    exchange of the fp32 gradients, grad_scale = 1/W in Adam).  Gates as for every bf16 path: averaged step-0 gradient
    within 4e-2 relative L2 of the fp32 oracle on the global batch [1.6e-2], loss curve within 2e-3, replicas bit-identical."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mgr = mp.get_context("spawn").Manager()       # not a fork: the parent holds live GPU state (a GC in a forked child frees it there)
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fused, ret, variant, "bf16"), nprocs=world, join=True)
    for r in range(world):
        assert "error" not in ret[r], ret[r].get("error")
        assert ret[r]["in_sync"] and not ret[r]["timed_out"] and ret[r]["steps"] == STEPS
    m, trace, g0 = _oracle_global(variant, world)
    got_tr = _dp_trace(ret, world, variant)
    dev_ = float(np.max(np.abs(got_tr - trace) / np.maximum(np.abs(trace), 1e-2)))
    cases.report("dp_bf16_trace_rel_%s_W%d" % (variant, world), dev_)
    assert dev_ < 2e-3, (got_tr[-1], trace[-1])
    worst = ("", 0.0)
    for n, r in g0.items():
        if r is None:
            continue
        g, r = ret[0]["grads0"][n].astype(np.float64).ravel(), r.astype(np.float64).ravel()
        nr = np.linalg.norm(r)
        if nr < 1e-9:
            continue
        rel = np.linalg.norm(g - r) / nr
        if rel > worst[1]:
            worst = (n, rel)
    cases.report("dp_bf16_grad0_relL2_%s_W%d" % (variant, world), worst[1])
    assert 1e-5 < worst[1] < 4e-2, worst          # (measured worst 1.6e-2, B = 64 global: the ill-conditioned tensors of test_gpu_bf16.py need B >= ~200)


@pytest.mark.parametrize("n,extra", [(2, []), (8, []), (2, ["--dtype", "bf16"]), (2, ["--model", "mmd"])])
def test_bench_ranks_on_one_device_stay_in_sync(n, extra):
    """`torchrun --nproc-per-node N bench.py --gpus N` under MFM_BENCH_ONE_DEVICE=1: the driver's multi-GPU command
    line (N = 2 and the full 8), all ranks on cuda:0, must produce one JSON line with replicas in sync."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env.update(MFM_BENCH_ONE_DEVICE="1", MFM_P2P_TIMEOUT_MS="20000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps",
           "20", "--warmup", "5", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 20 and out["scaling"] == "weak"
    assert out["config"]["replicas_in_sync"] is True
    assert out["config"]["global_batch"] == 32 * n
    assert out["config"]["collective"] in ("p2p-two-shot", "rccl")
    assert out["value"] > 0
    # the exchange's cost inside a step: the timed K steps minus the same K steps on local gradients (a float; on one
    # shared device it also contains the ranks' queueing behind each other, so only its presence is checked here)
    assert isinstance(out["config"]["exposed_collective_us"], float)


def _nccl_world1(q):
    """everything bench.py does with torch.distributed on the `nccl` (= RCCL) backend, on ONE rank: the 8-GPU box must
    not be the first place these lines execute."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))      # bench.py:70
        from factorized_amd import comm, engine, train
        cfgs = configs.canonical_configs(dropout=False)
        cfg = cfgs[0]
        e = engine.MFMEngine(cfgs, device="cuda:0")
        e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
        p0 = e.params.clone()
        dist.broadcast(e.params, src=0)                      # train.broadcast_params' call
        assert torch.equal(e.params, p0)
        ar = comm.TorchAllReduce()
        ok, worst = comm.validate(ar, 1, 0, e.grads.numel(), e.device)      # device tensors over RCCL
        assert ok and worst == 0.0, (ok, worst)
        assert comm._agree(True, e.device) and not comm._agree(False, e.device)
        # the fallback data-parallel step: grad_step -> torch.distributed.all_reduce -> mfm_adam_flat
        stepper = train.DataParallelStep(e, 1, lr=1e-3, allreduce=ar, rank=0)
        stepper.world = 2                                   # force the exchange branch (sum over the one rank, then / 2)
        xn, yn = synth.make_batch(cfg["input_dims"], B, T, seed=17)
        x, y = torch.from_numpy(xn).cuda(), torch.from_numpy(yn).cuda()
        l = stepper.step(x, y)
        torch.cuda.synchronize()
        assert np.isfinite(e.loss_dict(l)["loss"]) and e.step_count == 1 and not torch.equal(e.params, p0)
        tt = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)             # bench.py's max-over-ranks timing
        assert float(tt.item()) == 1.5
        dist.barrier()
        torch.cuda.synchronize()
        ar.close()
        dist.destroy_process_group()                          # ordered teardown
        q.put("ok")
    except Exception as ex:
        import traceback
        q.put("%s: %s\n%s" % (type(ex).__name__, ex, traceback.format_exc()))


def test_nccl_backend_world_size_one_smoke():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1, args=(q,))
    p.start()
    p.join(300)
    if p.is_alive():
        p.kill()
        pytest.fail("nccl world-size-1 smoke hung")
    assert not q.empty() and q.get() == "ok"
