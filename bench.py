#!/usr/bin/env python3
"""bench.py -- training samples/sec of the MFM_KL_EF step (MOSI shape, T=20, 3 modalities).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full training step of the reference's hot loop (mfm_mosi.py:424-442): batch
fetch from the HBM-resident synthetic split + forward + joint loss + backward + (all-reduce) +
Adam, train mode with the canonical dropouts.  Workload (BASELINE.json configs[1]): canonical
MOSI sizes (mfm_mosi.py:1239-1286), B=32 per GPU, T=20 (configs/mosi.json seqlength), D=325,
fp32.  N>1 is weak scaling: every rank processes its own B=32 shard.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, timed inside the timed region by
two HIP events that carry the dispatch's own begin / end timestamps (hipExtLaunchKernelGGL on the
launch stream: what rocprofv3 --kernel-trace reports, no bracket overhead to subtract);
`cpu_baseline` times the CPU oracle (a restatement of the reference's PyTorch-CPU path) on this host
for a bounded number of steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (2:1-sparsity figures are never used)
HBM_PEAK_GBS = 8000.0
MODEL_NAME = {"kl_ef": "MFM_KL_EF", "kl": "MFM_KL (MFN encoder, KLD)", "mmd": "MFM (MFN encoder, MMD)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (reference batchsize=32)")
    ap.add_argument("--seq", type=int, default=0, help="sequence length; 0 = read configs/mosi.json")
    ap.add_argument("--shape", default="mosi", choices=["mosi", "you", "mosei"])
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = the reference's arithmetic (BASELINE config 1, the headline); bf16 = bf16 MFMA operands, "
                         "fp32 accumulate / master weights / cell state / loss (BASELINE configs 2-4)")
    ap.add_argument("--model", default="kl_ef", choices=["kl_ef", "kl", "mmd"],
                    help="kl_ef = MFM_KL_EF (the headline workload); kl / mmd = MFM_KL / MFM with the Memory Fusion Network "
                         "encoder, the classes train_mfm picks by config['type'] (reference mfm_mosi.py:398-401)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel time table to stderr")
    args = ap.parse_args()

    import torch
    from factorized_amd import configs as C
    from factorized_amd import comm, engine, synth, train

    rank, local_rank, world = train.dp_env()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # MFM_BENCH_ONE_DEVICE=1 (testing only): every rank uses cuda:0 and the collectives run on gloo, so the
    # multi-process path can be exercised on a 1-GPU box; the number it prints is not a scaling result.
    one_dev = os.environ.get("MFM_BENCH_ONE_DEVICE") == "1"
    if one_dev and world > 1:
        # several ranks drive ONE GPU at the same time: the in-launch hand-overs of the role workgroups need the device to
        # themselves (lstm_seq_small.hip, shared_device()); the separate launches are used instead
        os.environ["MFM_SHARED_DEVICE"] = "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg_fn = {"mosi": C.canonical_configs, "you": C.you_configs, "mosei": C.mosei_configs}[args.shape]
    cfgs = cfg_fn(dropout=True)
    cfg = cfgs[0]
    if args.seq:
        T = args.seq
    else:
        _, T = C.load_json_config(os.path.join(ROOT, "configs", {"mosi": "mosi.json", "you": "you.json",
                                                                 "mosei": "mosi.json"}[args.shape]))
    B = args.batch

    e = engine.MFMEngine(cfgs, device="cuda:%d" % local_rank, precision=args.dtype, variant=args.model)
    e.load_weights(synth.make_weights(e.layout.shapes, seed=1234))
    train.broadcast_params(e, world)
    n_samples = max(1280, B * world * 8)          # MOSI-scale split (1,284 train utterances)
    data = train.DeviceDataset(cfg, n_samples, T, B, e.device, seed=11)
    my_batches = train.shard_batches(data.nb, rank, world)
    # the gradient collective: the P2P kernel if it sets up and validates on this job's devices, else RCCL
    allreduce = comm.make_allreduce(world, rank, e.grads.numel(), e.device) if world > 1 else None
    stepper = train.DataParallelStep(e, world, lr=1e-3, allreduce=allreduce, rank=rank)

    def run(n, first=0):
        for i in range(n):
            x, y = data.batch(my_batches[(first + i) % len(my_batches)])
            stepper.step(x, y)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed) + a per-kernel breakdown pass (untimed) to find the dominant kernel
    dbg = os.environ.get("MFM_BENCH_DEBUG")
    if dbg:
        sys.stderr.write("setup done, warmup...\n"); sys.stderr.flush()
    run(args.warmup)
    barrier()
    if dbg:
        sys.stderr.write("warmup done\n"); sys.stderr.flush()
    e.set_timing(T, B, (1 << 30) - 1)
    run(10, args.warmup)
    barrier()
    table = e.collect_timing(T, B)
    dom = max(table, key=lambda k: table[k]["ms"])
    if args.breakdown and rank == 0:
        tot = sum(v["ms"] for v in table.values()) / 10
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            if v["count"]:
                sys.stderr.write("%-14s %8.2f us/launch  %5.1f%%\n" % (k, 1e3 * v["ms"] / v["count"],
                                                                      100 * v["ms"] / 10 / tot))
    # inside the timed region the dominant kernel is timed on every 8th step only (its two events are extra work for the
    # runtime on the steps they are active on)
    TIMING_EVERY = max(1, min(8, args.steps // 4))
    e.set_timing(T, B, 1 << table[dom]["kid"], every=TIMING_EVERY)

    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides, max over ranks
    rank_dt = [0.0]

    def timed_region():
        barrier()
        if world > 1 and hasattr(allreduce, "wait_stats"):
            allreduce.wait_stats(reset=True)
        t0 = time.perf_counter()
        run(args.steps, args.warmup + 10)
        torch.cuda.synchronize()
        rank_dt[0] = time.perf_counter() - t0          # this rank's own K steps (before the closing barrier)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    dt = timed_region()
    # a hand-over inside a launch that gave up (something else ran on this GPU) means steps of the timed region were skipped by
    # the optimizer: not a valid measurement.  The engine is on separate launches now (check_status); time the K steps again.
    # All ranks take the same branch (a raised guard reaches every rank through the exchange, but the status word is local).
    ho_failed = e.check_status(raise_on_error=False) != 0
    if world > 1:
        import torch.distributed as dist
        flag = torch.tensor([1.0 if ho_failed else 0.0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        ho_failed = bool(flag.item() > 0)
        if ho_failed:
            e.set_handover(False)
    if ho_failed:
        if rank == 0:
            sys.stderr.write("[bench] an in-launch hand-over timed out during the run: repeating the timed region on separate launches\n")
        if world > 1:
            train.broadcast_params(e, world)
        e.collect_timing(T, B)
        dt = timed_region()
    in_sync = None
    flag_wait = allreduce.wait_stats(reset=True) if (world > 1 and hasattr(allreduce, "wait_stats")) else None
    if world > 1:
        import torch.distributed as dist
        # a P2P wait that gave up on any rank invalidates the measurement: switch every rank to RCCL,
        # re-align the replicas and time the K steps again
        if not comm._agree(not allreduce.timed_out(), e.device):
            if rank == 0:
                sys.stderr.write("[bench] p2p all-reduce timed out during the run: repeating the timed region on RCCL\n")
            allreduce.close()
            allreduce = stepper.allreduce = comm.TorchAllReduce()
            train.broadcast_params(e, world)
            e.collect_timing(T, B)
            dt = timed_region()
            flag_wait = None
        # replicas must still hold identical parameters (same reduced gradients on every rank)
        lo, hi = e.params.clone(), e.params.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool((lo == hi).all().item())
        # cost of the exchange alone (outside the timed region): the collective in use next to
        # torch.distributed's, 50 back-to-back calls on the gradient buffer each
        coll_us, coll_rank = {}, {}
        entries = [(allreduce.name, allreduce)]
        if allreduce.name != "rccl":
            entries.append(("rccl", comm.TorchAllReduce()))
        for name, fn in entries:
            for _ in range(5):
                fn(e.grads)
            barrier()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(50):
                fn(e.grads)
            ev1.record()
            torch.cuda.synchronize()
            coll_rank[name] = round(1e3 * ev0.elapsed_time(ev1) / 50, 1)
            tt = torch.tensor([coll_rank[name]], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            coll_us[name] = round(float(tt.item()), 1)
    exposed_us = None
    if world > 1:
        # what the exchange ADDS to a step: the same K steps without it (fwd + bwd + Adam on the local gradients), timed the
        # same way.  Last: the replicas diverge from here on.
        def run_local(n, first=0):
            for i in range(n):
                x, y = data.batch(my_batches[(first + i) % len(my_batches)])
                e.grad_step(x, y, check=False)
                e.adam(lr=1e-3, grad_scale=1.0 / world)
        run_local(5)
        barrier()
        t0 = time.perf_counter()
        run_local(args.steps, args.warmup + 10)
        torch.cuda.synchronize()
        local_dt = time.perf_counter() - t0
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        exposed_us = round(1e6 * (dt - float(tt.item())) / args.steps, 1)
        # forward + backward alone (no optimizer, no exchange), this rank
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            x, y = data.batch(my_batches[(args.warmup + 10 + i) % len(my_batches)])
            e.grad_step(x, y, check=False)
        torch.cuda.synchronize()
        grad_dt = time.perf_counter() - t0
        # one line per rank, so that a sub-par scaling result can be read from this run alone: which rank is slow, whether the
        # time goes into its own step, into waiting for a late peer (flag round 1) or into the exchange itself (round 2)
        mine = {"rank": rank, "step_us": round(1e6 * rank_dt[0] / args.steps, 1),
                "grad_step_us": round(1e6 * grad_dt / args.steps, 1),
                "local_step_us": round(1e6 * local_dt / args.steps, 1),
                "exposed_collective_us": round(1e6 * (rank_dt[0] - local_dt) / args.steps, 1),
                "collective_us_per_call": coll_rank, "flag_wait_us": flag_wait,
                "preflight": stepper.preflight_report, "role_workgroups": bool(e.handover),
                "handover_failures": e.handover_failures}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    timed = e.collect_timing(T, B)[dom]
    e.set_timing(T, B, 0)

    if rank == 0:
        ms = 1e3 * dt / args.steps
        value = B * world * args.steps / dt
        # the dominant kernel's own dispatch timestamps (csrc/common.h, MFM_LAUNCH_TIMED): nothing to subtract.  Round 5
        # subtracted the cost of an empty event bracket from a bracket around the launch and landed below rocprofv3's fastest
        # sample; the bracket's cost is still reported (empty_bracket_us) for kernels that share a timer with other launches
        k_ms = timed["ms"] / max(timed["count"], 1)
        ev_ms = e.bracket_overhead_ms()
        k_flops = timed["flops"]
        achieved = (k_flops / (k_ms * 1e-3)) / 1e12 if k_ms > 0 and k_flops > 0 else 0.0
        work = e.work_per_step(T, B)
        peak = FP32_MATRIX_PEAK_TFLOPS if args.dtype == "fp32" else BF16_MATRIX_PEAK_TFLOPS
        # HBM bytes per launch of the dominant kernel: measured in a SEPARATE rocprofv3 --pmc pass of this
        # command (scripts/profile_round6.sh -> profiles/r06_traffic_B*.json); only valid for the workload it
        # was measured on, otherwise null.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r06_traffic_B%d%s.json" % (B, "" if args.dtype == "fp32" else "_bf16"))
        if os.path.exists(tpath) and args.shape == "mosi" and T == 20 and args.model == "kl_ef":
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("total_bytes")
            except Exception:
                traffic = None
        out = {
            "metric": "training samples/sec (MOSI-shape, T=%d, 3 modalities)" % T,
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": MODEL_NAME[args.model] + " %s canonical sizes, per-GPU B=%d, T=%d, D=%d, train mode "
                                   "(fwd + joint loss + bwd + Adam), %s HIP" % (args.shape, B, T, sum(cfg["input_dims"]),
                                                                            "fp32" if args.dtype == "fp32" else
                                                                            "bf16-operand / fp32-accumulate"),
                       "global_batch": B * world, "seq_len": T, "parallelism": "dp%d" % world,
                       "params": e.layout.numel,
                       "handover_failures": e.handover_failures, "role_workgroups": bool(e.handover),
                       "collective": allreduce.name if allreduce is not None else None,
                       "replicas_in_sync": in_sync,
                       "collective_us_per_call": coll_us if world > 1 else None,
                       "exposed_collective_us": exposed_us,
                       "per_rank": per_rank if world > 1 else None},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 4),
                         "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 6), "traffic": traffic,
                         "note": ("fp32 matrix peak == fp32 vector peak on gfx950 (157.3 TF); at B<=512 the "
                                  "recurrent products run on the VALU small-tile kernels, above on the MFMA")
                         if args.dtype == "fp32" else
                         ("dense bf16 MFMA peak; GEMMs and, from B=128, the recurrences feed v_mfma_f32_16x16x32_bf16 (below: fp32 one-row "
                          "VALU recurrences; decoder fc1 up to 5120 rows: bf16-rounded operands on the fp32 MFMA)"),
                         "kernel_us": round(1e3 * k_ms, 2), "kernel_time_source": "dispatch begin/end timestamps (hipExtLaunchKernelGGL events)",
                         "empty_bracket_us": round(1e3 * ev_ms, 2), "kernel_launches_timed": timed["count"],
                         "timed_every_nth_step": TIMING_EVERY, "kernel_flops": k_flops,
                         "step_flops": work["flops"], "step_bytes": work["bytes"],
                         "step_tflops": round(work["flops"] / (ms * 1e-3) / 1e12, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import mfm_oracle as O
            import torch as _t
            cores = os.cpu_count() or 1
            # the reference step is ~165 tiny ops: it stops scaling at a handful of threads, so time
            # it at 1 thread and at min(cores, 8) threads (bounded ~8 s each) and report the faster.
            runs = [O.time_cpu_steps(cfgs, B, T, budget_s=8.0, threads=1, variant=args.model)]
            if cores > 1:
                runs.append(O.time_cpu_steps(cfgs, B, T, budget_s=8.0, threads=min(cores, 8), variant=args.model))
            if cores > 8 and os.environ.get("MFM_BENCH_ALL_CORES") == "1":
                # SURVEY.md section 8d asks for the all-physical-cores figure once: it is slower than 1 thread for this
                # ~165-tiny-op step (recorded in profiles/), so the default run does not spend 8 s on it
                runs.append(O.time_cpu_steps(cfgs, B, T, budget_s=8.0, threads=cores // 2, variant=args.model))
            r = max(runs, key=lambda q: q["samples_per_s"])
            cpu_model = "?"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            out["cpu_baseline"] = {"value": round(r["samples_per_s"], 1), "unit": "samples/s", "cores": r["threads"],
                                   "kind": "port", "ms_per_step": round(r["ms_per_step"], 2),
                                   "host_cores": cores, "host_cpu": cpu_model,
                                   "all_runs": [{"threads": q["threads"], "samples_per_s": round(q["samples_per_s"], 1),
                                                 "steps": q["steps"]} for q in runs],
                                   "sample": "%d training steps (~8 s) of the torch-CPU oracle (restated reference "
                                             "path: per-timestep nn.LSTMCell loop, joint loss, optim.Adam), B=%d T=%d, "
                                             "%d threads" % (r["steps"], B, T, r["threads"])}
            out["speedup_vs_cpu"] = round(value / r["samples_per_s"], 1)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        # ordered teardown: no rank unmaps or frees its staging block while a peer could still touch it
        barrier()
        if allreduce is not None:
            allreduce.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
