#!/usr/bin/env python3
"""bench.py -- training images/sec of the rcn hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): synthetic MNIST-shape 28x28x1 u8 images (16 384 per rank,
seeded), the default rcn net conv(Same),pool,conv(Same),pool -> 784 -> 30 -> 10 (sigmoid, quadratic cost), N(0,1)
parameters, eta = 3.0, batch 256 PER GPU, fp32 arithmetic.  A "step" is one train_batch (rcn.rs:176-223) over one
256-image batch of the resident, already feature-extracted set -- exactly the reference's epoch-loop semantics
(features are computed once at load, rcn.rs:399-401); every step is a full forward + backward + SGD update on a
fresh batch, shuffled per epoch like rcn.rs:146.  N > 1: one process per GPU, weak scaling (global batch 256*N),
gradients combined by one RCCL all-reduce of the flat 23 860-element buffer per step (mercer_research_amd/dp.py).

Rank 0 prints ONE JSON line.  Extra objects on it: "roofline" (dominant kernel, HIP events) and "cpu_baseline"
(the oracle's restatement of rcn's rayon loop timed on this host's cores -- a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 256
N_IMAGES = 16384
ETA = 3.0
DIMS = [784, 30, 10]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(seconds_budget: float = 12.0):
    """rcn's CPU path (oracle/rcn_oracle.c restated from rcn.rs:176-314, threaded like the rayon loop) on this host:
    B = 32 (BASELINE.json configs[0]), all host cores, bounded sample."""
    import tempfile
    from oracle.rcn_oracle import COracle, DEFAULT_LAYERS, build_oracle, one_hot, synthetic_images, synthetic_params
    try:
        path = build_oracle(native=True, out_dir=tempfile.mkdtemp(prefix="rcn_oracle_"))
    except Exception:
        path = build_oracle()
    o = COracle(path)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = 2048
    imgs, labels = synthetic_images(n)
    feats = o.features(imgs[:n], DEFAULT_LAYERS)
    m, s = o.gen_scales(feats)
    X, Y = o.standardize(feats, m, s), one_hot(labels[:n])
    ws, bs = synthetic_params(DIMS, seed=42)
    holder = o.net(ws, bs)
    B = 32
    o.train_steps_inplace(holder, X, Y, B, 8, ETA, threads=cores)          # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:
        o.train_steps_inplace(holder, X, Y, B, 16, ETA, threads=cores)
        steps += 16
        el = time.perf_counter() - t0
        if el >= seconds_budget or steps >= 20000:
            break
    multi = steps * B / el
    holder1 = o.net(ws, bs)
    t0 = time.perf_counter()
    s1 = 0
    while time.perf_counter() - t0 < min(4.0, seconds_budget / 3):
        o.train_steps_inplace(holder1, X, Y, B, 8, ETA, threads=0)
        s1 += 8
    single = s1 * B / (time.perf_counter() - t0)
    return {"value": round(max(multi, single), 1), "unit": "images/s", "cores": cores if multi >= single else 1, "kind": "port",
            "sample": f"{steps} train_batch steps of B={B} (784-30-10, f64) over {n} synthetic feature vectors, "
                      f"{cores} threads: {multi:.0f} img/s; 1 thread: {single:.0f} img/s",
            "threads_all_cores_images_per_s": round(multi, 1), "single_thread_images_per_s": round(single, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 sample-tile kernels, 2 feature-sliced pipeline")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = max(args.gpus, world) if world > 1 else args.gpus
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")

    import mercer_research_amd as amd
    from mercer_research_amd.device import DeviceRCN
    from mercer_research_amd.dp import DataParallelStep
    from oracle.rcn_oracle import synthetic_images, synthetic_params   # data generator only (shared with the tests)

    dtype = amd.F64 if args.dtype == "f64" else amd.F32
    d = DeviceRCN(classes=10, feedforward_cfg=[30], input_shape=(28, 28), dtype=dtype, device=local_rank)
    imgs, labels = synthetic_images(N_IMAGES, seed=1234 + rank)          # every rank owns a different shard of data
    ws, bs = synthetic_params(DIMS, seed=42)                             # identical replicas (rcn.rs:500-523 shapes)
    d.set_params(ws, bs)
    d.set_dense_path(args.path)
    with torch.cuda.stream(d.stream):
        imgs_d = torch.from_numpy(imgs).to(d.device)
        labels_d = torch.from_numpy(labels).to(d.device)
    X, Y = d.load_data(imgs_d, labels_d)                                 # HIP features + gen_scales + standardise (load time)
    nb_epoch = N_IMAGES // B_PER_GPU
    B = B_PER_GPU
    step_no = [0]
    # training_set.shuffle (rcn.rs:146): the permutation of epoch e+1 is drawn on a side stream while epoch e trains
    # (two index buffers, events in both directions), so the epoch loop on the main stream is graph launch after graph
    # launch.  It is still drawn once per epoch, inside the timed region.
    perms = [torch.empty(N_IMAGES, dtype=torch.int32, device=d.device) for _ in range(2)]
    shuf_stream = torch.cuda.Stream(device=d.device)
    ready = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    epoch_no = [0]

    def draw(buf: int):
        with torch.cuda.stream(shuf_stream):
            shuf_stream.wait_event(consumed[buf])                        # the epoch that last read this buffer has finished
            perms[buf].copy_(torch.randperm(N_IMAGES, device=d.device))
            ready[buf].record(shuf_stream)

    if world == 1:
        for b_ in range(2):
            consumed[b_].record(d.stream)
        draw(0)

        def run(k: int):
            """k consecutive train_batch steps; every 64 steps (one pass over the set) a new permutation."""
            done = 0
            while done < k:
                pos = step_no[0] % nb_epoch
                buf = epoch_no[0] % 2
                if pos == 0:
                    d.stream.wait_event(ready[buf])
                    draw(1 - buf)                                        # next epoch's permutation, off the critical path
                take = min(k - done, nb_epoch - pos)
                d.train_epoch(X, Y, perms[buf][pos * B:], B, take, ETA, None)
                step_no[0] += take
                done += take
                if step_no[0] % nb_epoch == 0:
                    consumed[buf].record(d.stream)
                    epoch_no[0] += 1
    else:
        dp = DataParallelStep(d)
        dp.broadcast_params(0)
        xb = d.empty(B, d.F)
        yb = d.empty(B, d.classes)

        perm = perms[0]

        def run(k: int):
            with torch.cuda.stream(d.stream):
                for _ in range(k):
                    pos = step_no[0] % nb_epoch
                    if pos == 0:
                        perm.copy_(torch.randperm(N_IMAGES, device=d.device))
                    sel = perm[pos * B:(pos + 1) * B].long()
                    torch.index_select(X, 0, sel, out=xb)
                    torch.index_select(Y, 0, sel, out=yb)
                    dp.train_batch(xb, yb, ETA, B * world)
                    step_no[0] += 1

    def sync():
        d.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(d.stream)
    run(args.steps)
    ev1.record(d.stream)
    sync()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=d.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_t = d.empty(1)
    d.train_batch(X[:B], Y[:B], 0.0, loss_t)                             # eta = 0: reads the current cost, changes nothing
    d.synchronize()
    final_loss = float(loss_t.item())

    images = args.steps * B * world
    result = {
        "metric": "training images/sec, MNIST-shape 28x28x1 batch=256, at 1/2/4/8 MI355X",
        "value": round(images / elapsed, 1), "unit": "images/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 6), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "MNIST-shape 28x28x1, rcn default net conv(Same)-pool-conv(Same)-pool -> 784-30-10 sigmoid/MSE, "
                               "train_batch B=256 per GPU over 16384 resident pre-extracted feature vectors per GPU, eta=3.0",
                   "global_batch": B * world, "parallelism": f"dp{world}", "images_per_rank": N_IMAGES,
                   "device_ms_per_step_rank0": round(dev_ms / args.steps, 6), "final_cost_rank0": final_loss},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events on the stream the kernels run on) ----
        us_first, us_second = d.time_kernels(X[:B], Y[:B], reps=400)
        es = 8 if args.dtype == "f64" else 4
        P, F, H, C = d.P, DIMS[0], DIMS[1], DIMS[2]
        G = (F + 15) // 16
        # algorithmic bytes per launch (DESIGN.md "Kernels"), feature-sliced path:
        #   k_pipe_a: both batches' feature rows once (finish step i-1, start step i), delta_1/delta_2/a_1 of step i-1,
        #             every parameter read + written once.  (Its slab write is an implementation artefact, excluded.)
        #   k_pipe_b: the targets, tail parameters + b_0; writes a_1, delta_1, delta_2.  (Slab read excluded likewise.)
        bytes_a = (2 * B * F + B * (2 * H + C) + 2 * P) * es
        bytes_b = (B * C + (P - F * H) + B * (2 * H + C)) * es
        names = ("k_pipe_b", "k_pipe_a") if args.path != 1 else ("k_dense_fwd", "k_dense_wgrad")
        if args.path == 1:
            sumd = H + C
            bytes_b = (B * F + B * C + P + B * H + B * sumd) * es          # k_dense_fwd
            bytes_a = (B * F + B * sumd + B * H + 2 * P) * es              # k_dense_wgrad
        dom, us, by = (names[0], us_first, bytes_b) if us_first >= us_second else (names[1], us_second, bytes_a)
        flops_step = 2 * B * ((F * H + H * C) * 2 + H * C)
        result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(by / us / 1e3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(by / us / 1e3 / HBM_PEAK_GBS, 5), "traffic": None,
                              "algorithmic_bytes_per_launch": by, "us_per_launch_hip_events": round(us, 3),
                              "us_" + names[0]: round(us_first, 3), "us_" + names[1]: round(us_second, 3),
                              "step_gflops_per_s": round(flops_step / (elapsed / args.steps) / 1e9, 1),
                              "note": "one train_batch at B=256 moves ~1 MB and ~25 MFLOP: launch/latency-bound, far from either roof"}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
            result["config"]["gpu_over_cpu"] = round(result["value"] / max(result["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
