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
    ap.add_argument("--no-e2e", action="store_true", help="skip the secondary end-to-end (u8 image -> features -> step) measurement")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 sample-tile kernels, 2 feature-sliced pipeline")
    ap.add_argument("--dp-impl", choices=["native", "torch"], default="native",
                    help="data-parallel loop: native = RCCL calls inside librcn_hip (rcn_hip_dp_*), torch = torch.distributed all_reduce per step")
    ap.add_argument("--dp", action="store_true", help="use the data-parallel step (gradient -> all-reduce -> apply) even at world size 1")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dp = world > 1 or args.dp
    real_stdout = None
    if use_dp:
        # RCCL prints a version banner on STDOUT when a communicator is created; rank 0's stdout must carry exactly one JSON
        # line, so everything written to fd 1 during the run goes to stderr and the line is written to the saved descriptor
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = max(args.gpus, world) if world > 1 else args.gpus
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")

    import mercer_research_amd as amd
    from mercer_research_amd.device import DeviceRCN
    from mercer_research_amd.dp import DataParallelStep
    from mercer_research_amd.synth import synthetic_images, synthetic_params

    dtype = amd.F64 if args.dtype == "f64" else amd.F32
    d = DeviceRCN(classes=10, feedforward_cfg=[30], input_shape=(28, 28), dtype=dtype, device=local_rank)
    imgs, labels = synthetic_images(N_IMAGES, seed=1234 + rank)          # every rank owns a different shard of data
    ws, bs = synthetic_params(DIMS, seed=42)                             # identical replicas (rcn.rs:500-523 shapes)
    d.set_params(ws, bs)
    # data-parallel: with the peer-read exchange the step runs on the feature-sliced pipeline (k_p2_b, k_p2_dp_grad,
    # k_p2_dp_apply); on the RCCL fallback the gradient-out form runs on the sample-tile kernels (--path 1 forces those)
    d.set_dense_path(args.path)
    with torch.cuda.stream(d.stream):
        imgs_d = torch.from_numpy(imgs).to(d.device)
        labels_d = torch.from_numpy(labels).to(d.device)
    X, Y = d.load_data(imgs_d, labels_d)                                 # HIP features + gen_scales + standardise (load time)
    nb_epoch = N_IMAGES // B_PER_GPU
    B = B_PER_GPU
    step_no = [0]
    # training_set.shuffle (rcn.rs:146): one fresh permutation per pass over the set, drawn ON the device by
    # rcn_hip_shuffle_dev in-stream (one small kernel per chunk of passes -- a side-stream torch.randperm serialises
    # against graph replays on this stack and cost 1.8 us/step), inside the timed region.  One hipGraph replay covers
    # EPG passes; every step sees a fresh batch of a fresh shuffle.
    EPG = 8                                                          # passes (epochs) per graph replay
    chunk_steps = EPG * nb_epoch
    perm = torch.empty(EPG * N_IMAGES, dtype=torch.int32, device=d.device)
    chunk_no = [0]

    if not use_dp:
        def plan(k: int):
            """chunk sizes (in steps) that run(k) will issue"""
            return [min(chunk_steps, k - i) for i in range(0, k, chunk_steps)]

        def prime(k: int):
            """instantiate the graphs run(k) will replay (set-up, untimed): one per chunk length"""
            for take in set(plan(k)):
                d.prepare_epoch(X, Y, perm, B, take, ETA, None)

        def run(k: int):
            """k consecutive train_batch steps in chunks of up to EPG passes; a new permutation every pass."""
            for take in plan(k):
                d.shuffle(perm, N_IMAGES, EPG, seed=0x5DEECE66D + rank * 7919 + chunk_no[0])
                d.train_epoch(X, Y, perm, B, take, ETA, None)
                step_no[0] += take
                chunk_no[0] += 1
    else:
        # one process per GPU: every rank shuffles its own resident shard of the data, takes 256 rows per step, and the
        # summed shard gradients meet in ONE all-reduce of the flat parameter-shaped buffer (RCCL over xGMI)
        native = args.dp_impl == "native"
        if native:
            try:
                d.dp_init()                   # RCCL communicator owned by the library; torch only carried its 128-byte id
                ok = 1
            except Exception as e:            # e.g. librccl not loadable: every rank falls back together
                print(f"[bench] native RCCL loop unavailable on rank {rank}: {e}", file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=d.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(flag.item())
            args.dp_impl = "native" if native else "torch"
        if native:
            d.dp_broadcast_params(0)
            allreduce_kind = {0: "ncclAllReduce (RCCL)", 1: "xgmi peer reads between kernels, fused with the update (csrc/dense_p2_dp.hpp, 3 kernels/step)",
                              2: "xgmi peer reads inside the gradient kernel (csrc/dense_p2_dp.hpp, 2 kernels/step)"}[d.dp_p2p_mode()]

            def prime(k: int):
                """instantiate the graphs run(k) will replay from the current position (set-up, untimed, not collective)"""
                pos, done, seen = step_no[0] % nb_epoch, 0, set()
                while done < k:
                    take = min(nb_epoch - pos, k - done)
                    if (pos, take) not in seen:
                        seen.add((pos, take))
                        d.dp_prepare_epoch(X, Y, perm[pos * B:], B, take, ETA, None)
                    done += take
                    pos = (pos + take) % nb_epoch

            def run(k: int):
                # the whole loop is native: per step gradient kernels -> ncclAllReduce -> update, enqueued by
                # rcn_hip_dp_train_epoch_dev; Python only starts each pass (one in-stream shuffle + one call per 64 steps)
                done = 0
                while done < k:
                    pos = step_no[0] % nb_epoch
                    if pos == 0:
                        d.shuffle(perm, N_IMAGES, 1, seed=0x5DEECE66D + rank * 7919 + chunk_no[0])
                        chunk_no[0] += 1
                    take = min(nb_epoch - pos, k - done)
                    d.dp_train_epoch(X, Y, perm[pos * B:], B, take, ETA, None)
                    step_no[0] += take
                    done += take
        else:
            allreduce_kind = "torch.distributed all_reduce (RCCL)"
            dp = DataParallelStep(d)
            dp.broadcast_params(0)

            def run(k: int):
                with torch.cuda.stream(d.stream):
                    for _ in range(k):
                        pos = step_no[0] % nb_epoch
                        if pos == 0:
                            d.shuffle(perm, N_IMAGES, 1, seed=0x5DEECE66D + rank * 7919 + chunk_no[0])
                            chunk_no[0] += 1
                        dp.train_batch(X, Y, ETA, B * world, perm=perm[pos * B:(pos + 1) * B])
                        step_no[0] += 1

    def sync():
        d.synchronize()
        torch.cuda.synchronize()
        if use_dp:
            dist.barrier()
            torch.cuda.synchronize()

    can_prime = (not use_dp) or (use_dp and args.dp_impl == "native")
    if can_prime:
        prime(args.warmup)
    run(args.warmup)
    if can_prime:
        # twice: a call shape that needs a larger workspace moves it, which drops the graphs captured before it (they point into
        # the old one); the second pass re-instantiates those, so that nothing is captured inside the timed region
        prime(args.steps)
        prime(args.steps)
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(d.stream)
    run(args.steps)
    ev1.record(d.stream)
    sync()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if use_dp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=d.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_t = d.empty(1)
    d.train_batch(X[:B], Y[:B], 0.0, loss_t)                             # eta = 0: reads the current cost, changes nothing
    d.synchronize()
    final_loss = float(loss_t.item())

    # SURVEY.md §8(d) asks for two numbers: train-only (the headline `value`: features resident, the reference's epoch-loop
    # semantics) and END-TO-END: u8 image -> feature kernel (+ standardise, fused) -> train_batch, i.e. the features of
    # every pass are recomputed from the resident u8 images inside the timed region.
    e2e = None
    if not use_dp and not args.no_e2e:
        perm1 = perm[:N_IMAGES]
        try:
            d.train_epoch_images(imgs_d, Y, perm1, B, nb_epoch, ETA, None, prepare_only=True)
            fused_e2e = True
        except amd.RcnPanic:                  # --path 1 (sample-tile kernels): no packed epoch image to fuse into
            fused_e2e = False
            Xe = d.empty(N_IMAGES, DIMS[0])
            d.prepare_epoch(Xe, Y, perm1, B, nb_epoch, ETA, None)

        def e2e_pass(i):
            d.shuffle(perm1, N_IMAGES, 1, seed=0xE2E + i)
            if fused_e2e:
                # one kernel per segment of the pass: flatten_feature_set + standardise + gather into the training layout, straight
                # from the u8 pictures (rcn_hip_train_epoch_images_dev); no feature matrix is written
                d.train_epoch_images(imgs_d, Y, perm1, B, nb_epoch, ETA, None)
            else:
                d.features(imgs_d, True, Xe)
                d.train_epoch(Xe, Y, perm1, B, nb_epoch, ETA, None)
        for i in range(4):
            e2e_pass(i)
        d.synchronize()
        reps = 32
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record(d.stream)
        for i in range(reps):
            e2e_pass(100 + i)
        eb.record(d.stream)
        d.synchronize()
        e2e = reps * N_IMAGES / (ea.elapsed_time(eb) * 1e-3)

    # data-parallel runs: every rank must hold bit-identical parameters (identical update from rank-ordered sums); a stale or
    # torn read in the exchange would show up here as diverged replicas
    replicas_identical = None
    if use_dp:
        with torch.cuda.stream(d.stream):
            pf = d.params_flat().to(torch.float64)
            chk = torch.stack([pf.sum(), (pf * pf).sum(), pf.abs().max()])
        d.synchronize()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi)) and bool(torch.isfinite(chk).all())

    images = args.steps * B * world
    result = {
        "metric": "training images/sec, MNIST-shape 28x28x1 batch=256, at 1/2/4/8 MI355X",
        "value": round(images / elapsed, 1), "unit": "images/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 6), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "MNIST-shape 28x28x1, rcn default net conv(Same)-pool-conv(Same)-pool -> 784-30-10 sigmoid/MSE, "
                               "train_batch B=256 per GPU over 16384 resident pre-extracted feature vectors per GPU, eta=3.0",
                   "global_batch": B * world, "parallelism": f"dp{world}", "step_form": f"gradient -> all-reduce -> apply ({args.dp_impl} loop)" if use_dp else "fused update", "images_per_rank": N_IMAGES,
                   "device_ms_per_step_rank0": round(dev_ms / args.steps, 6), "final_cost_rank0": final_loss,
                   "end_to_end_images_per_s": round(e2e, 1) if e2e else None,
                   "allreduce": allreduce_kind if use_dp else None, "replicas_identical": replicas_identical},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events on the stream the kernels run on) ----
        # HIP events on the kernels' own stream: each kernel back to back with itself, and the alternating pair as the epoch
        # loop issues it (walking the last epoch's packed batches).  A kernel's duration inside the real loop is the pair
        # time split in the ratio of the two stand-alone times; that is what a profiler's per-dispatch average shows.
        us_first, us_second, us_pair = d.time_kernels(X[:B], Y[:B], reps=504)
        es = 8 if args.dtype == "f64" else 4
        P, F, H, C = d.P, DIMS[0], DIMS[1], DIMS[2]
        # Algorithmic bytes per launch = SURVEY.md §8(d)'s per-image figure x the B images one launch processes
        # (DESIGN.md "Kernels"): features F*es + one-hot C*es + every parameter read and written once per step (2*P*es/B).
        # The feature rows and the parameters belong to the kernel that owns W_0 (k_p2_a / k_dense_fwd+wgrad); the targets to
        # the tail kernel.  The second pass over the features, the partial-sum slab and the activation/delta exchange are
        # implementation traffic: they show up in `traffic` (PMC), not here.
        per_img_main = F * es + 2 * P * es / B
        per_img_tail = C * es
        if args.path == 1:
            names, by = ("k_dense_fwd", "k_dense_wgrad"), (B * (F * es + C * es + P * es / B), B * (P * es / B))
        else:
            names, by = ("k_p2_b", "k_p2_a"), (B * per_img_tail, B * per_img_main)
        k = 0 if us_first >= us_second else 1
        if args.path != 1:
            k = 1                                   # k_p2_a owns the features and W_0; k_p2_b is the small tail kernel
        us = us_pair * (us_first, us_second)[k] / (us_first + us_second)
        flops_step = 2 * B * ((F * H + H * C) * 2 + H * C)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_summary.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(names[k], {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {"bound": "hbm", "kernel": names[k], "achieved": round(by[k] / us / 1e3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(by[k] / us / 1e3 / HBM_PEAK_GBS, 5), "traffic": traffic,
                              "traffic_source": "profiles/r1_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" if traffic else None,
                              "algorithmic_bytes_per_launch": round(by[k]), "us_per_launch_hip_events": round(us, 3),
                              "us_alternating_pair": round(us_pair, 3),
                              "us_standalone_" + names[0]: round(us_first, 3), "us_standalone_" + names[1]: round(us_second, 3),
                              "step_gflops_per_s": round(flops_step / (elapsed / args.steps) / 1e9, 1),
                              "note": "one train_batch at B=256 is ~1 MB and ~25 MFLOP: bound by launch + dependent-latency floors "
                                      "(1.6 us per dependent launch, >=1 us per global round trip), far from either roof"}
        if not use_dp and not args.no_e2e and args.dtype == "f32":
            # second kernel of the path with a roofline worth quoting: flatten_feature_set (conv, pool, conv, pool -> 784 features),
            # one launch over 8 copies of this rank's pictures, timed like the step kernels (HIP events on the context's stream)
            try:
                big = imgs_d.repeat(8, 1, 1).contiguous()
                nfe = big.shape[0]
                outf = d.empty(nfe, d.F)
                for _ in range(3):
                    d.features(big, True, outf)
                fa, fb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                fa.record(d.stream)
                for _ in range(10):
                    d.features(big, True, outf)
                fb.record(d.stream)
                d.synchronize()
                fus = fa.elapsed_time(fb) * 1e3 / 10
                fbytes = nfe * (784 + 784 * 4)
                result["roofline"]["feature_kernel"] = {
                    "kernel": "k_features_cpcp (fused standardisation)", "bound": "hbm", "images_per_launch": nfe,
                    "algorithmic_bytes_per_launch": fbytes, "us_per_launch_hip_events": round(fus, 1), "achieved": round(fbytes / fus / 1e3, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fbytes / fus / 1e3 / HBM_PEAK_GBS, 4), "images_per_s": round(nfe / fus * 1e6, 1)}
                del big, outf
            except Exception as ex:                      # a secondary figure must never cost the headline line
                result["roofline"]["feature_kernel"] = {"error": str(ex)[:200]}
        if not args.no_cpu_baseline and world == 1:      # the CPU path is timed beside the N = 1 run only
            result["cpu_baseline"] = cpu_baseline()
            result["config"]["gpu_over_cpu"] = round(result["value"] / max(result["cpu_baseline"]["value"], 1e-9), 1)
        line = json.dumps(result) + "\n"
        if real_stdout is not None:
            sys.stdout.flush()
            os.write(real_stdout, line.encode())
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    if use_dp:
        dist.barrier()                     # nobody unmaps its buffers while a peer may still read them
        if args.dp_impl == "native":
            d.dp_finalize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
