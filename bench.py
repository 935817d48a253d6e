#!/usr/bin/env python3
"""bench.py -- training images/sec of the rcn hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            # N > 1: this process only spawns the N ranks (no GPU call in it)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus 2 --dry-launch                    # CPU rehearsal of the launch path: gloo rendezvous, one line, no GPU

Workload (BASELINE.json configs[1]): synthetic MNIST-shape 28x28x1 u8 images (16 384 per rank,
seeded), the default rcn net conv(Same),pool,conv(Same),pool -> 784 -> 30 -> 10 (sigmoid, quadratic cost), N(0,1)
parameters, eta = 3.0, batch 256 PER GPU, fp32 arithmetic.  A "step" is one train_batch (rcn.rs:176-223) over one
256-image batch of the resident, already feature-extracted set -- exactly the reference's epoch-loop semantics
(features are computed once at load, rcn.rs:399-401).  The run is ONE continuous training session: step s is batch
s mod 64 of epoch s div 64; at every epoch boundary the set is shuffled on the device (rcn.rs:146) and the shuffled order
is materialised once (rcn_hip_epoch_begin_dev), then the chunks are walked (rcn.rs:147-149, rcn_hip_epoch_steps_dev).
Warm-up steps are the first W steps of that session, the timed K steps follow them directly; whatever epoch boundaries
fall inside the timed region are paid inside it.  N > 1: one process per GPU, weak scaling (global batch 256*N), summed
shard gradients exchanged once per step (include/rcn_hip.h: rcn_hip_dp_*).

Rank 0 prints ONE JSON line.  Extra objects on it: "roofline" (dominant kernel, HIP events), "cpu_baseline" (the oracle's
restatement of rcn's rayon loop timed on this host -- a reported baseline, not the target), config.steady_state_* (>= 4096
device-timed steps in the same process), config.f64_* (the reference's own arithmetic type), config.loss_curve (256 steps
against the CPU restatement on the same batches) and "trackx" (the north-star's trainable-convolution extension, CIFAR shape).
"""
from __future__ import annotations

import argparse
import hashlib
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
F32_MFMA_PEAK_TFLOPS = 157.3   # same guide: fp32 MFMA = fp32 vector peak
STEADY_STEPS = 4096
METRIC = "training images/sec, MNIST-shape 28x28x1 batch=256, at 1/2/4/8 MI355X"
WORKLOAD = ("MNIST-shape 28x28x1, rcn default net conv(Same)-pool-conv(Same)-pool -> 784-30-10 sigmoid/MSE, "
            "train_batch B=256 per GPU over 16384 resident pre-extracted feature vectors per GPU, eta=3.0")


def csrc_sha16() -> str:
    """Fingerprint of the kernel sources librcn_hip.so is built from (mercer_research_amd.build SOURCES + DEPS): profiles/r3_pmc_summary.json
    records the one it was measured on.  (Track X's sources -- convnet*.hpp, rcn_hipx_api.hip -- have their own fingerprint, tools/mfma_pmc_summary.py.)"""
    from mercer_research_amd import build as hipbuild
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mercer_research_amd", "csrc")
    for f in sorted(set(hipbuild.SOURCES + [x for x in hipbuild.DEPS if os.sep not in x and "/" not in x])):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


_NATIVE_ORACLE = []


def native_oracle():
    """oracle/rcn_oracle.c built -O3 -march=native for THIS host (the Rust --release analogue), once per process; the checker only."""
    import tempfile
    from oracle.rcn_oracle import COracle, build_oracle
    if not _NATIVE_ORACLE:
        try:
            path = build_oracle(native=True, out_dir=tempfile.mkdtemp(prefix="rcn_oracle_"))
        except Exception:
            path = build_oracle()
        _NATIVE_ORACLE.append(COracle(path))
    return _NATIVE_ORACLE[0]


def cpu_baseline(seconds_budget: float = 8.0):
    """rcn's CPU path (oracle/rcn_oracle.c restated from rcn.rs:176-314, threaded like the rayon loop) on this host:
    B = 32 (BASELINE.json configs[0]), all host cores and one core, bounded sample."""
    from oracle.rcn_oracle import DEFAULT_LAYERS, one_hot, synthetic_images, synthetic_params
    o = native_oracle()
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
        if el >= seconds_budget * 0.6 or steps >= 20000:
            break
    multi = steps * B / el
    holder1 = o.net(ws, bs)
    t0 = time.perf_counter()
    s1 = 0
    while time.perf_counter() - t0 < seconds_budget * 0.4:
        o.train_steps_inplace(holder1, X, Y, B, 8, ETA, threads=0)
        s1 += 8
    single = s1 * B / (time.perf_counter() - t0)
    return {"value": round(max(multi, single), 1), "unit": "images/s", "cores": cores if multi >= single else 1, "kind": "port",
            "host_hardware_threads": cores,
            "sample": f"{steps} train_batch steps of B={B} (784-30-10, f64) over {n} synthetic feature vectors on all {cores} hardware threads "
                      f"of this host: {multi:.0f} img/s; {s1} steps on 1 thread: {single:.0f} img/s (the mutex-serialised, reallocating sum of "
                      f"rcn.rs:192-204 does not scale, so `value` is the faster of the two and `cores` says which)",
            "threads_all_cores_images_per_s": round(multi, 1), "single_thread_images_per_s": round(single, 1)}


def dry_launch(args) -> int:
    """The launch path without a GPU: every rank joins a gloo group over 127.0.0.1, one all-reduce proves the rendezvous,
    rank 0 prints the one line.  What tests/test_bench_launch.py runs on the CPU."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    sys.stdout.flush()
    real_stdout = os.dup(1)                 # gloo (like RCCL) prints a connection banner on stdout: keep fd 1 to the one JSON line
    os.dup2(2, 1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        ranks_sum = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks_sum = 1.0
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps({"metric": METRIC, "value": None, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                          "data": "synthetic", "dry_launch": True, "rendezvous_check": ranks_sum == world * (world + 1) / 2,
                          "config": {"workload": WORKLOAD, "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}"}}) + "\n").encode())
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the secondary end-to-end (u8 image -> features -> step) measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the f64 / loss-curve / Track X legs (headline, roofline and steady state only)")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 sample-tile kernels, 2 feature-sliced pipeline")
    ap.add_argument("--dp-impl", choices=["native", "torch"], default="native",
                    help="data-parallel loop: native = exchange inside librcn_hip (rcn_hip_dp_*), torch = torch.distributed all_reduce per step")
    ap.add_argument("--dp", action="store_true", help="use the data-parallel step (gradient -> all-reduce -> apply) even at world size 1")
    ap.add_argument("--dry-launch", action="store_true", help="rendezvous the ranks over gloo on the CPU and print the line; no GPU work")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the spawning parent waits for its ranks")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="DIAGNOSTIC, not a measurement: run the N > 1 flow of this script -- rehearsal, voted trial of the exchange forms, timed loop, replica check, "
                         "phase clocks -- with all N ranks on GPU 0 (gloo carries the votes, the library's admission runs over it, every rank's resident workers sit on "
                         "a physical XCD of their own, shards of 128 so that the peers' other kernels find free CUs); what a one-GPU box can check of that flow before a node runs it")
    args = ap.parse_args()

    from mercer_research_amd.launch import spawn_ranks, under_launcher
    if args.gpus > 1 and not under_launcher():
        # `python bench.py --gpus N`: this parent makes no GPU call at all; it starts N fresh ranks and relays rank 0's line
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout_s=args.launch_timeout))
    if args.dry_launch:
        sys.exit(dry_launch(args))

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
        if args.rehearse_on_one_gpu:
            local_rank = 0                                    # every rank on this box's one GPU; RCCL refuses two ranks on one device: gloo
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = world

    import mercer_research_amd as amd
    from mercer_research_amd.device import DeviceRCN
    from mercer_research_amd.dp import DataParallelStep
    from mercer_research_amd.synth import synthetic_images, synthetic_params

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
    rehearse = bool(use_dp and args.rehearse_on_one_gpu and world > 1)
    B = 128 if rehearse else B_PER_GPU                       # (rehearsal: 28 workers per rank leave four CUs of an XCD free for the peers' other kernels)
    nb_epoch = N_IMAGES // B
    vdev = torch.device("cpu") if rehearse else d.device     # where the small vote / timing tensors of torch.distributed live
    if rehearse:
        d.set_option("xcd_select", 8 + rank % 8)
        d.set_option("xcd_exact_lds", 1)
        d.set_option("xcd_timeout_ticks", 400000000)
        d.set_option("dp_timeout_ticks", 400000000)
    step_no = [0]
    epoch_no = [0]
    perm = torch.empty(N_IMAGES, dtype=torch.int32, device=d.device)
    fallbacks = []
    dp_timeouts = []                 # records of expired in-kernel waits (rcn_hip_last_timeout): which wait, which worker, which peer

    def note_timeout(where: str):
        try:
            rec = d.last_timeout()
        except Exception:
            rec = None
        if rec and (not dp_timeouts or dp_timeouts[-1].get("launch") != rec.get("launch")):
            rec = dict(rec, during=where)
            dp_timeouts.append(rec)
            print(f"[bench] rank {rank}: expired wait during {where}: {rec.get('text', '')}", file=sys.stderr, flush=True)

    def epoch_seed():
        return 0x5DEECE66D + rank * 7919 + epoch_no[0]

    def segments(k: int):
        """(position in the epoch, number of steps) pieces that the next k steps of the session fall into"""
        pos, done, out = step_no[0] % nb_epoch, 0, []
        while done < k:
            take = min(nb_epoch - pos, k - done)
            out.append((pos, take))
            done += take
            pos = (pos + take) % nb_epoch
        return out

    allreduce_kind = None
    dp_form_trial = None
    if not use_dp:
        # begun["ok"]: the epoch image form (rcn_hip_epoch_begin_dev / _steps_dev) is available -- it is for the feature-sliced
        # pipeline; with --path 1 (sample-tile kernels) each piece gathers its rows by index instead.  live: the image holds the
        # session's current epoch.
        # Where the resident kernel fetches its rows itself (rcn_hip_train_epoch_gathers) there is nothing to materialise: the session
        # is shuffle + rcn_hip_train_epoch_dev over the shuffled order, one launch per piece.
        gathers = bool(d.train_epoch_gathers(B))
        begun = {"ok": not gathers, "live": False}
        with torch.cuda.stream(d.stream):
            perm.copy_(torch.arange(N_IMAGES, dtype=torch.int32, device=d.device))

        def lay_out():
            if begun["ok"]:
                try:
                    d.epoch_begin(X, Y, perm, B, nb_epoch)              # the shuffled order materialised once per epoch
                    begun["live"] = True
                except amd.RcnHipError:
                    begun["ok"] = False

        def prime(k: int):
            """instantiate the graphs the next k steps will replay (set-up, untimed)"""
            if begun["ok"] and not begun["live"]:
                lay_out()                                               # any image will do for capturing; run() lays out the real one
            for pos, take in set(segments(k)):
                if begun["ok"]:
                    d.epoch_steps(pos, take, ETA, None, prepare_only=True)
                else:
                    d.prepare_epoch(X, Y, perm[pos * B:], B, take, ETA, None)

        def run(k: int):
            for pos, take in segments(k):
                if pos == 0:
                    d.shuffle(perm, N_IMAGES, 1, seed=epoch_seed())     # training_set.shuffle, rcn.rs:146 -- on the device, in-stream
                    epoch_no[0] += 1
                    if take == nb_epoch:                                # a whole epoch: gather + steps in ONE library call
                        d.train_epoch(X, Y, perm, B, nb_epoch, ETA, None)
                        begun["live"] = False
                        step_no[0] += take
                        continue
                    lay_out()
                elif begun["ok"] and not begun["live"]:
                    lay_out()
                if begun["ok"]:
                    d.epoch_steps(pos, take, ETA, None)
                else:
                    d.train_epoch(X, Y, perm[pos * B:], B, take, ETA, None)
                step_no[0] += take
    else:
        # one process per GPU: every rank shuffles its own resident shard of the data, takes 256 rows per step, and the
        # summed shard gradients meet once per step (xGMI peer exchange inside the kernels, or ONE ncclAllReduce of the flat
        # parameter-shaped buffer)
        def bcast_params():
            if not rehearse:                  # (rehearsal: no RCCL communicator -- every rank sets the identical seed-42 parameters itself)
                d.dp_broadcast_params(0)

        def dp_native_setup() -> bool:
            try:
                if rehearse:
                    ok = 1 if d.dp_p2p_admit() != 0 else 0      # the library's admission procedure with gloo as its transport
                else:
                    d.dp_init()               # RCCL communicator owned by the library; torch only carried its 128-byte id
                    ok = 1
            except Exception as e:            # e.g. librccl not loadable: every rank falls back together
                print(f"[bench] native data-parallel loop unavailable on rank {rank}: {e}", file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=vdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        def dp_kind():
            if d.dp_p2p_mode() != 0 and d.dp_resident(B):
                return ("xgmi reduce-scatter + all-gather on pushed self-validating words inside the resident one-XCD step kernel "
                        "(csrc/dense_xcd.hpp DP form, csrc/dp_push.hpp: one launch per 64-step segment)")
            return {0: "ncclAllReduce (RCCL)", 1: "xgmi peer reads between kernels, fused with the update (csrc/dense_p2_dp.hpp, 3 kernels/step)",
                    2: "xgmi peer reads inside the gradient kernel (csrc/dense_p2_dp.hpp, 2 kernels/step)"}[d.dp_p2p_mode()]

        native = args.dp_impl == "native" and dp_native_setup()
        args.dp_impl = "native" if native else "torch"
        dp = None

        def prime(k: int):
            if args.dp_impl != "native":
                return
            for pos, take in set(segments(k)):
                d.dp_prepare_epoch(X, Y, perm[pos * B:], B, take, ETA, None)

        def run(k: int):
            if args.dp_impl == "native":
                # the whole loop is native: per step gradient kernels -> exchange -> update, enqueued by
                # rcn_hip_dp_train_epoch_dev; Python only starts each piece (one in-stream shuffle per epoch)
                for pos, take in segments(k):
                    if pos == 0:
                        d.shuffle(perm, N_IMAGES, 1, seed=epoch_seed())
                        epoch_no[0] += 1
                    d.dp_train_epoch(X, Y, perm[pos * B:], B, take, ETA, None)
                    step_no[0] += take
            else:
                with torch.cuda.stream(d.stream):
                    for _ in range(k):
                        pos = step_no[0] % nb_epoch
                        if pos == 0:
                            d.shuffle(perm, N_IMAGES, 1, seed=epoch_seed())
                            epoch_no[0] += 1
                        dp.train_batch(X, Y, ETA, B * world, perm=perm[pos * B:(pos + 1) * B])
                        step_no[0] += 1

        def replicas_check():
            with torch.cuda.stream(d.stream):
                pf = d.params_flat().to(torch.float64)
                chk = torch.stack([pf.sum(), (pf * pf).sum(), pf.abs().max()])
            d.stream.synchronize()
            chk = chk.to(vdev)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            return bool(torch.equal(lo, hi)) and bool(torch.isfinite(chk).all())

        def healthy(k: int) -> bool:
            """k steps of the loop as configured, then: no sticky timeout on any rank and bit-identical replicas"""
            ok = 1
            try:
                prime(k)
                run(k)
                d.synchronize()
            except Exception as e:
                print(f"[bench] rank {rank}: data-parallel loop failed its rehearsal: {e}", file=sys.stderr, flush=True)
                note_timeout("rehearsal")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=vdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            same = replicas_check() if flag.item() else False
            return bool(flag.item()) and same

        def voted(fn):
            """fn() on every rank, then ONE vote: (every rank succeeded, fn's value here).  Every rank executes the same collectives
            whatever happened locally, so a failure on one rank never leaves its peers inside a different collective."""
            ok, val = 1, None
            try:
                val = fn()
            except Exception as e:
                print(f"[bench] rank {rank}: {e}", file=sys.stderr, flush=True)
                note_timeout("a voted stretch")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=vdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item()), val

        def reset_session():
            d.set_params(ws, bs)
            step_no[0] = 0
            epoch_no[0] = 0

        # The default at N > 1 is the form that has run between real peers before: the two-kernel pipeline with the exchange inside
        # its gradient kernel (tests/: two and four processes), not the resident kernel's data-parallel form, which no run on more
        # than one GPU has been recorded for yet.  Rehearse it before anything is timed, and step down together if the
        # rehearsal fails: -> ncclAllReduce inside the library -> torch.distributed all_reduce.  Every rank takes the same branch.
        rehearsal = max(8, min(args.warmup, 64))
        resident_offered = False
        while True:
            if args.dp_impl == "native":
                if d.dp_p2p_mode() != 0 and d.dp_resident(B):
                    resident_offered = True
                    d.set_dense_path(2)                     # same group, the exchange between the halves of the two-kernel pipeline
                bcast_params()
                allreduce_kind = dp_kind()
            else:
                allreduce_kind = "torch.distributed all_reduce (RCCL)"
                dp = DataParallelStep(d)
                dp.broadcast_params(0)
            if healthy(rehearsal):
                break
            fallbacks.append(allreduce_kind)
            if rehearse:
                raise SystemExit(f"[bench] one-GPU rehearsal: the form '{allreduce_kind}' failed its rehearsal (RCCL / torch fall-backs need a GPU per rank)")
            if args.dp_impl != "native":
                raise SystemExit("[bench] the torch.distributed data-parallel loop failed its rehearsal too")
            was_p2p = d.dp_p2p_mode() != 0
            resident_offered = False
            try:
                d.dp_finalize()
            except Exception as e:
                print(f"[bench] rank {rank}: dp_finalize after a failed rehearsal: {e}", file=sys.stderr, flush=True)
            reset_session()
            if was_p2p:
                d.set_option("dp_p2p", 0)                   # same library loop on ncclAllReduce
                if not dp_native_setup():
                    args.dp_impl = "torch"
            else:
                args.dp_impl = "torch"

        # The resident one-XCD kernel carries the same step with the exchange inside it (reduce-scatter + all-gather on pushed words,
        # csrc/dp_push.hpp).  It is tried for a short, untimed stretch -- rehearsal, replica check and timing all inside votes -- and
        # kept only if it is healthy on every rank AND faster (the maximum over the ranks decides, identically everywhere).
        if args.dp_impl == "native" and resident_offered:
            def stretch(k: int) -> float:
                prime(k)
                d.synchronize(); dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(k)
                d.synchronize()
                return time.perf_counter() - t0

            def timed_trial(k: int):
                ok, el = voted(lambda: stretch(k))
                t = torch.tensor([el if ok and el is not None else float("inf")], dtype=torch.float64, device=vdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return ok, float(t.item()) * 1e6 / k

            trial_steps = 2 * nb_epoch
            ok_two, t_two = timed_trial(trial_steps)                      # (the two-kernel form is warm from its rehearsal)
            reset_session()
            ok_sw, _ = voted(lambda: d.set_dense_path(0))
            if ok_sw:
                bcast_params()
            ok_res = ok_sw and healthy(rehearsal)
            t_res = float("inf")
            if ok_res:
                ok_res, t_res = timed_trial(trial_steps)
            ok_res = ok_res and (replicas_check() if ok_res else False)
            dp_form_trial = {"two_kernel_us_per_step": round(t_two, 3) if ok_two else None, "resident_us_per_step": round(t_res, 3) if ok_res else None,
                             "resident_healthy": bool(ok_res), "steps": trial_steps}
            keep_resident = ok_res and t_res < t_two
            if not keep_resident:
                if not ok_res:
                    # the resident form failed on some rank (its sticky words stay set, replicas may be out of step): a fresh group
                    fallbacks.append("xgmi exchange inside the resident one-XCD step kernel (untimed trial)")
                    try:
                        d.dp_finalize()
                    except Exception as e:
                        print(f"[bench] rank {rank}: dp_finalize after the resident trial: {e}", file=sys.stderr, flush=True)
                    d.set_dense_path(2)
                    if not dp_native_setup():
                        raise SystemExit("[bench] the data-parallel group could not be set up again after the resident trial")
                else:
                    d.set_dense_path(2)
            dp_form_trial["kept"] = "resident kernel" if keep_resident else "two-kernel pipeline"
            reset_session()
            bcast_params()
            if not healthy(rehearsal):
                raise SystemExit("[bench] the selected data-parallel form failed its final rehearsal")
            reset_session()
            bcast_params()
            allreduce_kind = dp_kind()

    def sync():
        d.synchronize()
        torch.cuda.synchronize()
        if use_dp:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(k: int):
        """k steps bracketed by barrier + synchronise on both sides: (wall seconds [max over ranks], device ms on this rank)"""
        prime(k)
        prime(k)            # twice: a call shape that needs a larger workspace moves it, which drops the graphs captured before it
        sync()
        t0 = time.perf_counter()
        run(k)
        torch.cuda.synchronize()          # every stream of the device, the library's included
        el = time.perf_counter() - t0     # this rank's K steps are done; the MAX over the ranks (below) is the job's time
        if use_dp:
            dist.barrier()                # the closing barrier + synchronise of the bracket: the slowest rank's clock has stopped by now,
            torch.cuda.synchronize()      # and the barrier's own latency (an all-reduce launch) is not training time
        d.synchronize()                   # (outside the bracket: reports a sticky in-kernel time-out of the timed steps, if any)
        # the device-side time of the same k steps, from a second pass of the session (event records are host calls of several
        # microseconds each: kept out of the wall-clock bracket above, which is the contract's figure)
        prime(k)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(d.stream)
        run(k)
        ev1.record(d.stream)
        sync()
        dev_ms = ev0.elapsed_time(ev1)
        if use_dp:
            t = torch.tensor([el], dtype=torch.float64, device=vdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, dev_ms

    prime(args.warmup)
    run(args.warmup)
    start_pos = step_no[0] % nb_epoch
    epoch_boundary_in_bracket = start_pos == 0 or start_pos + args.steps > nb_epoch      # a shuffle + gather of the epoch falls inside the timed steps
    elapsed, dev_ms = timed(args.steps)
    loss_t = d.empty(1)
    d.train_batch(X[:B], Y[:B], 0.0, loss_t)                             # eta = 0: reads the current cost, changes nothing
    d.synchronize()
    final_loss = float(loss_t.item())
    if not use_dp:
        step_no[0] = (step_no[0] + nb_epoch - 1) // nb_epoch * nb_epoch     # train_batch re-packed the image: the session resumes at the next epoch
        begun["live"] = False

    # ---- steady state: the same session carried on for >= 4096 more steps, in this process, device-timed ----
    if args.steps >= STEADY_STEPS:
        steady_el, steady_dev_ms, steady_k = elapsed, dev_ms, args.steps
    else:
        steady_k = STEADY_STEPS
        steady_el, steady_dev_ms = timed(steady_k)

    # SURVEY.md §8(d) asks for two numbers: train-only (the headline `value`: features resident, the reference's epoch-loop
    # semantics) and END-TO-END: u8 image -> feature kernel (+ standardise, fused) -> train_batch, i.e. the features of
    # every pass are recomputed from the resident u8 images inside the timed region.
    e2e = e2e_steady = None
    if not use_dp and not args.no_e2e:
        perm1 = perm
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
        e2e_steady = reps * N_IMAGES / (ea.elapsed_time(eb) * 1e-3)

        # ... and over exactly --steps steps inside the same barrier + synchronise bracket and by the same clock as `value`, so that
        # the two are comparable at any step count (32 device-timed passes next to a 20-step wall-clock figure were not)
        def e2e_steps(k: int, seed0: int):
            done = 0
            while done < k:
                take = min(nb_epoch, k - done)
                d.shuffle(perm1, N_IMAGES, 1, seed=seed0 + done)
                if fused_e2e:
                    d.train_epoch_images(imgs_d, Y, perm1, B, take, ETA, None)
                else:
                    d.features(imgs_d, True, Xe)
                    d.train_epoch(Xe, Y, perm1, B, take, ETA, None)
                done += take
        e2e_steps(args.steps, 0xE2E0000)                     # the call shapes of the timed run, once, untimed
        sync()
        t0 = time.perf_counter()
        e2e_steps(args.steps, 0xE2E1000)
        torch.cuda.synchronize()                                 # the same end of the bracket as in timed()
        e2e = args.steps * B / (time.perf_counter() - t0)
        d.synchronize()

    # where a data-parallel step on the resident kernel waits (diagnostic launches with per-worker clocks, AFTER everything timed): what a
    # first run on more than one GPU needs to explain itself -- rcn_hip_dp_phase_us
    dp_phase = None
    if use_dp and args.dp_impl == "native":
        def clocked():
            if not (d.dp_p2p_mode() != 0 and d.dp_resident(B)):
                return None
            d.set_option("xcd_dp_phase", 1)
            try:
                step_no[0] = 0
                run(nb_epoch)
                d.synchronize()
                return d.dp_phase_us()
            finally:
                d.set_option("xcd_dp_phase", 0)
        ok_ph, dp_phase = voted(clocked)
        if not ok_ph:
            dp_phase = None

    # data-parallel runs: every rank must hold bit-identical parameters (identical update from rank-ordered sums); a stale or
    # torn read in the exchange would show up here as diverged replicas
    replicas_identical = replicas_check() if use_dp else None

    images = args.steps * B * world
    result = {
        "metric": METRIC,
        "value": round(images / elapsed, 1), "unit": "images/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 6), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": WORKLOAD if not rehearse else WORKLOAD.replace("B=256 per GPU", "B=128 per rank (ONE-GPU REHEARSAL)"),
                   **({"one_gpu_rehearsal": f"DIAGNOSTIC, NOT A MEASUREMENT: all {world} ranks share GPU 0 (gloo votes, the library's admission over gloo, rank r's resident "
                                            "workers on physical XCD r, shards of 128); it exercises this script's N > 1 flow, its numbers mean nothing"} if rehearse else {}),
                   "global_batch": B * world, "parallelism": f"dp{world}", "step_form": f"gradient -> exchange -> apply ({args.dp_impl} loop)" if use_dp else "fused update", "images_per_rank": N_IMAGES,
                   "session": f"steps {args.warmup}..{args.warmup + args.steps} of one continuous training session (64 steps per epoch; device shuffle "
                              + ("at every epoch boundary, rows fetched by the step kernel itself in the shuffled order -- no packed copy of the epoch"
                                 if (not use_dp and gathers) else "+ one gather of the shuffled order at every epoch boundary")
                              + ", inside the timed region whenever it is crossed -- see epoch_boundary_in_bracket; steady_state_* always carries its 1/64 share)",
                   "epoch_boundary_in_bracket": bool(epoch_boundary_in_bracket),
                   "device_ms_per_step_rank0": round(dev_ms / args.steps, 6), "final_cost_rank0": final_loss,
                   "steady_state_steps": steady_k,
                   "steady_state_images_per_s": round(steady_k * B * world / steady_el, 1),
                   "steady_state_us_per_step": round(steady_el * 1e6 / steady_k, 4),
                   "steady_state_device_us_per_step_rank0": round(steady_dev_ms * 1e3 / steady_k, 4),
                   "end_to_end_images_per_s": round(e2e, 1) if e2e else None,                       # same steps, bracket and clock as `value`
                   "end_to_end_steady_state_images_per_s": round(e2e_steady, 1) if e2e_steady else None,   # 32 passes over the set, device-timed
                   "allreduce": allreduce_kind if use_dp else None, "replicas_identical": replicas_identical,
                   "dp_fallbacks_taken": fallbacks if use_dp else None, "dp_form_trial": dp_form_trial,
                   # an in-kernel wait that expired on rank 0 while the forms were rehearsed / tried (rcn_hip_last_timeout: site 1 placement vote,
                   # 2 tail flag, 3 slab flag, 4 delta flag, 5 pushed reduce-scatter, 6 pushed all-gather, 7 tail all-to-all, 8 cost all-to-all, 9 closing round)
                   # where a step of the resident data-parallel form waits, microseconds per step on rank 0 (clocked diagnostic launches: each clock read
                   # costs most of a microsecond, so `step` here is NOT the timed step): owners wait for the ranks' partial sums, members for the totals
                   "dp_phase_us": dp_phase,
                   "dp_rs_wait_us": dp_phase["owner_wait_mean"] if dp_phase else None, "dp_ag_wait_us": dp_phase["member_wait_mean"] if dp_phase else None,
                   "dp_tail_wait_us": dp_phase["tail_all_to_all_mean"] if dp_phase else None,
                   "dp_timeout_site": (dp_timeouts[0]["site"] if dp_timeouts else None) if use_dp else None,
                   "dp_timeout_worker": (dp_timeouts[0]["worker"] if dp_timeouts else None) if use_dp else None,
                   "dp_timeout_missing": (dp_timeouts[0]["missing"] if dp_timeouts else None) if use_dp else None,
                   "dp_timeout_record": (dp_timeouts[0].get("text", "")[:600] if dp_timeouts else None) if use_dp else None,
                   "timing": ("barrier + synchronise, clock started on every rank; K steps; synchronise, clock stopped on every rank, closing barrier + "
                              "synchronise; value uses the MAX over the ranks") if use_dp else "synchronise; K steps; one device-wide synchronise"},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events on the stream the kernels run on) ----
        # HIP events on the kernels' own stream: each kernel back to back with itself, and the alternating pair as the epoch
        # loop issues it (walking the last epoch's packed batches).  A kernel's duration inside the real loop is the pair
        # time split in the ratio of the two stand-alone times; that is what a profiler's per-dispatch average shows.
        if not use_dp and begun["ok"]:
            d.shuffle(perm, N_IMAGES, 1, seed=0x7001)
            lay_out()                                                    # time_kernels walks the batches of the image it finds
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
        resident = us_first == 0.0                  # the resident one-XCD kernel ran the step: one launch per epoch segment, no first / second kernel
        if resident:
            # k_xcd_epoch (csrc/dense_xcd.hpp) does the whole step; a launch processes `steps_per_launch` batches.  achieved = the
            # step's algorithmic bytes x steps per launch / the launch's duration = per-step bytes / per-step time.
            names, by, k = ("k_xcd_epoch", "k_xcd_epoch"), (B * (per_img_main + per_img_tail), B * (per_img_main + per_img_tail)), 1
        elif args.path == 1:
            names, by = ("k_dense_fwd", "k_dense_wgrad"), (B * (F * es + C * es + P * es / B), B * (P * es / B))
            k = 0 if us_first >= us_second else 1
        else:
            names, by = ("k_p2_b", "k_p2_a"), (B * per_img_tail, B * per_img_main)
            k = 1                                   # k_p2_a owns the features and W_0; k_p2_b is the small tail kernel
        us = us_second if resident else us_pair * (us_first, us_second)[k] / (us_first + us_second)
        flops_step = 2 * B * ((F * H + H * C) * 2 + H * C)
        traffic, traffic_src, traffic_stale = None, None, None
        pmc_name = next((f for f in ("r4_pmc_summary.json", "r3_pmc_summary.json", "r2_pmc_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
        pmc = os.path.join(ROOT, "profiles", pmc_name or "none")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get(names[k], {}).get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; tools/prof_r4.sh)"
                traffic_stale = pj.get("csrc_sha16") != csrc_sha16()     # True: kernels changed since the counters were collected
            except Exception:
                traffic = None
        steps_per_launch = nb_epoch if resident else 1
        result["roofline"] = {"bound": "hbm", "kernel": names[k], "achieved": round(by[k] / us / 1e3, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(by[k] / us / 1e3 / HBM_PEAK_GBS, 5), "traffic": round(traffic) if traffic else None,
                              "traffic_per_step": round(traffic / steps_per_launch) if traffic else None,
                              "traffic_source": traffic_src, "traffic_measured_on_other_kernel_sources": traffic_stale,
                              "traffic_over_algorithmic": round(traffic / (by[k] * steps_per_launch), 2) if traffic else None,
                              "steps_per_launch": steps_per_launch,
                              "algorithmic_bytes_per_launch": round(by[k] * steps_per_launch), "algorithmic_bytes_per_step": round(by[k]) if resident else None,
                              "us_per_launch_hip_events": round(us * steps_per_launch, 3), "us_per_step_hip_events": round(us, 3) if resident else None,
                              "us_alternating_pair": None if resident else round(us_pair, 3),
                              **({} if resident else {"us_standalone_" + names[0]: round(us_first, 3), "us_standalone_" + names[1]: round(us_second, 3)}),
                              "step_gflops_per_s": round(flops_step / (steady_el / steady_k) / 1e9, 1),
                              "mfma_floor_us_per_step": round(2 * 256 * 8 / 2.4e3, 3) if resident else None,
                              "note": ("one train_batch at B=256 is ~1 MB and ~25 MFLOP.  The resident kernel keeps the step on the 32 CUs of one XCD: its floor is the 512 "
                                       "f32 MFMAs per feature worker and step (8 cycles per CU each: 1.7 us) plus two in-launch hand-offs through that XCD's L2, "
                                       "far from the HBM roof by construction -- the batch is read from memory exactly once per step, parameters and partial sums never leave the chip")
                                      if resident else
                                      "one train_batch at B=256 is ~1 MB and ~25 MFLOP: bound by launch + dependent-latency floors "
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
        extras = not use_dp and not args.no_extras
        if extras and args.dtype == "f32":
            try:
                result["config"].update(f64_leg(torch, amd, DeviceRCN, imgs_d, labels_d, ws, bs, local_rank))
            except Exception as ex:
                result["config"]["f64_error"] = str(ex)[:200]
        if extras and args.dtype == "f32":
            try:
                result["config"]["reference_test_net"] = test_net_leg(torch, amd, DeviceRCN, imgs_d, labels_d, local_rank)
            except Exception as ex:
                result["config"]["reference_test_net"] = {"error": str(ex)[:200]}
        if extras and args.dtype == "f32":
            try:
                result["config"]["small_batch"] = small_batch_leg(torch, amd, DeviceRCN, imgs_d, labels_d, ws, bs, local_rank)
            except Exception as ex:
                result["config"]["small_batch"] = {"error": str(ex)[:300]}
        if extras:
            try:
                result["config"]["loss_curve"] = loss_curve_leg(torch, amd, DeviceRCN, imgs, labels, imgs_d, labels_d, ws, bs, local_rank, dtype)
            except Exception as ex:
                result["config"]["loss_curve"] = {"error": str(ex)[:300]}
            try:
                result["trackx"] = trackx_leg(torch, local_rank)
            except Exception as ex:
                result["trackx"] = {"error": str(ex)[:300]}
        # the round's figures once more as FLAT scalar keys of `config` (the driver's record keeps scalars of `config` only and the last
        # 2000 characters of the line; the nested objects above stay for readers)
        def _g(obj, *path):
            for k in path:
                if not isinstance(obj, dict) or k not in obj:
                    return None
                obj = obj[k]
            return obj
        cfg = result["config"]
        sb, tx = cfg.get("small_batch"), result.get("trackx")
        flat = {
            "sb_b10_f32_us": _g(sb, "B10", "default_path", "us_per_step"), "sb_b32_f32_us": _g(sb, "B32", "default_path", "us_per_step"),
            "sb_b10_f64_us": _g(sb, "B10", "f64_default_path", "us_per_step"), "sb_b32_f64_us": _g(sb, "B32", "f64_default_path", "us_per_step"),
            "sb_b10_f32_resident": _g(sb, "B10", "default_path", "resident_kernel"), "sb_b32_f32_resident": _g(sb, "B32", "default_path", "resident_kernel"),
            "sb_b10_f64_resident": _g(sb, "B10", "f64_default_path", "resident_kernel"), "sb_b32_f64_resident": _g(sb, "B32", "f64_default_path", "resident_kernel"),
            "sb_b10_cpu_steps_per_s": _g(sb, "B10", "cpu_restatement_f64", "one_thread", "steps_per_s"),
            "curve_max_rel_dev": _g(cfg, "loss_curve", "max_rel_dev_over_curve"),
            "testnet_us": _g(cfg, "reference_test_net", "default_path", "us_per_step"),
            "x_cifar_f32_ms": _g(tx, "fp32", "ms_per_step"), "x_cifar_f32_frac": _g(tx, "fp32", "frac_of_mfma_peak"), "x_cifar_bf16_ms": _g(tx, "bf16", "ms_per_step"),
            "x_224_f32_ms": _g(tx, "synth224", "fp32", "ms_per_step"), "x_224_f32_frac": _g(tx, "synth224", "fp32", "frac_of_mfma_peak"),
            "x_224_bf16_ms": _g(tx, "synth224", "bf16", "ms_per_step"), "x_mnist4096_bf16_ms": _g(tx, "mnist4096_bf16", "ms_per_step"),
            "x_cifar_bf16_stored_ms": _g(tx, "bf16_stored", "ms_per_step"), "x_224_bf16_stored_ms": _g(tx, "synth224", "bf16_stored", "ms_per_step"),
            "x_224_bf16_stored_frac_of_hbm_floor": _g(tx, "synth224", "bf16_stored", "frac_of_hbm_floor"),
            "x_mnist4096_bf16_stored_ms": _g(tx, "mnist4096_bf16_stored", "ms_per_step"),
            "x_conv_gemm_mfma_busy": _g(tx, "conv_gemm_mfma_busy", "time_weighted_over_3x3_conv_gemm_kernels"),
        }
        for k, v in flat.items():
            if isinstance(v, bool):
                v = int(v)
            if v is not None:
                cfg[k] = v
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
            try:
                d.dp_finalize()
            except Exception as e:
                print(f"[bench] rank {rank}: dp_finalize: {e}", file=sys.stderr, flush=True)
        dist.destroy_process_group()


def f64_leg(torch, amd, DeviceRCN, imgs_d, labels_d, ws, bs, dev):
    """The reference's own arithmetic type (f64 throughout, rcn.rs:28,31,49) on the same pipeline and workload: 1024 device-timed
    steps of the same session shape."""
    d = DeviceRCN(classes=10, feedforward_cfg=[30], input_shape=(28, 28), dtype=amd.F64, device=dev)
    d.set_params(ws, bs)
    X, Y = d.load_data(imgs_d, labels_d)
    nb, B = N_IMAGES // B_PER_GPU, B_PER_GPU
    perm = torch.empty(N_IMAGES, dtype=torch.int32, device=d.device)

    def epochs(n, seed0):
        for e in range(n):
            d.shuffle(perm, N_IMAGES, 1, seed=seed0 + e)
            d.epoch_begin(X, Y, perm, B, nb)
            d.epoch_steps(0, nb, ETA, None)
    epochs(2, 11)
    d.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 16
    a.record(d.stream)
    epochs(n, 100)
    b.record(d.stream)
    d.synchronize()
    us = a.elapsed_time(b) * 1e3 / (n * nb)
    d.rcn.close()
    return {"f64_images_per_s": round(B / us * 1e6, 1), "f64_us_per_step": round(us, 4), "f64_steps": n * nb}


def small_batch_leg(torch, amd, DeviceRCN, imgs_d, labels_d, ws, bs, dev):
    """The reference's OWN operating points: batch_size 10 (rcn/src/main.rs:36-37, benches/train.rs:22, rcn.rs:581) and 32 (BASELINE.json
    configs[0]), same data and session shape -- one shuffled epoch of chunks_exact(B) per call, device-timed -- on the default path
    (the resident one-XCD kernel's instantiation for batches <= 32) and on the sample-tile kernels it replaces there, beside the CPU
    restatement of rcn's rayon loop (oracle/rcn_oracle.c) at the same B on one thread and on all threads of this host."""
    from oracle.rcn_oracle import DEFAULT_LAYERS, one_hot
    from mercer_research_amd.synth import synthetic_images
    out = {}
    o = native_oracle()
    n_cpu = 2048
    imgs_h, labels_h = synthetic_images(n_cpu)
    feats = o.features(imgs_h, DEFAULT_LAYERS)
    m, s = o.gen_scales(feats)
    Xh, Yh = o.standardize(feats, m, s), one_hot(labels_h)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for B in (10, 32):
        nb = N_IMAGES // B                                  # chunks_exact(B): the tail is dropped (rcn.rs:147)
        row = {}
        # f64_default_path: the reference's own arithmetic type at its own batch size -- since round 4 on the resident kernel's f64 instantiation
        for name, path, dt in (("default_path", 0, amd.F32), ("sample_tile_kernels", 1, amd.F32), ("f64_default_path", 0, amd.F64)):
            d = DeviceRCN(classes=10, feedforward_cfg=[30], input_shape=(28, 28), dtype=dt, device=dev)
            d.set_params(ws, bs)
            if path:
                d.set_dense_path(path)
            X, Y = d.load_data(imgs_d, labels_d)
            perm = torch.empty(N_IMAGES, dtype=torch.int32, device=d.device)

            def epochs(n, seed0):
                for e in range(n):
                    d.shuffle(perm, N_IMAGES, 1, seed=seed0 + e)
                    d.train_epoch(X, Y, perm, B, nb, ETA, None)
            epochs(1, 11)
            d.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 3 if name == "default_path" else 1
            a.record(d.stream)
            epochs(n, 100)
            b.record(d.stream)
            d.synchronize()
            us = a.elapsed_time(b) * 1e3 / (n * nb)
            k1, _, _ = d.time_kernels(X[:B], Y[:B], reps=64)
            row[name] = {"us_per_step": round(us, 3), "steps_per_s": round(1e6 / us, 1), "images_per_s": round(B / us * 1e6, 1),
                         "resident_kernel": k1 == 0.0, "steps": n * nb, "fallbacks": d.fallbacks_taken()}
            d.rcn.close()
        cpu = {}
        for label, thr in (("one_thread", 0), ("all_threads", cores)):
            h = o.net(ws, bs)
            o.train_steps_inplace(h, Xh, Yh, B, 8, ETA, threads=thr)
            t0, k = time.perf_counter(), 0
            while time.perf_counter() - t0 < 0.6:
                o.train_steps_inplace(h, Xh, Yh, B, 16, ETA, threads=thr)
                k += 16
            el = time.perf_counter() - t0
            cpu[label] = {"steps_per_s": round(k / el, 1), "images_per_s": round(k * B / el, 1)}
        row["cpu_restatement_f64"] = dict(cpu, host_hardware_threads=cores)
        row["gpu_over_cpu"] = round(row["default_path"]["images_per_s"] / max(cpu["one_thread"]["images_per_s"], cpu["all_threads"]["images_per_s"]), 1)
        out[f"B{B}"] = row
    return out


def test_net_leg(torch, amd, DeviceRCN, imgs_d, labels_d, dev):
    """The reference's OWN test net (two hidden layers of ten, rcn.rs:558,577) on the same data and session shape: device-timed
    steps on the default path (the resident kernel's two-hidden-layer instantiation where it applies) and on the generic two-kernel
    pipeline it replaces there."""
    from mercer_research_amd.synth import synthetic_params
    nb, B = N_IMAGES // B_PER_GPU, B_PER_GPU
    out = {"dims": [784, 10, 10, 10]}
    for name, path in (("default_path", 0), ("generic_pipeline", 2)):
        d = DeviceRCN(classes=10, feedforward_cfg=[10, 10], input_shape=(28, 28), dtype=amd.F32, device=dev)
        ws, bs = synthetic_params([784, 10, 10, 10], seed=42)
        d.set_params(ws, bs)
        if path:
            d.set_dense_path(path)
        X, Y = d.load_data(imgs_d, labels_d)
        perm = torch.empty(N_IMAGES, dtype=torch.int32, device=d.device)

        def epochs(n, seed0):
            for e in range(n):
                d.shuffle(perm, N_IMAGES, 1, seed=seed0 + e)
                d.train_epoch(X, Y, perm, B, nb, ETA, None)
        epochs(2, 11)
        d.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 16
        a.record(d.stream)
        epochs(n, 100)
        b.record(d.stream)
        d.synchronize()
        us = a.elapsed_time(b) * 1e3 / (n * nb)
        k1, _, kp = d.time_kernels(X[:B], Y[:B], reps=128)
        out[name] = {"us_per_step": round(us, 3), "images_per_s": round(B / us * 1e6, 1), "resident_kernel": k1 == 0.0, "steps": n * nb}
        d.rcn.close()
    return out


def loss_curve_leg(torch, amd, DeviceRCN, imgs, labels, imgs_d, labels_d, ws, bs, dev, dtype, steps: int = 256):
    """Matched loss curve: `steps` consecutive train_batch steps of the bench workload (B = 256, eta = 3, N(0,1) parameters of seed
    42, four shuffled epochs) on the GPU with the per-step cost recorded, against the CPU restatement (oracle/rcn_oracle.c, f64)
    fed the identical batches in the identical order."""
    from oracle.rcn_oracle import COracle, DEFAULT_LAYERS, one_hot
    d = DeviceRCN(classes=10, feedforward_cfg=[30], input_shape=(28, 28), dtype=dtype, device=dev)
    d.set_params(ws, bs)
    X, Y = d.load_data(imgs_d, labels_d)
    nb, B = N_IMAGES // B_PER_GPU, B_PER_GPU
    perm = torch.empty(N_IMAGES, dtype=torch.int32, device=d.device)
    loss = d.empty(steps)
    perms = []
    for e in range(steps // nb):
        d.shuffle(perm, N_IMAGES, 1, seed=0xC0FFEE + e)
        d.synchronize()
        perms.append(perm.cpu().numpy().astype(np.int64))
        d.epoch_begin(X, Y, perm, B, nb)
        d.epoch_steps(0, nb, ETA, loss[e * nb:])
    d.synchronize()
    gpu = loss.double().cpu().numpy()
    d.rcn.close()
    o = COracle()
    t0 = time.perf_counter()
    feats = o.features(imgs, DEFAULT_LAYERS)
    m, s = o.gen_scales(feats)
    Xo, Yo = o.standardize(feats, m, s), one_hot(labels)
    h = o.net(ws, bs)
    cpu = np.zeros(steps)
    import ctypes as C
    for e, p in enumerate(perms):
        for j in range(nb):
            idx = p[j * B:(j + 1) * B]
            xb, yb = np.ascontiguousarray(Xo[idx]), np.ascontiguousarray(Yo[idx])
            cpu[e * nb + j] = o.lib.rcn_o_train_batch(C.byref(h.net), xb.ctypes.data_as(C.POINTER(C.c_double)), yb.ctypes.data_as(C.POINTER(C.c_double)), B, ETA)
    rel = np.abs(gpu - cpu) / np.maximum(np.abs(cpu), 1e-300)
    return {"steps": steps, "final_cost_gpu": float(gpu[-1]), "final_cost_cpu_restatement_f64": float(cpu[-1]),
            "max_rel_dev_over_curve": float(rel.max()), "max_rel_dev_first_32_steps": float(rel[:32].max()),
            "mean_cost_gpu": float(gpu.mean()), "mean_cost_cpu": float(cpu.mean()),
            "cpu_seconds": round(time.perf_counter() - t0, 2),
            "note": "per-step quadratic cost before each update, identical batches in identical order; eta = 3 on un-scaled N(0,1) parameters "
                    "saturates the sigmoids, so f32 rounding differences grow along the trajectory (tests/test_gpu_parity.py states the envelope)"}


def trackx_leg(torch, dev):
    """North-star extension (no reference counterpart): the trainable-convolution net of BASELINE.json configs[2] -- CIFAR-10 shape
    32x32x3, 3 conv + 2 dense, B = 512, fp32 MFMA -- whole training step, hipGraph-replayed; plus the bf16-operand form."""
    from bench_convnet import BF16_MFMA_PEAK_TFLOPS, CONFIGS
    from mercer_research_amd.convnet import ConvNet
    in_shape, layers, B = CONFIGS["cifar"]
    out = {"config": "CIFAR-10 shape 32x32x3, conv3x3 3->32, pool, 32->64, pool, 64->128, pool -> 2048 -> 256 -> 10, batch 512, softmax + cross-entropy, SGD"}
    rng = np.random.default_rng(0)
    for prec in ("fp32", "bf16", "bf16_stored"):
        net = ConvNet(in_shape, layers, B, device=dev)
        net.init_params(1)
        net.set_precision(prec)
        xs = [net.to_device(rng.standard_normal((B,) + in_shape).astype(np.float32)) for _ in range(4)]
        ys = [net.to_device(rng.integers(0, 10, B).astype(np.int32)) for _ in range(4)]
        loss = torch.zeros(1, dtype=torch.float32, device=net.device)
        for i in range(16):
            net.train_step(xs[i % 4], ys[i % 4], 0.01, loss)
        net.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        runs = []
        for _ in range(3):                                      # three stretches of 100 steps, the median reported (one single-stretch reading of this round came out 1.7 x its neighbours: 0.707 vs 0.421 ms)
            a.record(net.stream)
            for i in range(n):
                net.train_step(xs[i % 4], ys[i % 4], 0.01, loss)
            b.record(net.stream)
            net.synchronize()
            runs.append(a.elapsed_time(b) / n)
        ms = sorted(runs)[1]
        flops = net.step_flops(B)
        tf = flops / (ms * 1e-3) / 1e12
        peak = F32_MFMA_PEAK_TFLOPS if prec == "fp32" else BF16_MFMA_PEAK_TFLOPS
        floor_ms = net.step_hbm_floor_bytes(B, stored16=prec == "bf16_stored") / (HBM_PEAK_GBS * 1e9) * 1e3
        out[prec] = {"ms_per_step": round(ms, 4), "images_per_s": round(B / ms * 1e3, 1), "step_gflop": round(flops / 1e9, 2), "tflops": round(tf, 2),
                     "mfma_peak_tflops": peak, "frac_of_mfma_peak": round(tf / peak, 4), "final_loss": round(float(loss.item()), 4),
                     "hbm_floor_ms_as_stored": round(floor_ms, 4), "frac_of_hbm_floor": round(floor_ms / ms, 4),
                     "ms_per_step_of_each_stretch": [round(r, 4) for r in runs]}
        net.close()
    # BASELINE configs[3] on one GPU: synthetic 224x224x3, 8 conv layers, 128 images per GPU (fewer timed steps: 20 ms each)
    try:
        in_shape, layers, B = CONFIGS["synth224"]
        out["synth224"] = {"config": "synthetic 224x224x3, conv 3->32->32 | 64->64 | 128->128 | 256->256 (pool after each pair) -> 10, 128 images per GPU"}
        for prec in ("fp32", "bf16", "bf16_stored"):
            net = ConvNet(in_shape, layers, B, device=dev)
            net.init_params(1)
            net.set_precision(prec)
            xs = [net.to_device(rng.standard_normal((B,) + in_shape).astype(np.float32)) for _ in range(2)]
            ys = [net.to_device(rng.integers(0, 10, B).astype(np.int32)) for _ in range(2)]
            loss = torch.zeros(1, dtype=torch.float32, device=net.device)
            for i in range(6):
                net.train_step(xs[i % 2], ys[i % 2], 1e-6, loss)
            net.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 12
            runs = []
            for _ in range(3):                                  # three stretches, the median reported
                a.record(net.stream)
                for i in range(n):
                    net.train_step(xs[i % 2], ys[i % 2], 1e-6, loss)
                b.record(net.stream)
                net.synchronize()
                runs.append(a.elapsed_time(b) / n)
            ms = sorted(runs)[1]
            flops = net.step_flops(B)
            tf = flops / (ms * 1e-3) / 1e12
            peak = F32_MFMA_PEAK_TFLOPS if prec == "fp32" else BF16_MFMA_PEAK_TFLOPS
            floor_ms = net.step_hbm_floor_bytes(B, stored16=prec == "bf16_stored") / (HBM_PEAK_GBS * 1e9) * 1e3
            out["synth224"][prec] = {"ms_per_step": round(ms, 3), "images_per_s": round(B / ms * 1e3, 1), "step_gflop": round(flops / 1e9, 1), "tflops": round(tf, 2),
                                     "mfma_peak_tflops": peak, "frac_of_mfma_peak": round(tf / peak, 4),
                                     "hbm_floor_ms_as_stored": round(floor_ms, 3), "frac_of_hbm_floor": round(floor_ms / ms, 4)}
            net.close()
            del xs, ys
            torch.cuda.empty_cache()
    except Exception as ex:                                     # the extension must not take the BASELINE line down with it
        out["synth224"] = {"error": str(ex)[:300]}
    # BASELINE configs[4] on one GPU: MNIST shape, bf16 MFMA, batch 4096, the step replayed as a captured hipGraph; then the same with the
    # convolutional stage's tensors stored as bf16
    for prec, key in (("bf16", "mnist4096_bf16"), ("bf16_stored", "mnist4096_bf16_stored")):
        try:
            in_shape, layers, _ = CONFIGS["mnist"]
            B = 4096
            net = ConvNet(in_shape, layers, B, device=dev)
            net.init_params(1)
            net.set_precision(prec)
            xs = [net.to_device(rng.standard_normal((B,) + in_shape).astype(np.float32)) for _ in range(2)]
            ys = [net.to_device(rng.integers(0, 10, B).astype(np.int32)) for _ in range(2)]
            loss = torch.zeros(1, dtype=torch.float32, device=net.device)
            for i in range(8):
                net.train_step(xs[i % 2], ys[i % 2], 0.01, loss)
            net.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            runs = []
            for _ in range(3):                                  # three stretches, the median reported (as for the CIFAR shape above)
                a.record(net.stream)
                for i in range(n):
                    net.train_step(xs[i % 2], ys[i % 2], 0.01, loss)
                b.record(net.stream)
                net.synchronize()
                runs.append(a.elapsed_time(b) / n)
            ms = sorted(runs)[1]
            flops = net.step_flops(B)
            floor_ms = net.step_hbm_floor_bytes(B, stored16=prec == "bf16_stored") / (HBM_PEAK_GBS * 1e9) * 1e3
            out[key] = {"config": "MNIST shape 28x28x1, conv 1->32, pool, 32->64, pool -> 128 -> 10, batch 4096, bf16 MFMA operands, hipGraph step" +
                                  (", conv-stage activations and gradients stored as bf16" if prec == "bf16_stored" else ""),
                        "ms_per_step": round(ms, 4), "images_per_s": round(B / ms * 1e3, 1), "tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                        "frac_of_mfma_peak": round(flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                        "hbm_floor_ms_as_stored": round(floor_ms, 4), "frac_of_hbm_floor": round(floor_ms / ms, 4),
                        "ms_per_step_of_each_stretch": [round(r, 4) for r in runs]}
            net.close()
            del xs, ys
            torch.cuda.empty_cache()
        except Exception as ex:
            out[key] = {"error": str(ex)[:300]}
    # the conv-GEMM MFMA-busy figures are PMC measurements of a separate profiled run (tools/prof_trackx.sh), relayed here with the
    # fingerprint of the kernel sources they were taken on -- like roofline.traffic, the line says when the sources have changed since
    from tools.mfma_pmc_summary import trackx_sha16
    for name in ("r4_trackx_mfma_pmc.json", "r3_trackx_mfma_pmc.json", "r2_trackx_mfma_pmc.json"):
        pm = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pm):
            continue
        try:
            pj = json.load(open(pm))
            meta = pj.get("_meta", {})
            conv = {k: v["mfma_busy_fraction_of_simd_cycles"] for k, v in pj.items() if ("k_conv_fwd<3, false" in k or "k_conv3x3_" in k) and "mfma_busy_fraction_of_simd_cycles" in v}
            wg = {k: v["mfma_busy_fraction_of_simd_cycles"] for k, v in pj.items() if ("k_conv_wgrad<3, false" in k or "k_wgrad3x3_" in k) and "mfma_busy_fraction_of_simd_cycles" in v}
            out["conv_gemm_mfma_busy"] = {"forward_and_dgrad_kernels": conv, "wgrad_kernels": wg,
                                          "time_weighted_over_3x3_conv_gemm_kernels": meta.get("conv3x3_gemm_time_weighted_mfma_busy"),
                                          "measured_on_other_kernel_sources": (meta.get("trackx_sha16") != trackx_sha16()) if meta.get("trackx_sha16") else True,
                                          "source": f"profiles/{name} (tools/prof_trackx.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over "
                                                    "bench_convnet.py --config cifar; fraction = MFMA-busy cycles / (GRBM_GUI_ACTIVE x 128), 1.0 = every SIMD's matrix pipe busy "
                                                    "for the kernel's whole duration; a relayed measurement, not taken in this run)"}
        except Exception:
            pass
        break
    return out


if __name__ == "__main__":
    main()
