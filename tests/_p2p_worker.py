"""Worker of the multi-process peer-exchange tests (test_gpu_parity.test_peer_allreduce_two_processes_one_gpu and
test_gpu_round2): one data-parallel rank.  Run as
    python tests/_p2p_worker.py <rank> <world> <port> <dtype> <outdir> [case]
ranks share GPU 0 (the dev box has one), which exercises the whole protocol -- hipIpc export/attach, flags, double buffering,
rank-order sums, the admission votes -- except the cross-device memory path itself (that is what rcn_hip_dp_init's known-answer
vote checks on a multi-GPU node).

cases:
  default   explicit handle exchange (rcn_hip_dp_p2p_export / _attach / _selftest), two epochs of the sharded loop
  admit     the SAME admission procedure rcn_hip_dp_init runs (rcn_hip_dp_p2p_admit, votes after every stage) over gloo; the
            environment's RCN_HIP_DP_FAULT makes one rank fail a stage; whatever form is admitted, two epochs are trained -- on the
            library's loop when a peer exchange was admitted, else by all-reducing rcn_hip_batch_gradient_dev's buffer through gloo
  sticky    rank 1 leaves after one step; rank 0's second step waits for data that never comes (20 ms timeout) and every
            "the work is done" entry point must say so
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, dtype, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    case_name = sys.argv[6] if len(sys.argv) > 6 else "default"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # ranks share one GPU here: were the resident one-XCD kernel chosen, every rank's workers must sit on an XCD of their own (two
    # resident kernels cannot share compute units: each workgroup takes a whole CU's LDS).  On a node every rank has its GPU and XCD 0.
    os.environ.setdefault("RCN_HIP_XCD_SELECT", str(8 + rank % 8))
    import torch
    import torch.distributed as dist
    import mercer_research_amd as amd
    if os.environ.get("RCN_TEST_LIB"):                      # (diagnostic: tools/dp_probe.py runs a case on a variant build of the library)
        from mercer_research_amd import _lib as _l
        _l.LIB_PATH = os.environ["RCN_TEST_LIB"]
    from mercer_research_amd.device import DeviceRCN
    from oracle.rcn_oracle import synthetic_params     # data generator only
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    case = np.load(os.path.join(outdir, "case.npz"))
    dims, Bs, nb = [int(v) for v in case["dims"]], int(case["Bs"]), int(case["nb"])
    ws, bs = synthetic_params(dims, seed=int(case["seed"]))
    ws = [w * 0.1 for w in ws]
    d = DeviceRCN(feedforward_cfg=dims[1:-1], classes=dims[-1], dtype=dtype)
    # Ranks that SHARE this box's one GPU: every rank's resident workers on an XCD of their own, and the kernel asking for exactly the
    # LDS it uses -- at a shard of <= 64 samples that is under half a CU's, so a rank's idle blocks (same launch, same LDS request)
    # fit beside a peer's workers instead of queueing behind them.  With a GPU per rank neither option is needed.
    # (8 + r: the blocks that landed on PHYSICAL XCD r.  Round 3 selected the residue class blockIdx % 8 == rank, and the record of an expired
    # wait showed why three / four ranks sometimes failed: which XCD a dispatch starts its round-robin on differs from queue to queue, so two
    # ranks' classes landed on ONE XCD -- 56 workgroups that each need a CU of their own on 32 CUs: profiles/r4_dp_4rank_timeout_record.txt)
    d.set_option("xcd_select", 8 + rank % 8)
    d.set_option("xcd_exact_lds", 1)
    d.set_params(ws, bs)
    X, Y = d.to_device(case[f"X{rank}"], d.tdtype), d.to_device(case[f"Y{rank}"], d.tdtype)
    loss = d.empty(nb)

    if case_name == "sticky":
        mode = d.dp_p2p_admit()
        assert mode != 0, "the peer exchange was not admitted"
        resident = d.dp_resident(Bs)
        if resident:                # no healthy step first: two resident kernels on ONE device need launch luck (see below); one does not
            d.epoch_begin(X, Y, d.to_device(np.arange(Bs * nb, dtype=np.int32)), Bs, nb)
        else:
            d.dp_train_epoch(X, Y, None, Bs, 1, 3.0, None)         # one healthy step on both ranks
        d.synchronize()
        dist.barrier()
        said = {}
        if rank == 0:
            print(f"RESIDENT {int(resident)}", flush=True)
            d.set_option("dp_timeout_ticks", 2000000)              # 20 ms of the 100 MHz clock, from the next call on
            if resident:
                d.dp_epoch_steps(0, 1, 3.0, None)                  # rank 1 never runs this step
            else:
                d.dp_train_epoch(X, Y, None, Bs, 1, 3.0, None)
            for name, fn in (("synchronize", d.synchronize), ("get_params", d.get_params), ("finalize", d.dp_finalize)):
                try:
                    fn()
                    said[name] = "ok"
                except amd.RcnHipError as e:
                    said[name] = "error"
                    assert "timed out" in str(e) or "expired" in str(e), str(e)
            print("STICKY " + " ".join(f"{k}={v}" for k, v in said.items()), flush=True)
        dist.barrier()                                             # rank 1 keeps its buffers mapped until rank 0 has given up
        if rank != 0:
            d.dp_finalize()
        d.rcn.close()
        dist.destroy_process_group()
        return

    if case_name == "admit":
        bad = timed_out = 0
        mode = d.dp_p2p_admit()
        modes = [None] * world
        dist.all_gather_object(modes, mode)
        assert len(set(modes)) == 1, f"ranks disagree on the admitted form: {modes}"
    else:
        bad, timed_out = d.dp_p2p_setup(selftest_iters=12)
        mode = d.dp_p2p_mode()

    resident = int(d.dp_resident(Bs))
    if mode != 0 and resident:
        # Ranks that SHARE a GPU: a resident kernel holds its XCD's compute units for a whole call, so no rank may have a kernel that
        # needs a place on every XCD (the batch gather) queued while a peer's resident kernel waits for it -- pack first, meet, then
        # step (rcn_hip_epoch_begin_dev + rcn_hip_dp_epoch_steps_dev), and let every kernel of a call drain before the next call.
        # With a GPU per rank none of this matters and rcn_hip_dp_train_epoch_dev does both.
        order = d.to_device(np.arange(Bs * nb, dtype=np.int32))
        if os.environ.get("RCN_TEST_L2_SWEEP") == "1":         # (diagnostic, tools/dp_probe.py: stream 256 MB through every XCD's L2 before the first exchange)
            with torch.cuda.stream(d.stream):
                big = torch.ones(64 << 20, dtype=torch.float32, device=d.device)
                chk = (big * 2.0).sum()
            d.synchronize()
            del big, chk
        try:
            for ep, (first, n, ls) in enumerate(((0, nb, loss), (0, 2, None), (2, nb - 2, None))):   # the second epoch in two calls
                if first == 0:
                    d.epoch_begin(X, Y, order, Bs, nb)
                d.synchronize()
                dist.barrier()
                d.dp_epoch_steps(first, n, 3.0, ls)
                d.synchronize()
                dist.barrier()
        except amd.RcnHipError:
            # an expired wait names its site (rcn_hip_last_timeout): kept beside the outputs so that the test -- and whoever reads a
            # first multi-GPU run -- sees which wait, which worker, which peer, and where every worker of this rank sat
            import json, time
            rec = d.last_timeout()
            rec["host_time_of_failure"] = round(time.time(), 3)   # (two ranks failing ~one time-out apart = their kernels ran one after the other)
            with open(os.path.join(outdir, f"timeout{rank}.json"), "w") as f:
                json.dump(rec, f)
            print(f"TIMEOUT rank {rank}: {json.dumps(rec)}", flush=True)
            raise
    elif mode != 0:
        d.dp_train_epoch(X, Y, None, Bs, nb, 3.0, loss)
        d.dp_train_epoch(X, Y, None, Bs, nb, 3.0, None)           # a second call: sequence numbers carry over
    else:
        # no exchange admitted: the data-parallel halves of train_batch with the caller's own all-reduce (here: gloo via the host)
        grad = d.empty(d.P)
        ls = d.empty(1)
        for ep in range(2):
            for j in range(nb):
                d.batch_gradient(X[j * Bs:(j + 1) * Bs], Y[j * Bs:(j + 1) * Bs], grad, ls)
                d.synchronize()
                g, l = grad.cpu(), ls.cpu()
                dist.all_reduce(g)
                dist.all_reduce(l)
                with torch.cuda.stream(d.stream):
                    grad.copy_(g.to(d.device))
                    if ep == 0:
                        loss[j] = float(l.item()) / (2.0 * Bs * world)
                d.apply_gradient(grad, 3.0 / (Bs * world))
    gw, gb = d.get_params()
    d.synchronize()
    phase = np.zeros(8)
    if mode != 0 and resident and Bs in (128, 256) and case_name == "default":
        # the diagnostic a first multi-GPU run would use: one more call with per-worker phase clocks (rcn_hip_dp_phase_us); the results above
        # are already taken, the parameters move on by these steps on every rank alike
        d.set_option("xcd_dp_phase", 1)
        d.epoch_begin(X, Y, d.to_device(np.arange(Bs * nb, dtype=np.int32)), Bs, nb)
        d.synchronize()
        dist.barrier()
        d.dp_epoch_steps(0, nb, 3.0, None)
        d.synchronize()
        ph = d.dp_phase_us()
        phase = np.array([ph[k] for k in ("owner_wait_mean", "owner_wait_max", "member_wait_mean", "member_wait_max", "tail_all_to_all_mean", "tail_all_to_all_max", "step", "steps")])
        print(f"PHASE rank {rank}: {ph}", flush=True)
        d.set_option("xcd_dp_phase", 0)
        dist.barrier()
    np.savez(os.path.join(outdir, f"out{rank}.npz"), bad=bad, timed_out=timed_out, active=mode, resident=resident, loss=loss.cpu().numpy(), phase=phase,
             **{f"w{i}": w for i, w in enumerate(gw)}, **{f"b{i}": b for i, b in enumerate(gb)})
    dist.barrier()                                             # nobody unmaps while a peer may still read
    d.dp_finalize()
    d.rcn.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
