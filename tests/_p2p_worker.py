"""Worker of test_gpu_parity.test_peer_allreduce_two_processes_one_gpu: one data-parallel rank.  Run as
python tests/_p2p_worker.py <rank> <world> <port> <dtype> <outdir>; ranks share GPU 0 (the dev box has one), which
exercises the whole protocol -- hipIpc export/attach, flags, double buffering, rank-order sums -- except the
cross-device memory path itself (that is what rcn_hip_dp_init's known-answer vote checks on a multi-GPU node)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, dtype, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from mercer_research_amd.device import DeviceRCN
    from oracle.rcn_oracle import synthetic_params     # data generator only
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    case = np.load(os.path.join(outdir, "case.npz"))
    dims, Bs, nb = [int(v) for v in case["dims"]], int(case["Bs"]), int(case["nb"])
    ws, bs = synthetic_params(dims, seed=int(case["seed"]))
    ws = [w * 0.1 for w in ws]
    d = DeviceRCN(feedforward_cfg=dims[1:-1], classes=dims[-1], dtype=dtype)
    d.set_params(ws, bs)
    bad, timed_out = d.dp_p2p_setup(selftest_iters=12)
    X, Y = d.to_device(case[f"X{rank}"], d.tdtype), d.to_device(case[f"Y{rank}"], d.tdtype)
    loss = d.empty(nb)
    d.dp_train_epoch(X, Y, None, Bs, nb, 3.0, loss)
    d.dp_train_epoch(X, Y, None, Bs, nb, 3.0, None)           # a second call: sequence numbers carry over
    gw, gb = d.get_params()
    d.synchronize()
    np.savez(os.path.join(outdir, f"out{rank}.npz"), bad=bad, timed_out=timed_out, active=d.dp_p2p_mode(), loss=loss.cpu().numpy(),
             **{f"w{i}": w for i, w in enumerate(gw)}, **{f"b{i}": b for i, b in enumerate(gb)})
    dist.barrier()                                             # nobody unmaps while a peer may still read
    d.dp_finalize()
    d.rcn.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
