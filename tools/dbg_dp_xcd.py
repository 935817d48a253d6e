#!/usr/bin/env python3
"""Diagnostic: do the blocks of two PROCESSES' kernels keep the static block -> XCD mapping when they run at the same time on one GPU?
usage: dbg_dp_xcd.py            (parent: spawns two children)
       dbg_dp_xcd.py <rank> <port> <mode>   mode: staggered | concurrent"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(rank, port, mode):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["RCN_HIP_XCD_SELECT"] = str(rank)
    import torch.distributed as dist
    import mercer_research_amd as amd
    from mercer_research_amd.device import DeviceRCN
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    d = DeviceRCN(dtype=0)
    for turn in range(2):
        if mode == "staggered":
            dist.barrier()
            if turn != rank:
                continue
        elif turn == 1:
            break
        else:
            dist.barrier()
        try:
            d.set_dense_path(5)
            print(f"rank {rank} {mode}: probe ok", flush=True)
        except amd.RcnHipError as e:
            print(f"rank {rank} {mode}: probe FAILED: {e}", flush=True)
    dist.barrier()
    d.rcn.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
    else:
        from mercer_research_amd.launch import free_port
        for mode in ("staggered", "concurrent"):
            port = free_port()
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(port), mode]) for r in range(2)]
            for p in ps:
                p.wait(timeout=200)
