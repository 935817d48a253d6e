"""Step time of the data-parallel loop on the peer-read all-reduce with W ranks that SHARE this box's one GPU (gloo carries
the hipIpc handles).  Sharing a GPU makes the ranks compete for CUs, so this is an upper bound on the per-step cost of the
protocol, not a scaling measurement.  Usage: python tools/p2p_bench.py [world]   (spawns the ranks itself)"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from mercer_research_amd.device import DeviceRCN
    from mercer_research_amd.synth import synthetic_images, synthetic_params
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    N, B = 16384, 256
    d = DeviceRCN(dtype=0)
    ws, bs = synthetic_params([784, 30, 10], seed=42)
    d.set_params(ws, bs)
    imgs, labels = synthetic_images(N, seed=1234 + rank)
    X, Y = d.load_data(d.to_device(imgs), d.to_device(labels))
    perm = torch.empty(N, dtype=torch.int32, device=d.device)
    if world > 1:
        bad, to = d.dp_p2p_setup(selftest_iters=16)
        assert bad == 0 and to == 0, (bad, to)
    else:
        d.dp_init()
    nb = N // B

    def epoch(i):
        d.shuffle(perm, N, 1, seed=77 + rank * 7919 + i)
        d.dp_train_epoch(X, Y, perm, B, nb, 3.0, None)
    for i in range(4):
        epoch(i)
    d.synchronize()
    dist.barrier()
    reps = 32
    t0 = time.perf_counter()
    for i in range(reps):
        epoch(100 + i)
    d.synchronize()
    el = time.perf_counter() - t0
    dist.barrier()
    if rank == 0:
        print(f"world={world} (one shared GPU) p2p_active={d.dp_p2p_active()}: {el / (reps * nb) * 1e6:.2f} us per step, "
              f"{world * B * reps * nb / el / 1e6:.2f} M images/s aggregate", flush=True)
    d.dp_finalize()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port)]) for r in range(world)]
        rc = 0
        for p in procs:
            try:
                rc |= p.wait(timeout=300)
            except subprocess.TimeoutExpired:
                p.kill()
                rc |= 1
        sys.exit(rc)
