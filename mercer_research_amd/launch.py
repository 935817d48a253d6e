"""One process per GPU, started from a parent that never touches the GPU.

`python bench.py --gpus N` (no launcher in front of it) lands here: the parent spawns N fresh children of the same script
-- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly what `torch.distributed.run`
would set -- relays rank 0's stdout (the one JSON line) to its own stdout, sends every other rank's stdout to stderr, and
returns the worst return code.  Nothing here imports torch or makes a HIP call: a process that has initialised the GPU
must never be replaced by another program, so the children are started with subprocess.Popen (fork + exec of a process
that never opened the device) and the parent only waits.  Pure stdlib.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import List, Optional, Sequence


def under_launcher() -> bool:
    """True inside a rank started by torch.distributed.run or by spawn_ranks (the rendezvous variables are set)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script: str, argv: Sequence[str], nproc: int, timeout_s: Optional[float] = None, extra_env: Optional[dict] = None) -> int:
    """Run `script argv...` as `nproc` ranks on this node; returns the worst child return code (124 on timeout)."""
    if nproc < 1:
        raise ValueError("spawn_ranks: nproc must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    relays: List[threading.Thread] = []

    def relay(stream, sink):
        for line in iter(stream.readline, b""):
            sink.write(line)
            sink.flush()
        stream.close()

    for r in range(nproc):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL and hipIpc need on this driver stack
        if extra_env:
            env.update(extra_env)
        p = subprocess.Popen([sys.executable, script, *argv], env=env, stdout=subprocess.PIPE, stderr=None, start_new_session=False)
        procs.append(p)
        # rank 0's stdout is the job's stdout (one JSON line); the other ranks' stdout is diagnostics
        t = threading.Thread(target=relay, args=(p.stdout, sys.stdout.buffer if r == 0 else sys.stderr.buffer), daemon=True)
        t.start()
        relays.append(t)

    def stop_all(sig):
        for q in procs:
            if q.poll() is None:
                try:
                    q.send_signal(sig)          # the exact PIDs started above, nothing by pattern
                except ProcessLookupError:
                    pass

    deadline = time.monotonic() + timeout_s if timeout_s else None
    worst = 0
    failed_at = None
    try:
        while True:
            alive = 0
            for q in procs:
                rc = q.poll()
                if rc is None:
                    alive += 1
                elif rc != 0 and failed_at is None:
                    failed_at = time.monotonic()         # a rank died: its peers would wait in a collective for ever
            if alive == 0:
                break
            now = time.monotonic()
            if failed_at is not None and now - failed_at > 15.0:
                stop_all(signal.SIGTERM)
                failed_at = now + 1e9
                deadline = now + 10.0
            if deadline is not None and now > deadline:
                stop_all(signal.SIGKILL)
                worst = 124
                deadline = None
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all(signal.SIGTERM)
        worst = 130
    for q in procs:
        rc = q.wait()
        rc = 128 - rc if rc < 0 else rc
        worst = max(worst, rc)
    for t in relays:
        t.join(timeout=5.0)
    return worst
