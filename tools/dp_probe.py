#!/usr/bin/env python3
"""Diagnostic: the two-process data-parallel case of tests/test_gpu_round4.py at a given shard, run several times, optionally on a variant build
of the library (mercer_research_amd/variants/librcn_hip_<name>.so), printing per run whether every rank finished and, if not, each rank's
time-out record.    python tools/dp_probe.py [--shard 256] [--world 2] [--runs 3] [--variant name]"""
import argparse, json, os, socket, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.rcn_oracle import one_hot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shard", type=int, default=256)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--variant", default="")
    ap.add_argument("--ticks", default="400000000")
    ap.add_argument("--sweep", type=int, default=0, help="1: every rank streams 256 MB through the device's L2s before the first exchange")
    a = ap.parse_args()
    dims, nb = [784, 30, 10], 3
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RCN_HIP_XCD_TIMEOUT_TICKS=a.ticks, RCN_HIP_DP_TIMEOUT_TICKS=a.ticks)
    if a.sweep:
        env["RCN_TEST_L2_SWEEP"] = "1"
    if a.variant:
        env["RCN_TEST_LIB"] = os.path.join(ROOT, "mercer_research_amd", "variants", f"librcn_hip_{a.variant}.so")
    for run in range(a.runs):
        with tempfile.TemporaryDirectory() as td:
            rng = np.random.default_rng(31 + run)
            case = {f"X{r}": np.maximum(rng.standard_normal((a.shard * nb, dims[0])), 0.0) for r in range(a.world)}
            case.update({f"Y{r}": one_hot(rng.integers(0, 10, a.shard * nb), 10) for r in range(a.world)})
            np.savez(os.path.join(td, "case.npz"), dims=dims, Bs=a.shard, nb=nb, seed=17, **case)
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_p2p_worker.py"), str(r), str(a.world), str(port), "0", td, "default"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(a.world)]
            outs = []
            for p in procs:
                try:
                    o, _ = p.communicate(timeout=120)
                except subprocess.TimeoutExpired:
                    p.kill(); o, _ = p.communicate()
                outs.append(o.decode(errors="replace"))
            ok = all(p.returncode == 0 for p in procs)
            recs = []
            for r in range(a.world):
                f = os.path.join(td, f"timeout{r}.json")
                if os.path.exists(f):
                    j = json.load(open(f)); recs.append((r, {k: j.get(k) for k in ("site", "worker", "xcc", "host_time_of_failure")}, j.get("text", "").split("; xcc =")[0][-150:]))
            phases = [l for o in outs for l in o.splitlines() if l.startswith("PHASE")]
            print(f"run {run} sweep {a.sweep} variant '{a.variant or 'default'}' shard {a.shard} world {a.world}: {'ok' if ok else 'FAILED'} {recs if recs else ''} {phases}", flush=True)


if __name__ == "__main__":
    main()
