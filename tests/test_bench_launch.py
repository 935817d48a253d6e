"""`python bench.py --gpus N` must start its own N ranks: the parent spawns the children before any GPU
call, relays rank 0's single JSON line and returns the worst return code.  Rehearsed on the CPU with --dry-launch (gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, script), *args], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_n_dry_launch_rendezvous_and_one_line(n):
    r = _run("bench.py", "--gpus", str(n), "--dry-launch", "--steps", "20", "--warmup", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one line on the job's stdout: rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["dry_launch"] is True and line["rendezvous_check"] is True
    assert line["config"]["global_batch"] == 256 * n and line["config"]["parallelism"] == f"dp{n}"
    assert line["scaling"] == "weak" and line["steps"] == 20 and line["warmup"] == 5


def test_bench_under_an_external_launcher_does_not_spawn_again():
    # the driver's other form: torch.distributed.run in front; the script must take the rank it is given
    r = _run("bench.py", "--gpus", "2", "--dry-launch", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_spawner_returns_worst_child_code_and_does_not_hang(tmp_path):
    from mercer_research_amd.launch import spawn_ranks
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\nr = int(os.environ['RANK'])\nassert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                     "print('line from', r, flush=True)\nsys.exit(7 if r == 1 else 0)\n")
    drv = tmp_path / "drv.py"
    drv.write_text(f"import sys\nsys.path.insert(0, {ROOT!r})\nfrom mercer_research_amd.launch import spawn_ranks\nsys.exit(spawn_ranks({str(child)!r}, [], 3, timeout_s=60))\n")
    r = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7
    assert r.stdout.strip() == "line from 0"              # rank 0 -> stdout, everything else -> stderr
    assert "line from 1" in r.stderr and "line from 2" in r.stderr


def test_spawning_parent_never_imports_torch_or_loads_hip():
    # a process that has initialised the GPU must not be the one that starts other programs: the parent's imports stay stdlib + numpy
    code = ("import sys; sys.argv=['bench.py','--gpus','2','--dry-launch']\n"
            "import runpy, mercer_research_amd.launch as L\n"
            "L.spawn_ranks = lambda *a, **k: (print('SPAWN', 'torch' in sys.modules, any('amdhip' in (getattr(m,'__file__','') or '') for m in sys.modules.values())), 0)[1]\n"
            "runpy.run_path('bench.py', run_name='__main__')\n")
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT, env=e)
    assert "SPAWN False False" in r.stdout, (r.stdout, r.stderr[-1500:])


def test_bench_convnet_spawns_too():
    src = open(os.path.join(ROOT, "bench_convnet.py")).read()
    assert "spawn_ranks" in src and "launch with torch.distributed.run" not in src
