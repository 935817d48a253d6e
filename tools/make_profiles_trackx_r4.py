#!/usr/bin/env python3
"""Copies what tools/prof_trackx_r4.sh left under gpurun_out/ into profiles/r4_trackx_* (run on the dev box after the gpurun call)."""
import glob, io, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
G = "gpurun_out"
for f in sorted(glob.glob(f"{G}/r4_trackx_bench_*.json")):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join("profiles", os.path.basename(f)), "w").write(lines[-1] + "\n")
for m in ("bf16", "bf16_stored"):
    c = max(glob.glob(f"{G}/prof4_trackx_stats_{m}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    shutil.copy(c, f"profiles/r4_trackx_224_{m}_kernel_stats.csv")
out = subprocess.run([sys.executable, "tools/pmc_kernel_table.py", f"{G}/prof4_trackx_sq"], capture_output=True, text=True, env=dict(os.environ, TOP="40")).stdout
open("profiles/r4_trackx_224_bf16_stored_sq_counters.txt", "w").write(
    "# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT\n"
    "#   -- python3 bench_convnet.py --config synth224 --precision bf16_stored --steps 6 --warmup 2      (tools/prof_trackx_r4.sh; per kernel, summed over its dispatches;\n"
    "# every counter as a fraction of SQ_WAVE_CYCLES -- WAIT_ANY: parked at s_waitcnt / a barrier, WAIT_INST_ANY: issue stalls, ACTIVE_INST_ANY: issuing;\n"
    "# SQ_VALU_MFMA_BUSY_CYCLES counts cycles, the others quad-cycles: MI355X_MICROARCH.md)\n" + out)
subprocess.run([sys.executable, "tools/mfma_pmc_summary.py", f"{G}/prof4_trackx", "profiles/r4_trackx_mfma_pmc.json"], check=True)
print(out[:1500])
