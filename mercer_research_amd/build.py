"""Builds librcn_hip.so (the C-ABI library of include/rcn_hip.h) for gfx950 with hipcc, in-tree.

    python -m mercer_research_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librcn_hip.so")
SOURCES = ["rcn_hip_api.hip"]
DEPS = ["rcn_hip_api_shapes.ipp", "rcn_hip_api_dense_launch.ipp", "rcn_hip_api_xcd.ipp", "rcn_hip_api_p2p.ipp", "rcn_hip_api_params.ipp", "rcn_hip_api_operators.ipp", "rcn_hip_api_features.ipp", "rcn_hip_api_dense.ipp", "rcn_hip_api_dp.ipp", "rcn_hip_api_sets.ipp", "common.hpp", "dense.hpp", "dense_pipe.hpp", "dense_p2.hpp", "features.hpp", "ops.hpp", "dp_rccl.hpp", "serve.hpp", "dp_p2p.hpp", "dense_p2_dp.hpp", "dense_xcd.hpp", "dense_wide.hpp", "dense_p2_persist.hpp", "dense_p2_step.hpp", os.path.join("..", "..", "include", "rcn_hip.h")]
# -ffp-contract=off: the reference (rustc) never fuses a*b+c; the operator kernels reproduce its f64 rounding.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-result",
         "-Wno-pass-failed"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + DEPS)


LIB_EXP = os.path.join(HERE, "librcn_hip_exp.so")   # the same library + the parked experiments (dense paths 3, 4, one-object step); tests / tools only


def build_experiments(force: bool = False, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, f) for f in SOURCES + DEPS]
    if not force and os.path.exists(LIB_EXP) and all(os.path.getmtime(f) <= os.path.getmtime(LIB_EXP) for f in deps):
        return LIB_EXP
    cmd = [hipcc()] + FLAGS + ["-DRCN_HIP_EXPERIMENTS", "-o", LIB_EXP] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_EXP


LIBX = os.path.join(HERE, "librcn_hipx.so")          # Track X (trainable conv net; include/rcn_hipx.h)
LIBX_SRC = os.path.join(CSRC, "rcn_hipx_api.hip")
LIBX_DEPS = [LIBX_SRC, os.path.join(CSRC, "convnet.hpp"), os.path.join(CSRC, "convnet_bf16.hpp"), os.path.join(CSRC, "convnet_halo.hpp"), os.path.join(CSRC, "convnet_halo_bf16.hpp"), os.path.join(HERE, "..", "include", "rcn_hipx.h")]


def build_x(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(LIBX) and all(os.path.getmtime(f) <= os.path.getmtime(LIBX) for f in LIBX_DEPS):
        return LIBX
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-result", "-o", LIBX, LIBX_SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIBX


CLI = os.path.join(HERE, "rcn_hip_cli")
CLI_SRC = os.path.join(CSRC, "host", "rcn_main.cpp")
CLI_DEPS = [CLI_SRC, os.path.join(CSRC, "host", "rcn.hpp"), os.path.join(CSRC, "host", "formats.hpp")]


def build_cli(force: bool = False, verbose: bool = False) -> str:
    """The C++ host side: rcn_hip_cli = rcn/src/main.rs over the C ABI (plain g++, links librcn_hip.so and zlib)."""
    if not force and os.path.exists(CLI) and all(os.path.getmtime(f) <= os.path.getmtime(CLI) for f in CLI_DEPS + [LIB]):
        return CLI
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", CLI_SRC, "-L" + HERE, "-lrcn_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-o", CLI]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return CLI


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    build_cli(force, verbose)
    build_x(force, verbose)
    build_experiments(force, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
