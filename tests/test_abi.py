"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/rcn_hip.h declares,
its pure-host shape helpers agree with the oracle, and it fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mercer_research_amd import build as hipbuild, _lib
    hipbuild.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rcn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rcn_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from mercer_research_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 40
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/rcn_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "python binding table and header disagree"
    assert lib.rcn_hip_abi_version() == 1


def test_no_torch_or_cxx_types_in_header():
    text = open(os.path.join(ROOT, "include", "rcn_hip.h")).read()
    assert "torch" not in text.lower().replace("no c++ / torch types", "") and "std::" not in text and "at::" not in text
    # it must compile as plain C
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "rcn_hip.h")], check=True)


def test_library_does_not_link_the_oracle():
    from mercer_research_amd import _lib
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rcn_oracle" not in out and "torch" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "mercer_research_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".ipp", ".cpp", ".h")):
                assert "rcn_oracle" not in open(os.path.join(root, f)).read(), f"{f} references the oracle"


def test_shape_helpers_match_oracle(lib, oracle):
    oR, oC = C.c_int(), C.c_int()
    for R in range(1, 9):
        for Cc in range(1, 9):
            for kr, kc in ((1, 1), (3, 1), (1, 3), (3, 3), (2, 2), (5, 5), (5, 1), (1, 5), (2, 3)):
                for pad in (0, 1):
                    st = lib.rcn_hip_conv_out_shape(R, Cc, kr, kc, pad, C.byref(oR), C.byref(oC))
                    r2, c2 = C.c_int(), C.c_int()
                    so = oracle.lib.rcn_o_conv_out_shape(R, Cc, kr, kc, pad, C.byref(r2), C.byref(c2))
                    assert (st == 0) == (so == 0), (R, Cc, kr, kc, pad)
                    if st == 0:
                        assert (oR.value, oC.value) == (r2.value, c2.value)
            for pad in (0, 1):
                st = lib.rcn_hip_pool_out_shape(R, Cc, pad, C.byref(oR), C.byref(oC))
                r2, c2 = C.c_int(), C.c_int()
                so = oracle.lib.rcn_o_pool_out_shape(R, Cc, pad, C.byref(r2), C.byref(c2))
                assert (st == 0) == (so == 0)
                if st == 0:
                    assert (oR.value, oC.value) == (r2.value, c2.value)
    assert lib.rcn_hip_conv_out_shape(8, 8, 3, 3, 7, C.byref(oR), C.byref(oC)) == -1     # bad enum -> invalid argument


def test_status_strings(lib):
    for s in range(0, -8, -1):
        assert lib.rcn_hip_status_string(s)
    assert b"unknown" in lib.rcn_hip_status_string(-99)


def test_fails_loudly_without_a_gpu(lib):
    """No GPU in this container: context creation must report RCN_HIP_ERR_NO_DEVICE (or succeed on a GPU box), never
    silently compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import mercer_research_amd as amd
    with pytest.raises(amd.RcnHipError) as e:
        amd.RCN(10, amd.default_convpool(), [30])
    assert e.value.status == -5
    with pytest.raises(amd.RcnHipError):
        amd.convolve_2d(np.ones((4, 4)), np.ones((3, 3)), amd.Padding.SAME)
    from mercer_research_amd.device import DeviceRCN
    with pytest.raises(RuntimeError):
        DeviceRCN()


def test_invalid_cfg_is_rejected(lib):
    from mercer_research_amd import _lib
    ctx = C.c_void_p()
    assert lib.rcn_hip_create(None, C.byref(ctx)) == -1
    cfg = _lib.Cfg()
    cfg.struct_size = 3
    assert lib.rcn_hip_create(C.byref(cfg), C.byref(ctx)) == -1
    assert lib.rcn_hip_last_error(None) == b"null context"
    lib.rcn_hip_destroy(None)        # no-op


def test_cpp_host_mirror_compiles_and_links(lib, tmp_path):
    """mercer_research_amd/csrc/host/rcn.hpp (C++ mirror of the Rust RCN API) builds against the C ABI with plain g++."""
    from mercer_research_amd import _lib
    exe = tmp_path / "host_demo"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "host_demo.cpp"),
                    "-L" + os.path.dirname(_lib.LIB_PATH), "-lrcn_hip", "-lz", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-o", str(exe)], check=True)
    assert exe.exists()


def test_rust_shim_declares_exactly_the_header_functions():
    """rust/rcn-hip-sys cannot be compiled here (no cargo), so at least its extern block is held in lock-step with the header:
    the same set of rcn_hip_* functions as include/rcn_hip.h (which test_library_exports_every_declared_symbol ties to the
    library and to the ctypes table)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "rcn_hip.h")).read()
    rust = open(os.path.join(root, "rust", "rcn-hip-sys", "src", "lib.rs")).read()
    in_header = set(re.findall(r"\b(rcn_hip_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    in_rust = set(re.findall(r"pub fn (rcn_hip_[a-z0-9_]+)\s*\(", rust))
    assert in_header - in_rust == set(), f"missing in the Rust shim: {sorted(in_header - in_rust)}"
    assert in_rust - in_header == set(), f"not in the header: {sorted(in_rust - in_header)}"


def test_parked_experiments_are_not_in_the_shipping_library(lib):
    """the resident epoch kernel (dense_p2_persist.hpp), the one-launch step (dense_p2_step.hpp) and the
    one-object step kernel (k_p2_ab) live only in librcn_hip_exp.so, which the product never loads."""
    from mercer_research_amd import _lib
    names = (b"k_p2_step", b"k_p2_epoch", b"k_p2_ab")
    ship = open(_lib.LIB_PATH, "rb").read()
    exp = open(_lib.LIB_EXP_PATH, "rb").read()
    for n in names:
        assert n not in ship, n
        assert n in exp, n
    for root, _, files in os.walk(os.path.join(ROOT, "mercer_research_amd")):
        for f in files:
            if f.endswith(".py") and f not in ("_lib.py", "build.py", "rcn.py", "device.py"):
                assert "load_experiments" not in open(os.path.join(root, f)).read(), f
    for f in ("bench.py", "bench_convnet.py", "__graft_entry__.py"):
        assert "experiments" not in open(os.path.join(ROOT, f)).read(), f


def test_trackx_library_exports_its_header_and_binding_table():
    """include/rcn_hipx.h (Track X: the trainable convolution net, no reference counterpart) against librcn_hipx.so and the ctypes
    table of mercer_research_amd/convnet.py; the header compiles as plain C; the library does not link the oracle or torch."""
    from mercer_research_amd import build as hipbuild, convnet
    hipbuild.build_x()
    text = open(os.path.join(ROOT, "include", "rcn_hipx.h")).read()
    declared = sorted(set(re.findall(r"\b(rcn_hipx_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S))))
    assert len(declared) >= 18
    raw = C.CDLL(convnet.LIBX_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/rcn_hipx.h but not exported"
    assert sorted(convnet.SIGNATURES) == declared, "python binding table and header disagree"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "rcn_hipx.h")], check=True)
    out = subprocess.run(["ldd", convnet.LIBX_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "torch" not in out
