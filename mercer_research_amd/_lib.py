"""ctypes loader for librcn_hip.so.  There is no fallback: if the HIP library is missing or a call fails,
this raises -- the product path never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librcn_hip.so")
LIB_EXP_PATH = os.path.join(HERE, "librcn_hip_exp.so")      # + the parked experiments (tests / tools only; build.py: build_experiments)

F32, F64 = 0, 1


class RcnHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"rcn_hip status {status}: {message}")
        self.status = status


class RcnPanic(RcnHipError, ValueError):
    """The call hit a condition on which the Rust reference panics (shape / unsupported pooling)."""


class Layer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("arg", C.c_int32)]


class Cfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("dtype", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("n_convpool", C.c_int32), ("convpool", C.POINTER(Layer)), ("n_hidden", C.c_int32), ("hidden", C.POINTER(C.c_int32)),
                ("classes", C.c_int32), ("stream", C.c_void_p)]


_vp, _dp, _sz, _i, _d = C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_int, C.c_double
_u8p, _i32p, _i64p, _ip = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int)

#: every symbol include/rcn_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rcn_hip_abi_version": (_i, []),
    "rcn_hip_status_string": (C.c_char_p, [_i]),
    "rcn_hip_create": (_i, [C.POINTER(Cfg), C.POINTER(_vp)]),
    "rcn_hip_destroy": (None, [_vp]),
    "rcn_hip_last_error": (C.c_char_p, [_vp]),
    "rcn_hip_set_stream": (_i, [_vp, _vp]),
    "rcn_hip_synchronize": (_i, [_vp]),
    "rcn_hip_feature_len": (_i, [_vp, _i64p]),
    "rcn_hip_num_layers": (_i, [_vp]),
    "rcn_hip_layer_dims": (_i, [_vp, _i, _i32p, _i32p]),
    "rcn_hip_param_count": (_i, [_vp, _i64p]),
    "rcn_hip_set_params": (_i, [_vp, _i, _dp, _dp]),
    "rcn_hip_get_params": (_i, [_vp, _i, _dp, _dp]),
    "rcn_hip_init_params": (_i, [_vp, C.c_uint64]),
    "rcn_hip_params_dev": (_i, [_vp, C.POINTER(_vp), _i64p]),
    "rcn_hip_conv_out_shape": (_i, [_i, _i, _i, _i, _i, _ip, _ip]),
    "rcn_hip_pool_out_shape": (_i, [_i, _i, _i, _ip, _ip]),
    "rcn_hip_convolve_2d": (_i, [_vp, _dp, _i, _i, _i, _dp, _i, _i, _i, _dp]),
    "rcn_hip_convolve_2d_separated": (_i, [_vp, _dp, _i, _i, _i, _i, _i, _dp]),
    "rcn_hip_relu": (_i, [_vp, _dp, _sz, _dp]),
    "rcn_hip_pool_2d": (_i, [_vp, _dp, _i, _i, _i, _i, _i, _dp]),
    "rcn_hip_features": (_i, [_vp, _u8p, _sz, _dp]),
    "rcn_hip_features_dev": (_i, [_vp, _vp, _sz, _vp, _i]),
    "rcn_hip_gen_scales": (_i, [_vp, _dp, _sz, _dp, _dp]),
    "rcn_hip_gen_scales_dev": (_i, [_vp, _vp, _sz, _dp, _dp]),
    "rcn_hip_set_scale": (_i, [_vp, _d, _d]),
    "rcn_hip_get_scale": (_i, [_vp, _dp, _dp]),
    "rcn_hip_standardize": (_i, [_vp, _dp, _sz]),
    "rcn_hip_standardize_dev": (_i, [_vp, _vp, _sz]),
    "rcn_hip_train_batch": (_i, [_vp, _dp, _dp, _sz, _d, _dp]),
    "rcn_hip_train_batch_dev": (_i, [_vp, _vp, _vp, _sz, _d, _vp]),
    "rcn_hip_train_epoch_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _d, _vp]),
    "rcn_hip_prepare_epoch_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _d, _vp]),
    "rcn_hip_train_epoch_images_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _d, _vp]),
    "rcn_hip_prepare_epoch_images_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _d, _vp]),
    "rcn_hip_epoch_begin_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz]),
    "rcn_hip_epoch_begin_images_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz]),
    "rcn_hip_epoch_steps_dev": (_i, [_vp, _sz, _sz, _d, _vp]),
    "rcn_hip_prepare_epoch_steps_dev": (_i, [_vp, _sz, _sz, _d, _vp]),
    "rcn_hip_shuffle_dev": (_i, [_vp, _vp, _sz, _sz, C.c_uint64]),
    "rcn_hip_batch_gradient_dev": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "rcn_hip_batch_gradient_perm_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "rcn_hip_apply_gradient_dev": (_i, [_vp, _vp, _d]),
    "rcn_hip_dp_unique_id": (_i, [_vp]),
    "rcn_hip_dp_init": (_i, [_vp, _vp, _i, _i]),
    "rcn_hip_dp_finalize": (_i, [_vp]),
    "rcn_hip_dp_world": (_i, [_vp]),
    "rcn_hip_dp_rank": (_i, [_vp]),
    "rcn_hip_dp_broadcast_params": (_i, [_vp, _i]),
    "rcn_hip_dp_p2p_export": (_i, [_vp, _vp]),
    "rcn_hip_dp_p2p_attach": (_i, [_vp, _vp, _i, _i]),
    "rcn_hip_dp_p2p_selftest": (_i, [_vp, _i, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "rcn_hip_dp_p2p_admit": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "rcn_hip_dp_admission_rehearse": (_i, [_i, _i, C.c_char_p, _vp, _vp, _vp, _ip, _ip]),
    "rcn_hip_dp_epoch_steps_dev": (_i, [_vp, _sz, _sz, _d, _vp]),
    "rcn_hip_train_epoch_gathers": (_i, [_vp, _sz]),
    "rcn_hip_train_epoch_resident": (_i, [_vp, _sz]),
    "rcn_hip_dp_resident": (_i, [_vp, _sz]),
    "rcn_hip_dp_phase_us": (_i, [_vp, _dp, _sz]),
    "rcn_hip_dp_p2p_active": (_i, [_vp]),
    "rcn_hip_dp_prepare_epoch_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _d, _vp]),
    "rcn_hip_dp_train_epoch_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _d, _vp]),
    "rcn_hip_load_data": (_i, [_vp, _i, _u8p, _i32p, _sz, _dp, _dp]),
    "rcn_hip_train_set_epoch": (_i, [_vp, _i, _i32p, C.c_uint64, _sz, _d, _dp]),
    "rcn_hip_evaluate_set": (_i, [_vp, _i, _i64p]),
    "rcn_hip_set_size": (_i, [_vp, _i, _i64p]),
    "rcn_hip_forward": (_i, [_vp, _dp, _sz, _dp]),
    "rcn_hip_forward_dev": (_i, [_vp, _vp, _sz, _vp]),
    "rcn_hip_classify": (_i, [_vp, _dp, _sz, _i32p]),
    "rcn_hip_evaluate": (_i, [_vp, _dp, _dp, _sz, _i64p]),
    "rcn_hip_evaluate_dev": (_i, [_vp, _vp, _vp, _sz, _i64p]),
    "rcn_hip_classify_images": (_i, [_vp, _u8p, _sz, _i32p]),
    "rcn_hip_set_dense_path": (_i, [_vp, _i]),
    "rcn_hip_fallbacks_taken": (_i, [_vp]),
    "rcn_hip_last_timeout": (_i, [_vp, C.POINTER(C.c_uint32), _sz]),
    "rcn_hip_last_timeout_text": (C.c_char_p, [_vp]),
    "rcn_hip_set_option": (_i, [_vp, C.c_char_p, C.c_int64]),
    "rcn_hip_get_option": (_i, [_vp, C.c_char_p, _i64p]),
    "rcn_hip_set_feature_kernel": (_i, [_vp, _i]),
    "rcn_hip_time_kernels_dev": (_i, [_vp, _vp, _vp, _sz, _i, _dp, _dp, _dp]),
}

FALLBACKS_SEEN = 0          # step-downs from the resident kernel counted over every context this process has closed (rcn.py: RCN.close)
_lib = None
_lib_exp = None
_hip_preloaded = False


def preload_hip_runtime() -> None:
    """One HIP runtime per process.  librcn_hip.so asks the loader for libamdhip64 by SONAME; PyTorch ships its own copy
    under the same SONAME.  Whichever is mapped first wins for both, and a process that mapped the system copy first
    (ctypes-only use of this package) and imports torch later ends up with a runtime torch cannot initialise
    (`torch.cuda.is_available()` turns False).  So: when torch is installed, map ITS copy before ours -- without
    importing torch -- and the order of use no longer matters.  Without torch the system copy is used."""
    global _hip_preloaded
    if _hip_preloaded:
        return
    _hip_preloaded = True
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass



def load(path: str | None = None) -> C.CDLL:
    """dlopen the library and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(f"{p} not found: build it with `python -m mercer_research_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    preload_hip_runtime()
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.rcn_hip_abi_version() != 1:
        raise ImportError("librcn_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def load_experiments() -> C.CDLL:
    """The build that also carries the parked experiments (dense paths 3 and 4).  Not used by the product path."""
    global _lib_exp
    if _lib_exp is None:
        _lib_exp = load(LIB_EXP_PATH)
    return _lib_exp


def check(lib, ctx, status: int) -> None:
    if status == 0:
        return
    msg = lib.rcn_hip_last_error(ctx).decode() if ctx else ""
    if not msg:
        msg = lib.rcn_hip_status_string(status).decode()
    raise (RcnPanic if status in (-2, -3) else RcnHipError)(status, msg)
