"""mercer_research_amd -- MI355X-native (gfx950 HIP) implementation of the `rcn` crate's training hot path.

Package layout (only what the path needs):
  csrc/        HIP kernels + the C-ABI (include/rcn_hip.h) -> librcn_hip.so
  rcn.py       host mirror of the reference's Rust API (RCN, RCNLayer, Padding, Pooling, Convolve2D/Pool2D ops)
  device.py    device-resident driver (torch tensors as HBM buffers, one HIP stream)
  dp.py        data-parallel step: shard gradients -> RCCL all-reduce -> identical update on every rank
"""
from ._lib import F32, F64, RcnHipError, RcnPanic  # noqa: F401
from .rcn import (RCN, Padding, Pooling, RCNLayer, SeparableOperator, convolve_2d, convolve_2d_separated,  # noqa: F401
                  default_convpool, pool_2d, relu)
