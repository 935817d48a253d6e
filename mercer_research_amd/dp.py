"""Data-parallel train_batch: the sum over samples of rcn.rs:190-205 is split over ranks, combined by ONE
all-reduce of the flat gradient buffer (RCCL over xGMI when the process group's backend is "nccl"), and every rank
applies the identical update W <- W - (eta / B_global) * sum dW, so replicas stay bit-identical without a broadcast.

The engine is duck-typed so that the same step logic runs over the HIP context on GPUs (mercer_research_amd.device
.DeviceRCN) and, in the CPU test-suite, over a stand-in engine with the gloo backend."""
from __future__ import annotations

from typing import Optional, Protocol

import torch
import torch.distributed as dist


class GradientEngine(Protocol):
    def batch_gradient(self, x, y, grad=None, loss_sum=None) -> torch.Tensor: ...
    def apply_gradient(self, grad: torch.Tensor, scale: float) -> None: ...
    def params_flat(self) -> torch.Tensor: ...


def shard_bounds(global_batch: int, world: int, rank: int):
    """Contiguous shard [lo, hi) of a global batch; requires world | global_batch so every rank does equal work."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class DataParallelStep:
    def __init__(self, engine: GradientEngine, group: Optional[dist.ProcessGroup] = None):
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._grad: Optional[torch.Tensor] = None

    def broadcast_params(self, src: int = 0):
        """Make every replica start from rank `src`'s parameters (the reference has one model, rcn.rs:139-141)."""
        if self.world > 1:
            dist.broadcast(self.engine.params_flat(), src=src, group=self.group)

    def train_batch(self, x_shard, y_shard, eta: float, global_batch: int, loss_sum: Optional[torch.Tensor] = None, perm=None):
        """One train_batch over a global batch of which this rank holds `x_shard` (or, with `perm`, the rows
        perm[0..len(perm)) of the resident set x_shard / y_shard)."""
        if perm is not None:
            self._grad = self.engine.batch_gradient_perm(x_shard, y_shard, perm, perm.numel(), self._grad, loss_sum)
        else:
            self._grad = self.engine.batch_gradient(x_shard, y_shard, self._grad, loss_sum)
        if self.world > 1:
            dist.all_reduce(self._grad, op=dist.ReduceOp.SUM, group=self.group)
            if loss_sum is not None:
                dist.all_reduce(loss_sum, op=dist.ReduceOp.SUM, group=self.group)
        self.engine.apply_gradient(self._grad, eta / float(global_batch))       # rcn.rs:214,221 with the GLOBAL batch length
