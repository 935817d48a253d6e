"""Device-resident driver: torch CUDA(ROCm) tensors are used purely as HBM buffers + stream handles; every
kernel that touches sample data is one of ours, reached through the `_dev` entry points of include/rcn_hip.h."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .rcn import RCN, RCNLayer, default_convpool


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DeviceRCN:
    """An RCN whose data set, parameters and training loop live on one GPU.

    All work is enqueued on `self.stream` (a torch side stream whose HIP handle the context shares), so torch ops
    issued under `with torch.cuda.stream(self.stream)` and our kernels are ordered with each other."""

    def __init__(self, classes: int = 10, convpool_cfg: Optional[Sequence[RCNLayer]] = None, feedforward_cfg: Sequence[int] = (30,),
                 input_shape: Tuple[int, int] = (28, 28), dtype: int = _lib.F32, device: int = 0, experiments: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceRCN needs a GPU; there is no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.rcn = RCN(classes, convpool_cfg if convpool_cfg is not None else default_convpool(), list(feedforward_cfg),
                       input_shape=input_shape, dtype=dtype, device=device, stream=self.stream.cuda_stream, experiments=experiments)
        self.tdtype = torch.float64 if dtype == _lib.F64 else torch.float32
        self.lib, self.ctx = self.rcn._lib, self.rcn._ctx
        self.F, self.classes = self.rcn.feature_len, classes
        self.P = sum(self.rcn.dims[l] * self.rcn.dims[l + 1] + self.rcn.dims[l + 1] for l in range(len(self.rcn.dims) - 1))

    @classmethod
    def adopt(cls, rcn: RCN) -> "DeviceRCN":
        """Device-resident driver around an existing context (its kernels move onto a fresh torch side stream)."""
        self = cls.__new__(cls)
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceRCN needs a GPU; there is no CPU fallback")
        self.device = torch.device("cuda", 0)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.rcn = rcn
        rcn._ck(rcn._lib.rcn_hip_set_stream(rcn._ctx, C.c_void_p(self.stream.cuda_stream)))
        self.tdtype = torch.float64 if rcn.dtype == _lib.F64 else torch.float32
        self.lib, self.ctx = rcn._lib, rcn._ctx
        self.F, self.classes = rcn.feature_len, rcn.classes
        self.P = sum(rcn.dims[l] * rcn.dims[l + 1] + rcn.dims[l + 1] for l in range(len(rcn.dims) - 1))
        return self

    def _ck(self, st):
        _lib.check(self.lib, self.ctx, st)

    def empty(self, *shape, dtype=None) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype or self.tdtype, device=self.device)

    def to_device(self, a: np.ndarray, dtype=None) -> torch.Tensor:
        with torch.cuda.stream(self.stream):
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            return t.to(dtype) if dtype is not None else t

    def synchronize(self):
        """Waits for the context's stream and surfaces a sticky in-kernel timeout of the last call (rcn_hip_synchronize)."""
        self._ck(self.lib.rcn_hip_synchronize(self.ctx))

    def set_dense_path(self, mode: int):
        self.rcn.set_dense_path(mode)

    def set_feature_kernel(self, mode: int):
        self.rcn.set_feature_kernel(mode)

    def set_option(self, name: str, value: int):
        self.rcn.set_option(name, value)

    def get_option(self, name: str) -> int:
        return self.rcn.get_option(name)

    def fallbacks_taken(self) -> int:
        return self.rcn.fallbacks_taken()

    def last_timeout(self):
        return self.rcn.last_timeout()

    # ---- feature pipeline --------------------------------------------------------------------------------------
    def features(self, imgs_u8: torch.Tensor, standardize: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert imgs_u8.dtype == torch.uint8 and imgs_u8.is_contiguous() and imgs_u8.device == self.device
        n = imgs_u8.shape[0]
        out = out if out is not None else self.empty(n, self.F)
        self._ck(self.lib.rcn_hip_features_dev(self.ctx, _p(imgs_u8), n, _p(out), 1 if standardize else 0))
        return out

    def gen_scales(self, feats: torch.Tensor) -> Tuple[float, float]:
        m, s = C.c_double(), C.c_double()
        self._ck(self.lib.rcn_hip_gen_scales_dev(self.ctx, _p(feats), feats.shape[0], C.byref(m), C.byref(s)))
        return m.value, s.value

    def standardize_(self, feats: torch.Tensor) -> torch.Tensor:
        self._ck(self.lib.rcn_hip_standardize_dev(self.ctx, _p(feats), feats.numel()))
        return feats

    def load_data(self, imgs_u8: torch.Tensor, labels: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """rcn.rs:399-414 on resident images: features -> gen_scales -> standardise; one-hot expectations."""
        x = self.features(imgs_u8)
        self.gen_scales(x)
        self.standardize_(x)
        with torch.cuda.stream(self.stream):
            y = torch.zeros(labels.shape[0], self.classes, dtype=self.tdtype, device=self.device)
            y.scatter_(1, labels.to(torch.int64).view(-1, 1), 1.0)
        return x, y

    # ---- dense network -------------------------------------------------------------------------------------------
    def set_params(self, weights, biases):
        self.rcn.set_params(weights, biases)

    def get_params(self):
        self.synchronize()
        return self.rcn.get_params()

    def params_flat(self) -> torch.Tensor:
        """Zero-copy torch view of the flat device parameter buffer [W_0|b_0|W_1|b_1|...]."""
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._ck(self.lib.rcn_hip_params_dev(self.ctx, C.byref(ptr), C.byref(cnt)))
        esz = 8 if self.tdtype == torch.float64 else 4
        iface = {"shape": (int(cnt.value),), "typestr": "<f8" if esz == 8 else "<f4", "data": (int(ptr.value), False), "version": 2}
        holder = type("_Buf", (), {"__cuda_array_interface__": iface})()
        return torch.as_tensor(holder, device=self.device)

    def train_batch(self, x: torch.Tensor, y: torch.Tensor, eta: float, loss: Optional[torch.Tensor] = None):
        self._ck(self.lib.rcn_hip_train_batch_dev(self.ctx, _p(x), _p(y), x.shape[0], float(eta), _p(loss)))

    def train_epoch(self, X: torch.Tensor, Y: torch.Tensor, perm: Optional[torch.Tensor], B: int, n_batches: int, eta: float,
                    loss: Optional[torch.Tensor] = None):
        if perm is not None:
            assert perm.dtype == torch.int32 and perm.numel() >= B * n_batches
        else:
            assert X.shape[0] >= B * n_batches
        self._ck(self.lib.rcn_hip_train_epoch_dev(self.ctx, _p(X), _p(Y), _p(perm), B, n_batches, float(eta), _p(loss)))

    def train_epoch_images(self, imgs_u8: torch.Tensor, Y: torch.Tensor, perm: Optional[torch.Tensor], B: int, n_batches: int, eta: float,
                           loss: Optional[torch.Tensor] = None, prepare_only: bool = False):
        """The epoch straight from the resident u8 pictures: features + standardise (current scale_set) + packing are one kernel
        per segment, then the same training steps (include/rcn_hip.h: rcn_hip_train_epoch_images_dev)."""
        assert imgs_u8.dtype == torch.uint8 and imgs_u8.is_contiguous()
        fn = self.lib.rcn_hip_prepare_epoch_images_dev if prepare_only else self.lib.rcn_hip_train_epoch_images_dev
        self._ck(fn(self.ctx, _p(imgs_u8), _p(Y), _p(perm), B, n_batches, float(eta), _p(loss)))

    def epoch_begin(self, X: torch.Tensor, Y: torch.Tensor, perm: Optional[torch.Tensor], B: int, n_batches: int):
        """training_set.shuffle + chunks_exact (rcn.rs:146-147) materialised once: batches 0..n_batches of (X, Y, perm) become the
        training kernels' epoch image; epoch_steps then walks them without packing again."""
        if X.dtype == torch.uint8:
            assert X.is_contiguous()
            self._ck(self.lib.rcn_hip_epoch_begin_images_dev(self.ctx, _p(X), _p(Y), _p(perm), B, n_batches))
        else:
            self._ck(self.lib.rcn_hip_epoch_begin_dev(self.ctx, _p(X), _p(Y), _p(perm), B, n_batches))

    def epoch_steps(self, first_batch: int, n_batches: int, eta: float, loss: Optional[torch.Tensor] = None, prepare_only: bool = False):
        """train_batch (rcn.rs:176-223) over batches first_batch .. first_batch + n_batches of the begun epoch."""
        fn = self.lib.rcn_hip_prepare_epoch_steps_dev if prepare_only else self.lib.rcn_hip_epoch_steps_dev
        self._ck(fn(self.ctx, first_batch, n_batches, float(eta), _p(loss)))

    def shuffle(self, perm: torch.Tensor, n: int, passes: int, seed: int):
        """training_set.shuffle (rcn.rs:146) on the device: `passes` pseudo-random permutations of 0..n-1 into perm."""
        assert perm.dtype == torch.int32 and perm.numel() >= n * passes
        self._ck(self.lib.rcn_hip_shuffle_dev(self.ctx, _p(perm), n, passes, seed & 0xFFFFFFFFFFFFFFFF))

    def prepare_epoch(self, X: torch.Tensor, Y: torch.Tensor, perm: Optional[torch.Tensor], B: int, n_batches: int, eta: float,
                      loss: Optional[torch.Tensor] = None):
        """Instantiate (do not run) the graph train_epoch will replay for exactly these arguments."""
        self._ck(self.lib.rcn_hip_prepare_epoch_dev(self.ctx, _p(X), _p(Y), _p(perm), B, n_batches, float(eta), _p(loss)))

    def batch_gradient(self, x: torch.Tensor, y: torch.Tensor, grad: Optional[torch.Tensor] = None,
                       loss_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
        grad = grad if grad is not None else self.empty(self.P)
        self._ck(self.lib.rcn_hip_batch_gradient_dev(self.ctx, _p(x), _p(y), x.shape[0], _p(grad), _p(loss_sum)))
        return grad

    def batch_gradient_perm(self, X: torch.Tensor, Y: torch.Tensor, perm: torch.Tensor, B: int, grad: Optional[torch.Tensor] = None,
                            loss_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Summed gradient of the shard made of rows perm[0..B) of the resident set (no gather copy)."""
        grad = grad if grad is not None else self.empty(self.P)
        self._ck(self.lib.rcn_hip_batch_gradient_perm_dev(self.ctx, _p(X), _p(Y), _p(perm), B, _p(grad), _p(loss_sum)))
        return grad

    def apply_gradient(self, grad: torch.Tensor, scale: float):
        self._ck(self.lib.rcn_hip_apply_gradient_dev(self.ctx, _p(grad), float(scale)))

    # ---- native data-parallel loop (RCCL inside the library; rcn_hip.h "rcn_hip_dp_*") ------------------------------
    def dp_init(self, group=None):
        """Collective.  Creates this context's RCCL communicator: rank 0 draws the unique id, torch.distributed (any
        backend) only carries those 128 bytes to the other ranks; the training loop itself never goes through torch."""
        import torch.distributed as dist
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        buf = C.create_string_buffer(128)
        if rank == 0:
            st = self.lib.rcn_hip_dp_unique_id(buf)
            if st != 0:
                raise RuntimeError(f"rcn_hip_dp_unique_id failed with status {st} (is librccl loadable?)")
        box = [bytes(buf.raw)]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self._ck(self.lib.rcn_hip_dp_init(self.ctx, box[0], rank, world))
        return rank, world

    def dp_p2p_active(self) -> bool:
        """True when the loop's all-reduce is the one-shot xGMI peer-read kernel (csrc/dp_p2p.hpp), False on ncclAllReduce."""
        return bool(self.lib.rcn_hip_dp_p2p_active(self.ctx))

    def dp_p2p_mode(self) -> int:
        """0 ncclAllReduce; 1 peer exchange between kernels; 2 peer exchange inside the gradient kernel (two kernels per step)."""
        return int(self.lib.rcn_hip_dp_p2p_active(self.ctx))

    def dp_epoch_steps(self, first_batch: int, n_batches: int, eta: float, loss: Optional[torch.Tensor] = None):
        """Data-parallel steps over this rank's batches first_batch .. of the epoch epoch_begin packed here (where dp_resident)."""
        self._ck(self.lib.rcn_hip_dp_epoch_steps_dev(self.ctx, first_batch, n_batches, float(eta), _p(loss)))

    def train_epoch_gathers(self, B: int) -> bool:
        """True when train_epoch at this batch size fetches its rows inside the resident kernel (no packed epoch image)."""
        return bool(self.lib.rcn_hip_train_epoch_gathers(self.ctx, B))

    def train_epoch_resident(self, B: int) -> bool:
        """True when train_epoch / epoch_steps at this batch size run on the resident one-XCD kernel (batches of 1..256, f32 and f64)."""
        return bool(self.lib.rcn_hip_train_epoch_resident(self.ctx, B))

    def dp_phase_us(self) -> dict:
        """After a data-parallel call with option xcd_dp_phase = 1 at a shard of 256: where a step waits (rcn_hip_dp_phase_us)."""
        out = (C.c_double * 8)()
        self._ck(self.lib.rcn_hip_dp_phase_us(self.ctx, out, 8))
        keys = ("owner_wait_mean", "owner_wait_max", "member_wait_mean", "member_wait_max", "tail_all_to_all_mean", "tail_all_to_all_max", "step", "steps")
        return {k: round(float(out[i]), 3) for i, k in enumerate(keys)}

    def dp_resident(self, B_shard: int) -> bool:
        """True when dp_train_epoch at this shard size runs on the resident one-XCD kernel with the exchange inside it."""
        return bool(self.lib.rcn_hip_dp_resident(self.ctx, B_shard))

    def dp_p2p_setup(self, group=None, selftest_iters: int = 8):
        """The peer all-reduce with the handle exchange carried by torch.distributed (any backend, e.g. gloo) instead of
        RCCL: export -> all_gather -> attach -> known-answer self-test.  Collective.  Returns (mismatches, timed_out)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = C.create_string_buffer(128)
        self._ck(self.lib.rcn_hip_dp_p2p_export(self.ctx, buf))
        box = [None] * world
        dist.all_gather_object(box, bytes(buf.raw), group=group)
        self._ck(self.lib.rcn_hip_dp_p2p_attach(self.ctx, b"".join(box), rank, world))
        bad, to = C.c_uint(), C.c_uint()
        self._ck(self.lib.rcn_hip_dp_p2p_selftest(self.ctx, selftest_iters, C.byref(bad), C.byref(to)))
        return int(bad.value), int(to.value)

    def dp_p2p_admit(self, group=None) -> int:
        """rcn_hip_dp_init's admission procedure (export -> gather -> attach -> known-answer votes) with torch.distributed (any
        backend, e.g. gloo) as the transport instead of RCCL.  Collective.  Returns the admitted form: 0 none, 1 peer exchange at
        kernel boundaries, 2 inside the gradient kernel -- the same on every rank."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
        VM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int))

        def allgather(_user, mine, allp, nbytes):
            try:
                box = [None] * world
                dist.all_gather_object(box, C.string_at(mine, nbytes), group=group)
                C.memmove(allp, b"".join(box), nbytes * world)
                return 0
            except Exception:
                return 1

        def vote_min(_user, v):
            try:
                t = torch.tensor([v[0]], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                v[0] = int(t.item())
                return 0
            except Exception:
                return 1
        ag, vm = AG(allgather), VM(vote_min)
        self._ck(self.lib.rcn_hip_dp_p2p_admit(self.ctx, rank, world, C.cast(ag, C.c_void_p), C.cast(vm, C.c_void_p), None))
        return self.dp_p2p_mode()

    def dp_finalize(self):
        self._ck(self.lib.rcn_hip_dp_finalize(self.ctx))

    def dp_broadcast_params(self, root: int = 0):
        self._ck(self.lib.rcn_hip_dp_broadcast_params(self.ctx, root))

    def dp_prepare_epoch(self, X: torch.Tensor, Y: torch.Tensor, perm: Optional[torch.Tensor], B_shard: int, n_batches: int, eta: float,
                         loss: Optional[torch.Tensor] = None):
        """Instantiate (do not run) the graph dp_train_epoch will replay for exactly these arguments.  Not collective."""
        self._ck(self.lib.rcn_hip_dp_prepare_epoch_dev(self.ctx, _p(X), _p(Y), _p(perm), B_shard, n_batches, float(eta), _p(loss)))

    def dp_train_epoch(self, X: torch.Tensor, Y: torch.Tensor, perm: Optional[torch.Tensor], B_shard: int, n_batches: int, eta: float,
                       loss: Optional[torch.Tensor] = None):
        """n_batches global train_batch steps; this rank contributes rows perm[j*B_shard ..] of its resident X / Y to
        step j.  Enqueued natively: gradient kernels -> ncclAllReduce -> update, per step, on this stream."""
        if perm is not None:
            assert perm.dtype == torch.int32 and perm.numel() >= B_shard * n_batches
        else:
            assert X.shape[0] >= B_shard * n_batches
        self._ck(self.lib.rcn_hip_dp_train_epoch_dev(self.ctx, _p(X), _p(Y), _p(perm), B_shard, n_batches, float(eta), _p(loss)))

    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = out if out is not None else self.empty(x.shape[0], self.classes)
        self._ck(self.lib.rcn_hip_forward_dev(self.ctx, _p(x), x.shape[0], _p(out)))
        return out

    def evaluate(self, x: torch.Tensor, y: torch.Tensor) -> int:
        n = C.c_int64()
        self._ck(self.lib.rcn_hip_evaluate_dev(self.ctx, _p(x), _p(y), x.shape[0], C.byref(n)))
        return int(n.value)

    def time_kernels(self, x: torch.Tensor, y: torch.Tensor, reps: int = 200) -> Tuple[float, float, float]:
        """Mean microseconds per launch, by HIP events on this stream, of the two kernels of a step back to back with
        themselves -- (k_dense_fwd, k_dense_wgrad) on the sample-tile path, (k_p2_b | k_pipe_b, k_p2_a | k_pipe_a) on the
        feature-sliced path -- and of the alternating pair as the real loop issues it."""
        a, b, p = C.c_double(), C.c_double(), C.c_double()
        self._ck(self.lib.rcn_hip_time_kernels_dev(self.ctx, _p(x), _p(y), x.shape[0], reps, C.byref(a), C.byref(b), C.byref(p)))
        return a.value, b.value, p.value
