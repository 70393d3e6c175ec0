"""ctypes view of the collectives behind the C ABI (``mh_comm_*``, ``mh_sharded_lookup_*``, ``mh_allreduce_dense``):
RCCL driven from ``libmerlin_hip.so`` itself, with host-known sizes only (hipGraph-capturable).

``Comm.create()`` builds the communicator of this process: the 128-byte RCCL id is made on rank 0 and travels over
``torch.distributed`` (any initialised backend; only used as the bootstrap channel) -- a host without torch would
broadcast it over MPI or a TCP store, see INTEGRATION.md.  ``models_amd.distributed`` uses ``torch.distributed`` for the
same exchange by default; ``MERLIN_HIP_COMM=rccl`` switches the row-sharded lookup to this path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import check
from .ops import _dev, _host_ptr_array, _ids_dtype, _ptr, _stream, _workspace


class Comm:
    def __init__(self, handle: C.c_void_p, rank: int, world: int):
        self.handle, self.rank, self.world = handle, rank, world

    @classmethod
    def create(cls) -> "Comm":
        import torch.distributed as dist

        lib = _lib.load()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        uid = (C.c_char * 128)()
        if world > 1:
            if rank == 0:
                check(lib.mh_comm_unique_id(uid), "mh_comm_unique_id")
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        check(lib.mh_comm_init(rank, world, uid, C.byref(h)), "mh_comm_init")
        return cls(h, rank, world)

    def destroy(self) -> None:
        if self.handle:
            _lib.load().mh_comm_destroy(self.handle)
            self.handle = None

    def alltoall(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        """Equal windows: ``send`` / ``recv`` are contiguous with a leading dimension divisible by the world size."""
        nbytes = send.numel() * send.element_size()
        if nbytes % self.world or recv.numel() * recv.element_size() != nbytes:
            raise ValueError("alltoall: buffers must have equal sizes divisible by the world size")
        check(_lib.load().mh_comm_alltoall(self.handle, _ptr(send), _ptr(recv), nbytes // self.world, _stream()), "mh_comm_alltoall")

    def allreduce_(self, flat: torch.Tensor) -> torch.Tensor:
        _dev(flat, "flat", torch.float32)
        check(_lib.load().mh_allreduce_dense(self.handle, _ptr(flat), flat.numel(), _stream()), "mh_allreduce_dense")
        return flat


class ShardedLookup:
    """``mh_sharded_lookup_fwd`` / ``_bwd`` over the concatenated local shards of F one-hot features."""

    def __init__(self, comm: Comm, local_shards: torch.Tensor, base: torch.Tensor, shard_rows: torch.Tensor, capacity: int):
        self.comm, self.local, self.base, self.shard_rows, self.capacity = comm, local_shards, base, shard_rows, int(capacity)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=local_shards.device)
        self._ws: Optional[torch.Tensor] = None
        self._shape = None

    def forward(self, ids: Sequence[torch.Tensor], out: torch.Tensor, out_offset: Sequence[int]) -> torch.Tensor:
        lib = _lib.load()
        F, B, D = len(ids), ids[0].numel(), self.local.shape[1]
        flat = [i.reshape(-1).contiguous() for i in ids]
        idt = _ids_dtype(flat[0], "ids[0]")
        nbytes = lib.mh_sharded_lookup_workspace_bytes(B, F, self.comm.world, self.capacity, D)
        self._ws = _workspace(nbytes, self.local.device, f"sharded_lookup_{id(self)}")
        self._shape = (F, B, D)
        check(lib.mh_sharded_lookup_fwd(self.comm.handle, _host_ptr_array([i.data_ptr() for i in flat]), idt, F, B, self.capacity,
                                        _ptr(self.local), _ptr(self.base), _ptr(self.shard_rows), D, _ptr(out),
                                        out.numel() // B, (C.c_int64 * F)(*[int(o) for o in out_offset]), _ptr(self.overflow),
                                        _ptr(self._ws), self._ws.numel(), _stream()), "mh_sharded_lookup_fwd")
        return out

    def backward(self, grad_stack: torch.Tensor, optimizer: str = "sgd", lr: float = 0.01, eps: float = 1e-7,
                 state: Optional[torch.Tensor] = None, state2: Optional[torch.Tensor] = None, beta1: float = 0.9,
                 beta2: float = 0.999, lr_device: Optional[torch.Tensor] = None) -> None:
        lib = _lib.load()
        F, B, D = self._shape
        if tuple(grad_stack.shape) != (B, F, D) or not grad_stack.is_contiguous():
            raise ValueError(f"grad_stack must be contiguous [{B}, {F}, {D}]")
        check(lib.mh_sharded_lookup_bwd(self.comm.handle, F, B, self.capacity, D, _ptr(grad_stack), _ptr(self.local), _ptr(state),
                                        _ptr(state2), self.local.shape[0], _lib.OPT[optimizer], lr, eps, beta1, beta2,
                                        _ptr(lr_device), _ptr(self._ws), self._ws.numel(), _stream()), "mh_sharded_lookup_bwd")
