"""ctypes view of the collectives behind the C ABI (``mh_comm_*``, ``mh_sharded_lookup_*``, ``mh_allreduce_dense``):
RCCL driven from ``libmerlin_hip.so`` itself, with host-known sizes only (hipGraph-capturable).

``Comm.create()`` builds the communicator of this process: the 128-byte RCCL id is made on rank 0 and travels over
``torch.distributed`` (any initialised backend; only used as the bootstrap channel) -- a host without torch would
broadcast it over MPI or a TCP store, see INTEGRATION.md.

``default()`` is what ``models_amd.distributed`` uses on GPUs: the fixed-window all-to-alls of the row-sharded lookup and the
dense-gradient bucket reduction go through this communicator (``Comm.alltoall_async`` / ``Comm.allreduce_async`` on a
dedicated HIP stream, so that they overlap the caller's kernels like torch's async collectives do).  The communicator is
verified once against ``torch.distributed`` on a small buffer when it is created; ``MERLIN_HIP_COMM=torch`` keeps the
``torch.distributed`` calls instead (debugging).  CPU / gloo runs (the world-size-2 tests) never come here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import check
from .ops import _dev, _host_ptr_array, _ids_dtype, _ptr, _stream, _workspace


class _Work:
    """Handle of a collective issued on the communicator's side stream: ``wait()`` orders the CURRENT stream behind it."""

    def __init__(self, event, keep):
        self.event, self.keep = event, keep

    def wait(self) -> None:
        from .ops import SIDE

        SIDE._ev_wait(torch.cuda.current_stream(), self.event)  # a recorded hand-off while the library records the step
        self.keep = None


_DEFAULT: Optional["Comm"] = None
_DEFAULT_TRIED = False


def default() -> Optional["Comm"]:
    """The process-wide communicator for GPU runs under an initialised NCCL (= RCCL) process group, created and verified on
    first use; None when the collectives should stay on ``torch.distributed`` (CPU / gloo, one rank, MERLIN_HIP_COMM=torch,
    or a failed self-check -- reported once on stderr)."""
    global _DEFAULT, _DEFAULT_TRIED
    if _DEFAULT_TRIED:
        return _DEFAULT
    _DEFAULT_TRIED = True
    import os
    import sys

    import torch.distributed as dist

    if os.environ.get("MERLIN_HIP_COMM", "rccl") == "torch" or not torch.cuda.is_available():
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or dist.get_backend() != "nccl":
        return None
    # Every rank must end up on the SAME side: one rank on torch.distributed while the others use this communicator is a set of
    # mismatched collectives, i.e. a hang (round-3 advisor finding).  The outcome of each stage is agreed with an all-reduce(MIN)
    # of an ok flag over torch.distributed BEFORE the next stage's collectives start.
    dev = torch.device("cuda", torch.cuda.current_device())

    def agreed(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    c, why = None, ""
    try:  # stage 0 (local): the library and RCCL resolve on this rank -- a rank that fails here never enters ncclCommInitRank
        probe = (C.c_char * 128)()
        check(_lib.load().mh_comm_unique_id(probe), "mh_comm_unique_id")
        ok = True
    except Exception as e:  # noqa: BLE001
        ok, why = False, f"{type(e).__name__}: {e}"
    if agreed(ok):
        try:
            c = Comm.create()
            ok = True
        except Exception as e:  # noqa: BLE001
            ok, why = False, f"{type(e).__name__}: {e}"
        if agreed(ok):
            try:
                c.self_check()
                ok = True
            except TimeoutError as e:
                # A collective of the new communicator never completed: it still blocks this device's current stream, so NOTHING
                # ordered behind that stream can run any more -- not the all-reduce the ranks would agree on the fallback with,
                # not its `.item()` (round-4 advisor finding: the agreement itself hung).  The only honest outcome is a message
                # and a non-zero exit; the launcher takes the other ranks down.
                print(f"[models_amd.comm] FATAL: {e}", file=sys.stderr, flush=True)
                os._exit(3)
            except Exception as e:  # noqa: BLE001
                ok, why = False, f"{type(e).__name__}: {e}"
            ok = agreed(ok)
        else:
            ok = False
    else:
        ok = False
    if ok:
        _DEFAULT = c
    else:
        print(f"[models_amd.comm] C-ABI communicator disabled on every rank ({why or 'another rank failed'}); using torch.distributed",
              file=sys.stderr)
        if c is not None:  # created on this rank, but the ranks agreed to fall back: do not leak the RCCL communicator
            try:
                c.destroy()
            except Exception:  # noqa: BLE001
                pass
        _DEFAULT = None
    return _DEFAULT


def which() -> str:
    """Which transport carries the collectives of ``models_amd.distributed`` in this process (reported by bench.py)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return "none (one rank)"
    if _DEFAULT is not None:
        return "mh_comm (RCCL driven from libmerlin_hip.so: mh_comm_alltoall / mh_allreduce_dense)"
    why = "not tried yet" if not _DEFAULT_TRIED else "fallback or MERLIN_HIP_COMM=torch or non-nccl backend"
    return f"torch.distributed ({dist.get_backend()}; {why})"


def _wait_or_die(what: str, seconds: float = 180.0) -> None:
    """Wait for the current stream with a deadline and RAISE when it passes (library code does not end the process: the
    caller -- ``default()`` -- reports the failure, the ranks agree on it and fall back to torch.distributed together;
    a collective that never completes still blocks its stream, which a caller that wants to go on must abandon)."""
    import os
    import time

    ev = torch.cuda.Event()
    ev.record()
    deadline = time.monotonic() + float(os.environ.get("MERLIN_HIP_COMM_TIMEOUT", seconds))
    while not ev.query():
        if time.monotonic() > deadline:
            raise TimeoutError(f"{what} did not complete within the deadline "
                               "(set MERLIN_HIP_COMM=torch to keep the collectives on torch.distributed)")
        time.sleep(0.002)


class Comm:
    def __init__(self, handle: C.c_void_p, rank: int, world: int):
        self.handle, self.rank, self.world = handle, rank, world
        self._stream: Optional[torch.cuda.Stream] = None

    def side_stream(self) -> torch.cuda.Stream:
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        return self._stream

    def _async(self, fn, keep) -> _Work:
        from .ops import SIDE

        st = self.side_stream()
        # the hand-offs to and from the communicator's stream go through ops.SIDE's event helpers: framework events in eager
        # steps, the library's own while it records the step's launch sequence (graph.RecordedStep) -- a replay then repeats
        # them together with the collective itself
        SIDE._ev_wait(st, SIDE._ev_record())  # == st.wait_stream(current stream)
        with torch.cuda.stream(st):
            fn()
            ev = SIDE._ev_record(st)
        return _Work(ev, keep)

    def alltoall_async(self, send: torch.Tensor, recv: torch.Tensor) -> _Work:
        """Equal-window all-to-all on the communicator's stream, ordered after the current stream's work so far."""
        return self._async(lambda: self.alltoall(send, recv), (send, recv))

    def allreduce_async(self, flat: torch.Tensor) -> _Work:
        return self._async(lambda: self.allreduce_(flat), (flat,))

    def self_check(self) -> None:
        """One all-to-all and one all-reduce of a small buffer against torch.distributed (same RCCL underneath): a
        communicator that disagrees must not carry gradients."""
        import torch.distributed as dist

        dev = torch.device("cuda", torch.cuda.current_device())
        W, r = self.world, self.rank
        send = (torch.arange(W * 64, device=dev, dtype=torch.float32) + 1000.0 * r).contiguous()
        got, want = torch.empty_like(send), torch.empty_like(send)
        red, red_t = send.clone(), send.clone()
        dist.all_to_all_single(want, send)
        dist.all_reduce(red_t)
        torch.cuda.synchronize()  # torch's communicator is idle: the two below are the only collectives in flight
        self.alltoall(send, got)
        self.allreduce_(red)
        _wait_or_die("the first collectives of the C-ABI communicator")
        if not (torch.equal(got, want) and torch.allclose(red, red_t, rtol=1e-6, atol=0)):
            raise RuntimeError("self-check against torch.distributed failed")

    @classmethod
    def create(cls, force_rccl: bool = False) -> "Comm":
        """``force_rccl``: at world size 1 build a REAL one-rank RCCL communicator (unique id passed to ``mh_comm_init``), so
        that every collective of this object goes through RCCL instead of the one-rank shortcuts (GPU tests, single-GPU
        bring-up of the N-rank code path)."""
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
        elif force_rccl:
            check(lib.mh_comm_unique_id(uid), "mh_comm_unique_id")
        h = C.c_void_p()
        check(lib.mh_comm_init(rank, world, uid if (world > 1 or force_rccl) else None, C.byref(h)), "mh_comm_init")
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
