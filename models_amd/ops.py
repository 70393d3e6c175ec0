"""Tensor-level wrappers over the C ABI (include/merlin_hip.h).

Every function takes PyTorch-ROCm device tensors purely as buffer carriers, validates shape /
dtype / contiguity, and launches the HIP kernel on the current torch stream.  CPU tensors are
rejected: the product path has no CPU fallback (the CPU restatement lives in ``oracle/`` and is
test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT, COMBINER, MH_I32, MH_I64, check

__all__ = [
    "embedding_gather",
    "embedding_bag",
    "embedding_dense_list",
    "linear",
    "dot_interaction",
]


class _KernelTimer:
    """Optional per-op HIP-event timing (bench.py): events are recorded on the stream the kernels
    are launched on (torch's current stream), one (start, end) pair around each C-ABI call.  Every launch also
    records the ALGORITHMIC bytes / flops of exactly that launch (computed from its own arguments), so a
    roofline figure never mixes launches of different sizes (sharded vs replicated lookups share an op name)."""

    def __init__(self):
        self.enabled = False
        self.events = {}

    def enable(self):
        self.enabled, self.events = True, {}

    def disable(self):
        self.enabled = False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in evs]
            out[name] = {"avg_ms": sum(ms) / len(ms), "launches": len(ms), "total_ms": sum(ms),
                         "bytes": sum(e[2] for e in evs), "flops": sum(e[3] for e in evs)}
        return out


TIMER = _KernelTimer()


class _timed:
    def __init__(self, name, nbytes=0, flops=0):
        self.name, self.nbytes, self.flops = name, nbytes, flops

    def __enter__(self):
        if TIMER.enabled:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if TIMER.enabled:
            self.b.record()
            TIMER.events.setdefault(self.name, []).append((self.a, self.b, self.nbytes, self.flops))
        return False


_OPT_ROW_PASSES = {"sgd": 3, "adagrad": 5, "adam": 7, "lazy_adam": 7}  # grad r + (weight, state...) r/w per touched row


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Marker:
    """Record mode: the END of segment ``seg`` of a step being recorded (the role an event plays in an eager step)."""

    __slots__ = ("seg",)

    def __init__(self, seg: int):
        self.seg = seg


class _CEvent:
    """An event owned by the open launch recording of the C library (``mh_record_event``)."""

    __slots__ = ("eid",)

    def __init__(self, eid: int):
        self.eid = int(eid)


class StepRecorder:
    """Records ONE step as a list of SEGMENTS -- maximal runs of launches on one logical stream -- each captured into its own
    hipGraph, with the dependency edges between them; ``graph.SegmentedStep`` replays the list on real streams.

    Why: a whole step captured into one hipGraph replays on ONE hardware queue (ROCm 7.2): no host work, but none of the overlap
    the eager step gets from its side streams.  Launching ~10 small graphs on 3-4 streams with events between them keeps both:
    host cost = a handful of graph launches, GPU overlap = the eager step's.  While recording, everything executes (is captured)
    sequentially on the capture stream; ``SIDE.on / mark / wait / join`` cut segments and note edges instead of touching streams."""

    def __init__(self):
        self.segments = []      # [{"stream": name, "graph": CUDAGraph, "deps": [segment ids]}]
        self._cur = None        # index of the segment being captured
        self._stack = []        # logical stream names: "main" at the bottom, a side name while inside SIDE.on(...)
        self._pending = {}      # side stream name -> its last segment (not yet joined by main)
        self._last_on = {}      # logical stream -> last segment captured for it

    # -- segment control ------------------------------------------------------------------------------------------------
    def begin(self) -> None:
        self._stack = ["main"]
        self._open("main", [])

    def finish(self) -> None:
        self._close()

    def _open(self, stream: str, deps) -> None:
        g = torch.cuda.CUDAGraph()
        deps = sorted({d for d in deps if d is not None and self.segments[d]["stream"] != stream})
        self.segments.append({"stream": stream, "graph": g, "deps": deps})
        self._cur = len(self.segments) - 1
        self._last_on[stream] = self._cur
        g.capture_begin()

    def _close(self) -> int:
        import warnings

        i = self._cur
        with warnings.catch_warnings(record=True) as caught:  # torch warns when a capture holds no node: two cuts in a row
            warnings.simplefilter("always")
            self.segments[i]["graph"].capture_end()
        self.segments[i]["empty"] = any("Graph is empty" in str(w.message) for w in caught)
        self._cur = None
        return i

    def _cut(self, extra_deps=()) -> int:
        """End the current segment and continue on the same logical stream; returns the index of the ended segment."""
        ended = self._close()
        self._open(self._stack[-1], list(extra_deps))
        return ended

    # -- what SIDE forwards to ------------------------------------------------------------------------------------------
    def mark(self) -> _Marker:
        return _Marker(self._cut())

    def wait(self, marker: _Marker) -> None:
        self._cut([marker.seg])

    def enter(self, name: str, after, kind: str = None) -> None:
        prev = self._close()
        deps = [m.seg for m in after] if after else [prev]
        self._stack.append(name)
        self._kinds = getattr(self, "_kinds", [])
        self._kinds.append(kind or name)
        self._open(name, deps)

    def leave(self) -> None:
        ended = self._close()
        name = self._stack.pop()
        self._pending[name] = ended
        self._kind_last = getattr(self, "_kind_last", {})
        self._kind_last[self._kinds.pop()] = ended   # several kinds may share one logical stream: join_kind waits for ONE of them
        self._open(self._stack[-1], [])

    def join_kind(self, kind: str) -> None:
        """Main waits for the last block of one KIND of side work only (the stream it shares with other kinds keeps running)."""
        seg = getattr(self, "_kind_last", {}).get(kind)
        if seg is not None:
            self._cut([seg])

    def join(self, names=None) -> None:
        names = list(self._pending) if names is None else [n for n in names if n in self._pending]
        if not names:
            return
        deps = [self._pending.pop(n) for n in names]
        self._cut(deps)


class _SideStreams:
    """Independent work of one train step on extra HIP streams: the dW / db GEMMs of an MLP backward (stream
    "dw") run beside the dX chain, the fused embedding backward (stream "sparse") beside the bottom-MLP backward, the id-only
    half of the sparse update (stream "sort") beside the top MLP.

    ``with SIDE.on(name, after=..., keep=...):`` runs a block on side stream ``name`` -- ordered after everything enqueued so
    far on the current stream, or after the given ``mark()``s; ``join`` makes the launch stream wait for the side streams.
    Tensors a side stream reads are kept alive until the join (the caching allocator would otherwise hand their memory to
    later launch-stream kernels).  Two modes: EAGER (real streams and events) and RECORD (a ``StepRecorder`` is installed:
    the same calls cut the step into per-stream graph segments, replayed by ``graph.SegmentedStep``).  A step captured into
    ONE graph gets neither (see ``active``).  ``MERLIN_HIP_SIDE_STREAMS=0`` turns it off; the per-op timer of bench.py runs
    single-stream."""

    def __init__(self):
        import os

        mode = os.environ.get("MERLIN_HIP_SIDE_STREAMS", "1")  # "0" | "1" | a comma list of kinds, e.g. "sort,sparse" (experiments)
        self.enabled = mode != "0"
        self.kinds = {"dw", "sparse", "sort"} if mode in ("0", "1") else {k.strip() for k in mode.split(",") if k.strip()}
        self._streams = {}
        self._pending = set()
        self._keep = []
        self._defer = 0
        self.recorder: Optional[StepRecorder] = None
        self.crec = False  # the C library is recording the launch sequence (graph.RecordedStep): hand-offs go through its events
        # ONE physical side stream for all three kinds: every extra stream is one more join at the end of the step and more
        # cross-stream hand-offs in between, and a hand-off costs of the order of 20 us here.  The side work is a chain anyway
        # (sort -> [dW, after dX] -> sparse apply, which needs the sort).  Measured on one box, alternating runs: eager
        # 0.966 -> 0.953 ms, segmented 1.000 -> 0.989 ms; dW alone moved onto the sort stream: +10..20 us.
        # MERLIN_HIP_SIDE_ALIAS=none restores one stream per kind; "dw=sort" style lists are accepted for experiments.
        self._kind_event = {}
        spec = os.environ.get("MERLIN_HIP_SIDE_ALIAS", "dw=sort,sparse=sort")
        self._alias = dict(kv.split("=") for kv in spec.split(",") if "=" in kv)

    def active(self, kind: str = None) -> bool:
        # not under a plain hipGraph capture: ROCm 7 replays a captured graph on ONE hardware queue (measured: the kernel
        # trace of a replayed multi-branch step shows no overlap and the event nodes cost ~2 %), so side streams only
        # pay in eager steps and in recorded (segmented) steps
        if not (self.enabled and (kind is None or kind in self.kinds) and not TIMER.enabled and torch.cuda.is_available()):
            return False
        return self.recorder is not None or not torch.cuda.is_current_stream_capturing()

    def stream(self, name: str):
        dev = torch.cuda.current_device()
        st = self._streams.get((dev, name))
        if st is None:
            st = torch.cuda.Stream(device=dev)  # default priority (a high-priority side stream was measured: no gain, profiles/r4_ab_side_streams.txt)
            self._streams[(dev, name)] = st
        return st

    # -- events ---------------------------------------------------------------------------------------------------------
    # Eager mode has two flavours of event: the framework's (torch.cuda.Event), and -- while the C library is RECORDING the launch
    # sequence of a step (graph.RecordedStep: ``crec`` set) -- the library's own (mh_record_event / mh_record_wait_event), so that
    # the hand-offs between the streams are part of the recorded sequence and are replayed with it.
    def _ev_record(self, stream=None):
        if self.crec:
            st = stream if stream is not None else torch.cuda.current_stream()
            eid = C.c_int64(-1)
            check(_lib.load().mh_record_event(C.c_void_p(st.cuda_stream), C.byref(eid)), "mh_record_event")
            return _CEvent(eid.value)
        ev = torch.cuda.Event()
        ev.record(stream) if stream is not None else ev.record()
        return ev

    def _ev_wait(self, stream, ev) -> None:
        if isinstance(ev, _CEvent):
            if not self.crec:
                raise RuntimeError("an event of a recorded launch sequence used outside its recording")
            check(_lib.load().mh_record_wait_event(C.c_void_p(stream.cuda_stream), ev.eid), "mh_record_wait_event")
        else:
            stream.wait_event(ev)

    def mark(self):
        """An ordering point on the current stream: pass it to ``on(after=[...])`` or ``wait``."""
        if self.recorder is not None:
            return self.recorder.mark()
        return self._ev_record()

    def wait(self, marker) -> None:
        """The current stream waits for ``marker``."""
        if isinstance(marker, _Marker):
            if self.recorder is None:
                raise RuntimeError("a marker of a recorded step used outside its recording")
            self.recorder.wait(marker)
        else:
            self._ev_wait(torch.cuda.current_stream(), marker)

    class _On:
        def __init__(self, owner, name, after, keep, kind=None):
            self.owner, self.name, self.after, self.keep, self.kind = owner, name, after, keep, kind or name
            self._ctx = None
            self._st = None

        def __enter__(self):
            o = self.owner
            if o.recorder is not None:
                o.recorder.enter(self.name, self.after, self.kind)
                return self
            st = o.stream(self.name)
            if self.after:
                for ev in self.after:
                    o._ev_wait(st, ev)
            else:
                o._ev_wait(st, o._ev_record())  # == st.wait_stream(current stream)
            o._pending.add(st)
            o._keep.extend(self.keep)
            self._st = st
            self._ctx = torch.cuda.stream(st)
            self._ctx.__enter__()
            return self

        def __exit__(self, *exc):
            if self._ctx is not None:
                r = self._ctx.__exit__(*exc)
                if self.owner._alias:  # kinds share a stream: remember where THIS kind's work ends on it
                    self.owner._kind_event[self.kind] = self.owner._ev_record(self._st)
                return r
            self.owner.recorder.leave()
            return False

    def on(self, name: str, after=None, keep=()):
        """Context: the block's launches go to side stream ``name``, ordered after ``after`` (a list of ``mark()``s) or,
        by default, after everything enqueued so far on the current stream."""
        return _SideStreams._On(self, self._alias.get(name, name), [a for a in (after or []) if a is not None], keep, kind=name)

    def retire(self, buf) -> None:
        """A workspace being replaced: keep it alive until the next join if any side stream has work in flight."""
        if self._pending:
            self._keep.append(buf)

    def join(self) -> None:
        self._kind_event.clear()
        if self.recorder is not None:
            self.recorder.join()
            return
        if self._pending:
            cur = torch.cuda.current_stream()
            for st in self._pending:
                self._ev_wait(cur, self._ev_record(st))  # == cur.wait_stream(st)
            self._pending.clear()
        self._keep.clear()

    def join_stream(self, name: str) -> None:
        """The current stream waits for ONE kind of side work (its kept tensors stay alive until the full join).  When kinds
        share a physical stream, only for the end of that kind's last block -- not for what other kinds queued behind it
        (the optimizer joins "dw" while the sparse apply is running on the same stream)."""
        if self._alias.get(name, name) != name or name in self._alias.values():
            if self.recorder is not None:
                self.recorder.join_kind(name)
                return
            ev = self._kind_event.pop(name, None)
            if ev is not None:
                self._ev_wait(torch.cuda.current_stream(), ev)
            return
        name = self._alias.get(name, name)
        if self.recorder is not None:
            self.recorder.join([name])
            return
        st = self._streams.get((torch.cuda.current_device(), name)) if torch.cuda.is_available() else None
        if st is not None and st in self._pending:
            self._ev_wait(torch.cuda.current_stream(), self._ev_record(st))
            self._pending.discard(st)

    def maybe_join(self) -> None:
        if self._defer == 0:
            self.join()

    class _Deferred:
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.owner._defer += 1

        def __exit__(self, *exc):
            self.owner._defer -= 1
            if self.owner._defer == 0:
                self.owner.join()
            return False

    def deferred(self):
        """Within this context side work is joined only at exit (or by an explicit ``join()``)."""
        return _SideStreams._Deferred(self)


SIDE = _SideStreams()


def _dev(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _lib.MerlinHipError(
            f"{name} is a CPU tensor: models_amd ops only run on a HIP device (no CPU fallback)"
        )
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _ids_dtype(t: torch.Tensor, name: str) -> int:
    if t.dtype == torch.int32:
        return MH_I32
    if t.dtype == torch.int64:
        return MH_I64
    raise TypeError(f"{name} must be int32 or int64, got {t.dtype}")


def _host_ptr_array(vals: Sequence[int]):
    arr = (C.c_void_p * len(vals))(*vals)
    return arr


# --------------------------------------------------------------------------------------------
# embeddings
# --------------------------------------------------------------------------------------------
def embedding_gather(
    tables: Sequence[torch.Tensor],
    ids: Sequence[torch.Tensor],
    out: Optional[torch.Tensor] = None,
    out_slot: Optional[Sequence[int]] = None,
    n_slots: Optional[int] = None,
    out_offset: Optional[Sequence[int]] = None,
) -> torch.Tensor:
    """One-hot lookup of F features in one launch -> stacked ``[B, n_slots, D]``.

    ``tables[f]`` is ``[V_f, D]`` fp32, ``ids[f]`` is ``[B]`` or ``[B, 1]`` int32/int64.  Feature f
    lands in slot ``out_slot[f]`` (default f) of the stacked output -- the layout
    ``StackFeatures`` produces in the reference (core/aggregation.py:101-108).
    """
    lib = _lib.load()
    F = len(tables)
    if F == 0 or F != len(ids):
        raise ValueError("embedding_gather: need one ids tensor per table")
    D = tables[0].shape[1]
    B = ids[0].shape[0]
    idt = _ids_dtype(ids[0], "ids[0]")
    flat_ids = []
    for f, (w, i) in enumerate(zip(tables, ids)):
        _dev(w, f"tables[{f}]", torch.float32)
        _dev(i, f"ids[{f}]")
        if w.dim() != 2 or w.shape[1] != D or not w.is_contiguous():
            raise ValueError(f"tables[{f}] must be contiguous [V, {D}]")
        if _ids_dtype(i, f"ids[{f}]") != idt:
            raise TypeError("embedding_gather: all ids must share one dtype")
        i = i.reshape(-1)
        if i.shape[0] != B or not i.is_contiguous():
            raise ValueError(f"ids[{f}] must be contiguous with {B} entries")
        flat_ids.append(i)
    slots = list(range(F)) if out_slot is None else [int(s) for s in out_slot]
    if n_slots is None:
        n_slots = (max(slots) + 1) if out is None else out.shape[1]
    if out is None:
        out = torch.empty((B, n_slots, D), dtype=torch.float32, device=tables[0].device)
    else:
        _dev(out, "out", torch.float32)
        if out_offset is None and (out.dim() != 3 or out.shape[2] != D):
            raise ValueError("out must be [B, n_slots, D] (or pass out_offset for a [B, W] concat buffer)")
        if out.shape[0] != B or not out.is_contiguous():
            raise ValueError("out must be contiguous with B rows")
    row_stride = out.numel() // max(B, 1)
    offsets = [s_ * D for s_ in slots] if out_offset is None else [int(o) for o in out_offset]
    if B == 0:
        return out
    for start in range(0, F, _lib.MAX_FEATURES):
        sl = slice(start, min(F, start + _lib.MAX_FEATURES))
        n = sl.stop - sl.start
        tab = _host_ptr_array([w.data_ptr() for w in tables[sl]])
        idp = _host_ptr_array([i.data_ptr() for i in flat_ids[sl]])
        rows = (C.c_int64 * n)(*[w.shape[0] for w in tables[sl]])
        slot = (C.c_int64 * n)(*offsets[sl])
        with _timed("embedding_gather", nbytes=B * n * (2 * D * 4 + flat_ids[0].element_size())):
            check(
                lib.mh_embedding_gather_fwd(tab, rows, idp, idt, B, n, D, _ptr(out), row_stride, slot, _stream()),
                "mh_embedding_gather_fwd",
            )
    return out


def embedding_bag(
    table: torch.Tensor, values: torch.Tensor, offsets: torch.Tensor, combiner: str = "mean",
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Ragged multi-hot lookup (CSR ``values``/``offsets``) with a sum/mean/sqrtn combiner."""
    lib = _lib.load()
    _dev(table, "table", torch.float32)
    _dev(values, "values")
    _dev(offsets, "offsets")
    idt = _ids_dtype(values, "values")
    if offsets.dtype != values.dtype:
        raise TypeError("offsets and values must share one integer dtype")
    if combiner not in COMBINER or combiner == "max":
        raise ValueError(f"ragged combiner must be one of ['mean', 'sqrtn', 'sum'], got {combiner!r}")
    B = offsets.shape[0] - 1
    D = table.shape[1]
    values = values.reshape(-1).contiguous()
    offsets = offsets.reshape(-1).contiguous()
    if out is None:
        out = torch.empty((B, D), dtype=torch.float32, device=table.device)
    if B == 0:
        return out
    check(
        lib.mh_embedding_bag_fwd(_ptr(table), table.shape[0], _ptr(values), values.shape[0], _ptr(offsets),
                                 idt, B, D, COMBINER[combiner], _ptr(out), out.stride(0), _stream()),
        "mh_embedding_bag_fwd",
    )
    return out


def embedding_dense_list(
    table: torch.Tensor, ids: torch.Tensor, combiner: str = "mean", out: Optional[torch.Tensor] = None
) -> torch.Tensor:
    """Fixed-length list ``[B, L]`` lookup reduced over axis 1 (mean / sum)."""
    lib = _lib.load()
    _dev(table, "table", torch.float32)
    _dev(ids, "ids")
    idt = _ids_dtype(ids, "ids")
    if combiner not in ("mean", "sum", "max"):
        raise ValueError("Only 'mean', 'sum', and 'max' str combiners is implemented for dense list/multi-hot embedded features.")
    if ids.dim() == 3 and ids.shape[-1] == 1:
        ids = ids.squeeze(-1)
    if ids.dim() != 2:
        raise ValueError("ids must be [B, L]")
    ids = ids.contiguous()
    B, L = ids.shape
    D = table.shape[1]
    if out is None:
        out = torch.empty((B, D), dtype=torch.float32, device=table.device)
    check(
        lib.mh_embedding_dense_list_fwd(_ptr(table), table.shape[0], _ptr(ids), idt, B, L, D,
                                        COMBINER[combiner], _ptr(out), out.stride(0), _stream()),
        "mh_embedding_dense_list_fwd",
    )
    return out


# --------------------------------------------------------------------------------------------
# dense layers
# --------------------------------------------------------------------------------------------
def _rowmajor_2d(t: torch.Tensor, name: str) -> torch.Tensor:
    _dev(t, name, torch.float32)
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name} must be 2-D with unit inner stride")
    return t


def tower_arith() -> bool:
    """The tower layers (N = 128, K <= 1024, batch >= 4096) run in the fp32-grade six-term split "bf16x6" (mh_tower_split.hip) unless
    MERLIN_HIP_GEMM_ARITH=f32 asks for the exact fmaf-chain kernels everywhere."""
    return os.environ.get("MERLIN_HIP_GEMM_ARITH", "") != "f32"


def _tower_ok(M: int, K: int, N: int, x: torch.Tensor) -> bool:
    return (tower_arith() and N == 128 and bool(_lib.load().mh_tower_supported(M, K, N)) and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0)


def _linear_split_ok(M: int, K: int, N: int) -> bool:
    """The wide Dense layers that run on the split-bf16 GEMM of mh_gemm_split.hip (six terms by default, three under
    MERLIN_HIP_GEMM_ARITH=bf16x3, never under =f32).  The operands are split (and, for dW, transposed) per call: M (K + N) elements of
    preparation against M K N of saved MFMA time, so the layer has to be wide in BOTH directions -- K N / (K + N) >= 384 keeps the
    DCN-v2 deep tower's 3341 -> 512 (forward 1.98 -> 1.81 ms, backward 4.34 -> 3.99 at batch 64 K) and sends 512 -> 256 (two-tower and
    DCN-v2; backward 0.41 ms exact chain against 0.89 split, forward equal) to the exact-chain kernels."""
    return gemm_arith() != "f32" and M >= 1024 and K >= 512 and N >= 256 and K * N >= 384 * (K + N)


def linear(
    x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor] = None, activation: Optional[str] = None,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """``act(x @ W + b)`` with ``W`` in the Keras kernel layout ``[K, N]``."""
    lib = _lib.load()
    _rowmajor_2d(x, "x")
    _dev(W, "W", torch.float32)
    if activation not in ACT:
        raise ValueError(f"unsupported activation {activation!r}")
    M, K = x.shape
    if W.dim() != 2 or W.shape[0] != K or not W.is_contiguous():
        raise ValueError(f"W must be contiguous [{K}, N]")
    N = W.shape[1]
    if b is not None:
        _dev(b, "b", torch.float32)
        if b.numel() != N:
            raise ValueError("bias must have N entries")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    else:
        _rowmajor_2d(out, "out")
    if M == 0:
        return out
    if _tower_ok(M, K, N, x):  # the tower layers in the fp32-grade six-term split (default; MERLIN_HIP_GEMM_ARITH=f32: the exact chain)
        ws = _workspace(lib.mh_tower_workspace_bytes(M, K, N), x.device, f"tower_{K}x{N}")
        with _timed(f"linear_{K}x{N}", nbytes=4 * (M * K + K * N + M * N), flops=2 * M * K * N):
            check(lib.mh_tower_linear_fwd(_ptr(x), x.stride(0), _ptr(W), _ptr(b), M, K, N, ACT[activation], _ptr(out), out.stride(0),
                                          _ptr(ws), ws.numel(), _stream()), "mh_tower_linear_fwd")
        return out
    if _linear_split_ok(M, K, N):  # wide Dense layers on the split-bf16 GEMM (six terms by default)
        _sync_gemm_arith(lib)
        ws = _workspace(lib.mh_linear_split_workspace_bytes(M, K, N), x.device, "linear_split")
        with _timed(f"linear_{K}x{N}", nbytes=4 * (M * K + K * N + M * N), flops=2 * M * K * N):
            check(lib.mh_linear_bias_act_fwd_split(_ptr(x), x.stride(0), _ptr(W), _ptr(b), M, K, N, ACT[activation], _ptr(out),
                                                   out.stride(0), _ptr(ws), ws.numel(), _stream()), "mh_linear_bias_act_fwd_split")
        return out
    with _timed(f"linear_{K}x{N}", nbytes=4 * (M * K + K * N + M * N), flops=2 * M * K * N):
        check(
            lib.mh_linear_bias_act_fwd(_ptr(x), x.stride(0), _ptr(W), _ptr(b), M, K, N, ACT[activation],
                                       _ptr(out), out.stride(0), _stream()),
            "mh_linear_bias_act_fwd",
        )
    return out


def _chain_dims(x: torch.Tensor, Ws: Sequence[torch.Tensor]):
    dims = [int(x.shape[1])] + [int(W.shape[1]) for W in Ws]
    for i, W in enumerate(Ws):
        _dev(W, f"W[{i}]", torch.float32)
        if W.dim() != 2 or W.shape[0] != dims[i] or not W.is_contiguous():
            raise ValueError(f"W[{i}] must be contiguous [{dims[i]}, N]")
    return dims, (C.c_int32 * len(dims))(*dims)


def mlp_chain_supported(dims: Sequence[int]) -> bool:
    """True when one fused launch covers the Dense chain ``dims[0] -> dims[1] -> ...``."""
    if not 3 <= len(dims) <= 4:
        return False
    arr = (C.c_int32 * len(dims))(*[int(d) for d in dims])
    return bool(_lib.load().mh_mlp_chain_supported(len(dims) - 1, arr))


def mlp_chain(x: torch.Tensor, Ws: Sequence[torch.Tensor], bs: Sequence[Optional[torch.Tensor]],
              activations: Sequence[Optional[str]], outs: Optional[Sequence[Optional[torch.Tensor]]] = None):
    """``y_l = act_l(y_{l-1} @ W_l + b_l)`` for a run of small Dense layers in ONE launch; returns every layer's
    output (the backward needs them).  ``outs[l]`` may name a destination (e.g. a slot of a stacked buffer)."""
    lib = _lib.load()
    _rowmajor_2d(x, "x")
    L = len(Ws)
    dims, cdims = _chain_dims(x, Ws)
    M = x.shape[0]
    ys = []
    for l in range(L):
        o = outs[l] if outs is not None and outs[l] is not None else torch.empty((M, dims[l + 1]), dtype=torch.float32, device=x.device)
        _rowmajor_2d(o, f"outs[{l}]")
        if bs[l] is not None:
            _dev(bs[l], f"b[{l}]", torch.float32)
        if activations[l] not in ACT:
            raise ValueError(f"unsupported activation {activations[l]!r}")
        ys.append(o)
    if M == 0:
        return ys
    cW = _host_ptr_array([W.data_ptr() for W in Ws])
    cb = _host_ptr_array([0 if b is None else b.data_ptr() for b in bs])
    cact = (C.c_int32 * L)(*[ACT[a] for a in activations])
    cy = _host_ptr_array([y.data_ptr() for y in ys])
    cld = (C.c_int64 * L)(*[y.stride(0) for y in ys])
    name = "mlp_chain_" + "x".join(str(d) for d in dims)
    nbytes = 4 * (M * sum(dims) + sum(dims[i] * dims[i + 1] for i in range(L)))
    with _timed(name, nbytes=nbytes, flops=2 * M * sum(dims[i] * dims[i + 1] for i in range(L))):
        check(lib.mh_mlp_chain_fwd(_ptr(x), x.stride(0), M, L, cdims, cW, cb, cact, cy, cld, _stream()), "mh_mlp_chain_fwd")
    return ys


# Work whose result nothing on the step's critical path waits for (the slab reduction of a chain backward: dW / db are first read by the
# optimizer; the final sum of the BCE partials: the scalar is only reported) is collected here while a train step runs and issued at
# the step's TAIL, on the launch stream, where that stream idles behind the side stream's sparse apply -- not between the kernels of
# the dependent chain (6 + 5 us there).  Not a side stream: a cross-stream hand-off costs more than these kernels.
TAIL = [None]   # None: issue immediately; a list: collect (see tail_work / run_tail)


class _TailWork:
    def __enter__(self):
        self._outer = TAIL[0]
        import os as _os

        # Parking moves two small launches behind the sparse hand-off.  Measured (round 4, alternating runs on one box each): it LOSES
        # ~20 us in the eager step (0.945 -> 0.966 ms) and in the C-recorded replay (0.958 -> 0.976), and WINS ~10 us when the step is
        # replayed as hipGraph segments (segmented 0.977 -> 0.967).  So: on while a step is stream-captured, off otherwise.
        replayed = SIDE.recorder is not None or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
        TAIL[0] = [] if replayed else None
        return self

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                run_tail()
        finally:
            TAIL[0] = self._outer
        return False


def tail_work():
    """Context of one train step: ops may park their non-critical second halves; ``run_tail()`` (called by the optimizer in front of
    the dense update, and at exit) issues them in order."""
    return _TailWork()


def run_tail() -> None:
    if TAIL[0]:
        for f in TAIL[0]:
            f()
        TAIL[0].clear()


def mlp_chain_backward(x: torch.Tensor, Ws: Sequence[torch.Tensor], ys: Sequence[torch.Tensor],
                       activations: Sequence[Optional[str]], grad: torch.Tensor, pre_masked: bool = False,
                       need_dx: bool = True, need_db: Optional[Sequence[bool]] = None, x_activation: Optional[str] = None):
    """Backward of ``mlp_chain``: returns ``(dx | None, [dW_l], [db_l | None])``.  ``grad`` is d/dy_L (or dz_L when
    ``pre_masked``); dx has the producer's activation derivative (``x_activation``) folded in, like ``linear_backward``."""
    lib = _lib.load()
    _rowmajor_2d(x, "x")
    _rowmajor_2d(grad, "grad")
    L = len(Ws)
    dims, cdims = _chain_dims(x, Ws)
    M = x.shape[0]
    if grad.shape != (M, dims[-1]):
        raise ValueError(f"grad must be [{M}, {dims[-1]}]")
    need_db = [True] * L if need_db is None else list(need_db)
    dx, lddx = None, dims[0]
    if need_dx:
        lddx = (dims[0] + 3) // 4 * 4
        buf = torch.empty((M, lddx), dtype=torch.float32, device=x.device)
        if lddx != dims[0]:
            zero_pad_columns(buf, dims[0])
        dx = buf[:, :dims[0]]
    dWs = [torch.empty((dims[l], dims[l + 1]), dtype=torch.float32, device=x.device) for l in range(L)]
    dbs = [torch.empty((dims[l + 1],), dtype=torch.float32, device=x.device) if need_db[l] else None for l in range(L)]
    nbytes = lib.mh_mlp_chain_bwd_workspace_bytes(M, L, cdims)
    if nbytes < 0:
        raise _lib.MerlinHipError("mh_mlp_chain_bwd_workspace_bytes: unsupported chain")
    name = "mlp_chain_bwd_" + "x".join(str(d) for d in dims)
    # a parked slab reduction must not meet another chain's slabs: one workspace per parked position of the step
    ws = _workspace(nbytes, x.device, name if TAIL[0] is None else f"{name}#{len(TAIL[0])}")
    cW = _host_ptr_array([W.data_ptr() for W in Ws])
    cact = (C.c_int32 * L)(*[ACT[a] for a in activations])
    cy = _host_ptr_array([_rowmajor_2d(y, "y").data_ptr() for y in ys])
    cld = (C.c_int64 * L)(*[y.stride(0) for y in ys])
    cdW = _host_ptr_array([t.data_ptr() for t in dWs])
    cdb = _host_ptr_array([0 if t is None else t.data_ptr() for t in dbs])
    if TAIL[0] is not None and not TIMER.enabled:
        # the strip kernel (dx: what the previous layer's backward waits for) now, the slab reduction (dW / db) at the step's tail
        check(lib.mh_mlp_chain_bwd_partial(_ptr(x), x.stride(0), M, L, cdims, cW, cact, cy, cld, _ptr(grad), grad.stride(0),
                                           1 if pre_masked else 0, ACT[x_activation], _ptr(dx), lddx, _ptr(ws), ws.numel(), _stream()),
              "mh_mlp_chain_bwd_partial")
        TAIL[0].append(lambda: check(lib.mh_mlp_chain_bwd_reduce(M, L, cdims, cdW, cdb, _ptr(ws), ws.numel(), _stream()),
                                     "mh_mlp_chain_bwd_reduce"))
        return dx, dWs, dbs
    with _timed(name, nbytes=4 * M * (sum(dims) + (dims[0] if need_dx else 0) + dims[-1]),
                flops=(4 * M * sum(dims[i] * dims[i + 1] for i in range(L)))):
        check(lib.mh_mlp_chain_bwd(_ptr(x), x.stride(0), M, L, cdims, cW, cact, cy, cld, _ptr(grad), grad.stride(0),
                                   1 if pre_masked else 0, ACT[x_activation], _ptr(dx), lddx, cdW, cdb, _ptr(ws), ws.numel(),
                                   _stream()), "mh_mlp_chain_bwd")
    return dx, dWs, dbs


def dot_interaction(
    x: torch.Tensor, tail: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, tail_first: bool = True
) -> torch.Tensor:
    """Strict-upper-triangle pairwise dots of ``x[B, F, D]`` (row-major pair order) with the shortcut
    ``tail[B, T]`` in the same output row: ``[tail | pairs]`` (``tail_first``, the reference's DLRM order
    ``[bottom_block | interactions]``: tf/core/combinators.py:564-569 + tf/core/aggregation.py:54-66) or
    ``[pairs | tail]``."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    if x.dim() != 3 or not x.is_contiguous():
        raise ValueError("x must be contiguous [B, F, D]")
    B, F, D = x.shape
    P = F * (F - 1) // 2
    T = 0
    if tail is not None:
        _rowmajor_2d(tail, "tail")
        T = tail.shape[1]
    if out is None:
        out = torch.empty((B, P + T), dtype=torch.float32, device=x.device)
    else:
        _rowmajor_2d(out, "out")
    if B == 0:
        return out
    with _timed("dot_interaction", nbytes=B * (F * D + P + T) * 4, flops=B * D * F * (F - 1)):
        check(
            lib.mh_dot_interaction_fwd(_ptr(x), B, F, D, _ptr(tail), 0 if tail is None else tail.stride(0), T,
                                       1 if tail_first else 0, _ptr(out), out.stride(0), _stream()),
            "mh_dot_interaction_fwd",
        )
    return out


# --------------------------------------------------------------------------------------------
# backward / training ops
# --------------------------------------------------------------------------------------------
_WS = {}


# Persistent buffers (workspaces, a block's padded input buffer, a sampler's hold buffers) are baked BY ADDRESS into every step that
# was captured while they were current (graph.GraphedStep / SegmentedStep).  When such a buffer is replaced later -- a call with a
# larger batch, another model growing a shared workspace -- the old block must stay allocated: a captured step that is replayed
# afterwards still reads and writes it (returned to the caching allocator it is handed to someone else; after
# torch.cuda.empty_cache() it is unmapped and the replay dies with a GPU memory fault).  So every persistent buffer that is touched
# WHILE a step is being captured is marked (note_captured), and a marked buffer that is replaced is parked here for good.
# Unmarked buffers (eager-only use) are freed as usual: alternating batch sizes in eager mode must not accumulate memory.
CAPTURING = [0]   # > 0 while graph.GraphedStep / graph.SegmentedStep capture (they bracket their capture with it)
_PARKED: list = []


def note_captured(buf) -> None:
    """Call on every use of a persistent buffer that kernels address directly (cheap: one list read outside captures)."""
    if CAPTURING[0] and buf is not None:
        buf._mh_captured = True


def park_replaced(buf) -> None:
    """Call with the OLD tensor whenever such a buffer is replaced."""
    if buf is not None and getattr(buf, "_mh_captured", False):
        _PARKED.append(buf)


def _workspace(nbytes: int, device, tag: str) -> torch.Tensor:
    """Grow-only per-(device, tag) scratch buffers (caller-provided workspaces of the C ABI)."""
    key = (str(device), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # a kernel queued on a side stream may still be using the old block: it must not go back to the caching
            # allocator (which would hand it to a launch-stream allocation) before the side streams are joined
            SIDE.retire(buf)
            park_replaced(buf)  # ... and never, if a captured step may still address it
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS[key] = buf
    note_captured(buf)
    return buf


def linear_backward(x, W, y, dy, activation=None, need_dx: bool = True, need_db: bool = True, x_activation=None,
                    zero_pad: bool = True, late_dw: Optional[list] = None):
    """Backward of ``linear``: returns (dx | None, dW, db | None).  ``dy`` is overwritten with
    dz = dy * act'(y) when an activation is given.  ``x_activation`` names the activation that
    produced ``x``: its derivative is folded into dx, which is then the producer's dz.  ``late_dw`` (a list; side-stream steps
    only): dW / db are NOT launched here -- a closure that launches them on the then-current stream is appended to the list and
    the caller runs it later in the step (blocks.DLRMBlock.backward: behind the bottom-MLP backward, beside the sparse apply)."""
    lib = _lib.load()
    _rowmajor_2d(x, "x")
    _rowmajor_2d(dy, "dy")
    M, K = x.shape
    N = W.shape[1]
    if dy.shape != (M, N):
        raise ValueError(f"dy must be [{M}, {N}]")
    if activation not in ACT:
        raise ValueError(f"unsupported activation {activation!r}")
    act = ACT[activation]
    if act != 0:
        _rowmajor_2d(y, "y")
    dx, lddx = None, K
    if need_dx:
        lddx = (K + 3) // 4 * 4  # 16-byte aligned rows for whoever consumes dx next
        buf = torch.empty((M, lddx), dtype=torch.float32, device=x.device)
        if lddx != K and zero_pad:  # consumers that widen dx to its leading dimension expect zeros there (blocks._widen)
            zero_pad_columns(buf, K)
        dx = buf[:, :K]
    dW = torch.empty((K, N), dtype=torch.float32, device=x.device)
    db = torch.empty((N,), dtype=torch.float32, device=x.device) if need_db else None
    # dX of a tower layer (N = 128, K <= 1024, large batch) in the fp32-grade six-term split (mh_tower_split.hip), like its forward
    tower_dx = need_dx and ACT[x_activation] == 0 and _tower_ok(M, K, N, dy)

    def run_tower_dx():
        ws = _workspace(lib.mh_tower_workspace_bytes(M, K, N), x.device, f"tower_{K}x{N}")
        check(lib.mh_tower_linear_dx(_ptr(dy), dy.stride(0), _ptr(W), M, K, N, _ptr(dx), lddx, _ptr(ws), ws.numel(), _stream()),
              "mh_tower_linear_dx")

    split = _linear_split_ok(M, K, N)  # same contract, split-bf16 GEMMs; the dX phase needs the workspace too
    if split:
        _sync_gemm_arith(lib)
    bwd = lib.mh_linear_bias_act_bwd_split if split else lib.mh_linear_bias_act_bwd
    nbytes = lib.mh_linear_split_workspace_bytes(M, K, N) if split else lib.mh_linear_bwd_workspace_bytes(M, K, N)
    yp, ldy = (_ptr(y), y.stride(0)) if act != 0 else (None, 0)
    if SIDE.active("dw"):
        # dz first (in place), then dX on this stream while dW / db run on the "dw" side stream
        if act != 0:
            check(lib.mh_linear_bias_act_bwd(_ptr(x), x.stride(0), None, yp, ldy, _ptr(dy), dy.stride(0), M, K, N, act,
                                             0, None, 0, None, None, None, 0, _stream()), "mh_linear_bias_act_bwd")
        # dX first, on the launch stream; dW / db start on the "dw" stream only when dX is done, so that the MFMA-bound dW
        # runs beside whatever FOLLOWS dX on the launch stream (for the top-MLP layer of a DLRM: the HBM-bound interaction
        # backward) instead of competing with dX for the matrix pipe
        def run_dx():
            if tower_dx:
                return run_tower_dx()
            if need_dx:
                wsx = _workspace(nbytes, x.device, "linear_split_dx") if split else None
                check(bwd(_ptr(x), x.stride(0), _ptr(W), None, 0, _ptr(dy), dy.stride(0), M, K, N, 0,
                          ACT[x_activation], _ptr(dx), lddx, None, None, _ptr(wsx), 0 if wsx is None else wsx.numel(), _stream()),
                      "mh_linear_bias_act_bwd")

        def run_dw():
            ws = _workspace(nbytes, x.device, "linear_bwd_side")
            with SIDE.on("dw", keep=(x, dy)):
                check(bwd(_ptr(x), x.stride(0), None, None, 0, _ptr(dy), dy.stride(0), M, K, N, 0, 0,
                          None, 0, _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                      "mh_linear_bias_act_bwd")

        # (round 4 re-measured the alternatives on one box -- dW forked before dX, dW on its own stream, both, three streams: all
        # 15-25 us slower per step than dX first and dW behind it on the shared side stream: profiles/r4_ab_side_streams.txt)
        run_dx()
        if late_dw is not None:
            def run_dw_late():
                ws = _workspace(nbytes, x.device, "linear_bwd_late")
                check(bwd(_ptr(x), x.stride(0), None, None, 0, _ptr(dy), dy.stride(0), M, K, N, 0, 0,
                          None, 0, _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                      "mh_linear_bias_act_bwd")

            late_dw.append(run_dw_late)
        else:
            run_dw()
        SIDE.maybe_join()
        return dx, dW, db
    ws = _workspace(nbytes, x.device, "linear_bwd")
    if tower_dx:  # dz in place, dX by the tower kernel, dW / db by the exact-chain kernels
        with _timed(f"linear_bwd_{K}x{N}", nbytes=4 * (2 * M * K + 2 * K * N + 2 * M * N), flops=4 * M * K * N):
            if act != 0:
                check(lib.mh_linear_bias_act_bwd(_ptr(x), x.stride(0), None, yp, ldy, _ptr(dy), dy.stride(0), M, K, N, act,
                                                 0, None, 0, None, None, None, 0, _stream()), "mh_linear_bias_act_bwd")
            run_tower_dx()
            check(bwd(_ptr(x), x.stride(0), None, None, 0, _ptr(dy), dy.stride(0), M, K, N, 0, 0, None, 0, _ptr(dW), _ptr(db),
                      _ptr(ws), ws.numel(), _stream()), "mh_linear_bias_act_bwd")
        return dx, dW, db
    with _timed(f"linear_bwd_{K}x{N}", nbytes=4 * (2 * M * K + 2 * K * N + 2 * M * N), flops=(4 if need_dx else 2) * M * K * N):
        check(
            bwd(_ptr(x), x.stride(0), _ptr(W), yp, ldy, _ptr(dy), dy.stride(0), M, K, N, act,
                ACT[x_activation], _ptr(dx), lddx, _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), _stream()),
            "mh_linear_bias_act_bwd",
        )
    return dx, dW, db


def dot_interaction_backward(x: torch.Tensor, dout: torch.Tensor, tail_slot: int = -1, tail_width: int = 0,
                             tail_first: bool = True):
    """dx[B,F,D] of ``dot_interaction``; the gradient of the shortcut columns is folded into slot ``tail_slot``."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    _rowmajor_2d(dout, "dout")
    B, F, D = x.shape
    dx = torch.empty_like(x)
    with _timed("dot_interaction_bwd", nbytes=B * (2 * F * D + dout.shape[1]) * 4, flops=2 * B * D * F * (F - 1)):
        check(
            lib.mh_dot_interaction_bwd(_ptr(x), _ptr(dout), dout.stride(0), B, F, D, _ptr(dx), tail_slot,
                                       tail_width, 1 if tail_first else 0, _stream()),
            "mh_dot_interaction_bwd",
        )
    return dx


class PreparedSparseUpdate:
    """Handle of ``embedding_gather_backward_prepare``: the sorted ids / piece list live in ``ws`` until the matching
    ``embedding_gather_backward(..., prepared=handle)``; ``event`` marks the end of the preparation on its stream."""

    def __init__(self, ws, key, event):
        self.ws, self.key, self.event = ws, key, event


def _sparse_key(tables, ids, B, D):
    """What the prepared half depends on: the id buffers, the row counts and WHICH features share a table (the segments of
    the sort) -- not the table addresses themselves: the apply may target other buffers of the same shapes (the dense
    gradient accumulators of replicated tables in the data-parallel step)."""
    ptrs = [t.data_ptr() for t in tables]
    share = tuple(ptrs.index(p) for p in ptrs)
    return (share, tuple(int(t.shape[0]) for t in tables), tuple(i.data_ptr() for i in ids), int(B), int(D))


_DET = [None]


def _sync_deterministic(lib) -> None:
    """MERLIN_HIP_DETERMINISTIC is read HERE (a dict lookup per call) and handed to the library when it changes: tests flip it
    inside one process; the library itself no longer calls getenv on its launch path."""
    import os

    want = 1 if os.environ.get("MERLIN_HIP_DETERMINISTIC") == "1" else 0
    if _DET[0] != want:
        lib.mh_set_deterministic(want)
        _DET[0] = want


def embedding_gather_backward_prepare(tables: Sequence[torch.Tensor], ids: Sequence[torch.Tensor],
                                      tag: str = "") -> Optional[PreparedSparseUpdate]:
    """The id-only half of ``embedding_gather_backward`` (segmented sort + piece list) on the CURRENT stream -- call it on a
    side stream at the start of the step so that it runs beside the forward pass.  The handle owns the workspace named by
    ``tag`` (two updates prepared in the same step need two tags)."""
    lib = _lib.load()
    _sync_deterministic(lib)
    F = len(tables)
    if F == 0 or F > _lib.MAX_FEATURES - 1:
        return None
    flat = [i.reshape(-1) for i in ids]
    B, D = flat[0].shape[0], tables[0].shape[1]
    if B == 0 or any(i.shape[0] != B or not i.is_contiguous() or i.dtype != flat[0].dtype for i in flat):
        return None
    idt = _ids_dtype(flat[0], "ids[0]")
    nbytes = lib.mh_embedding_bwd_workspace_bytes(B, F, D)
    if nbytes < 0:
        return None
    ws = _workspace(nbytes, tables[0].device, "embedding_bwd_prepared" + tag)
    tab = _host_ptr_array([w.data_ptr() for w in tables])
    idp = _host_ptr_array([i.data_ptr() for i in flat])
    rows = (C.c_int64 * F)(*[w.shape[0] for w in tables])
    check(lib.mh_embedding_gather_bwd_prepare(tab, rows, idp, idt, B, F, D, _ptr(ws), ws.numel(), _stream()),
          "mh_embedding_gather_bwd_prepare")
    return PreparedSparseUpdate(ws, _sparse_key(tables, flat, B, D), SIDE.mark())


def embedding_gather_backward(tables: Sequence[torch.Tensor], states: Optional[Sequence[Optional[torch.Tensor]]],
                              ids: Sequence[torch.Tensor], grad: torch.Tensor, grad_offset: Sequence[int],
                              optimizer: str = "sgd", lr: float = 0.01, eps: float = 1e-7,
                              states2: Optional[Sequence[torch.Tensor]] = None, beta1: float = 0.9, beta2: float = 0.999,
                              lr_device: Optional[torch.Tensor] = None, prepared: Optional[PreparedSparseUpdate] = None) -> None:
    """Fused backward + sparse optimizer step for the one-hot lookup.  ``grad`` is a contiguous
    ``[B, ...]`` buffer; feature f's gradient row starts ``grad_offset[f]`` floats into row b.
    ``prepared``: handle of ``embedding_gather_backward_prepare`` for the SAME tables / ids (checked): only the
    gradient-dependent half runs, after waiting for the preparation's event."""
    lib = _lib.load()
    _sync_deterministic(lib)
    F = len(tables)
    if F == 0:
        return
    if F > _lib.MAX_FEATURES - 1:
        raise ValueError(f"at most {_lib.MAX_FEATURES - 1} features per backward call")
    _dev(grad, "grad", torch.float32)
    if not grad.is_contiguous():
        raise ValueError("grad must be contiguous")
    B = grad.shape[0]
    if B == 0:
        return  # no gradient rows (a rank that owns none of a skewed batch's requested rows): nothing to update
    D = tables[0].shape[1]
    row_stride = grad.numel() // max(B, 1)
    idt = _ids_dtype(ids[0], "ids[0]")
    flat = [i.reshape(-1) for i in ids]
    tab = _host_ptr_array([w.data_ptr() for w in tables])
    st = st2 = None
    if optimizer in ("adagrad", "adam", "lazy_adam"):
        if states is None or any(s is None for s in states):
            raise ValueError(f"{optimizer} needs a state tensor per table")
        st = _host_ptr_array([s.data_ptr() for s in states])
    if optimizer in ("adam", "lazy_adam"):
        if states2 is None or any(s is None for s in states2):
            raise ValueError("adam needs a second-moment tensor per table")
        st2 = _host_ptr_array([s.data_ptr() for s in states2])
    idp = _host_ptr_array([i.data_ptr() for i in flat])
    rows = (C.c_int64 * F)(*[w.shape[0] for w in tables])
    slot = (C.c_int64 * F)(*[int(s) for s in grad_offset])
    nbytes = lib.mh_embedding_bwd_workspace_bytes(B, F, D)
    if nbytes < 0:
        raise _lib.MerlinHipError("mh_embedding_bwd_workspace_bytes failed")
    if prepared is not None and prepared.key == _sparse_key(tables, flat, B, D) and prepared.ws.numel() >= nbytes:
        SIDE.wait(prepared.event)
        ws = prepared.ws
        with _timed("embedding_bwd_apply", nbytes=B * F * (_OPT_ROW_PASSES[optimizer] * D * 4 + flat[0].element_size())):
            check(
                lib.mh_embedding_gather_bwd_apply(tab, st, rows, idp, idt, B, F, D, _ptr(grad), row_stride, slot,
                                                  _lib.OPT[optimizer], lr, eps, st2, beta1, beta2, _ptr(lr_device), _ptr(ws),
                                                  ws.numel(), _stream()),
                "mh_embedding_gather_bwd_apply",
            )
        return
    ws = _workspace(nbytes, grad.device, "embedding_bwd")
    with _timed("embedding_bwd", nbytes=B * F * (_OPT_ROW_PASSES[optimizer] * D * 4 + flat[0].element_size())):
        check(
            lib.mh_embedding_gather_bwd(tab, st, rows, idp, idt, B, F, D, _ptr(grad), row_stride, slot,
                                        _lib.OPT[optimizer], lr, eps, st2, beta1, beta2, _ptr(lr_device), _ptr(ws), ws.numel(),
                                        _stream()),
            "mh_embedding_gather_bwd",
        )


def embedding_bag_expand(table: torch.Tensor, values: torch.Tensor, offsets: Optional[torch.Tensor], grad: torch.Tensor,
                         combiner: str = "mean") -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-value gradient rows of a list lookup: returns ``(values [nnz], gexp [nnz, D])`` with
    ``gexp[j] = grad[bag(j)] / div(bag(j))`` (the IndexedSlices of the lookup's table gradient, before any dedup)."""
    lib = _lib.load()
    _dev(table, "table", torch.float32)
    _dev(values, "values")
    _dev(grad, "grad", torch.float32)
    idt = _ids_dtype(values, "values")
    if combiner not in COMBINER:
        raise ValueError(f"combiner must be one of {sorted(COMBINER)}, got {combiner!r}")
    D = table.shape[1]
    if grad.dim() != 2 or grad.shape[1] != D or grad.stride(1) != 1:
        raise ValueError(f"grad must be [B, {D}] with unit inner stride")
    B, L = grad.shape[0], 0
    if offsets is None:
        if values.dim() == 3 and values.shape[-1] == 1:
            values = values.squeeze(-1)
        if values.dim() != 2 or values.shape[0] != B:
            raise ValueError("dense list values must be [B, L]")
        L = values.shape[1]
    else:
        _dev(offsets, "offsets")
        if offsets.dtype != values.dtype:
            raise TypeError("offsets and values must share one integer dtype")
        offsets = offsets.reshape(-1).contiguous()
    values = values.reshape(-1).contiguous()
    nnz = values.shape[0]
    gexp = torch.empty((nnz, D), dtype=torch.float32, device=grad.device)
    if B == 0 or nnz == 0:
        return values, gexp
    scale = torch.empty((B,), dtype=torch.float32, device=grad.device)
    check(lib.mh_embedding_bag_expand(_ptr(table), table.shape[0], _ptr(values), nnz, _ptr(offsets), L, idt, B, D,
                                      COMBINER[combiner], _ptr(grad), grad.stride(0), _ptr(gexp), _ptr(scale), _stream()),
          "mh_embedding_bag_expand")
    return values, gexp


def embedding_bag_backward(table: torch.Tensor, state: Optional[torch.Tensor], values: torch.Tensor,
                           offsets: Optional[torch.Tensor], grad: torch.Tensor, combiner: str = "mean",
                           optimizer: str = "sgd", lr: float = 0.01, eps: float = 1e-7,
                           state2: Optional[torch.Tensor] = None, beta1: float = 0.9, beta2: float = 0.999,
                           lr_device: Optional[torch.Tensor] = None) -> None:
    """Fused backward + sparse optimizer step of ``embedding_bag`` (``offsets`` given, CSR) or
    ``embedding_dense_list`` (``offsets=None``, ``values`` is ``[B, L]``).  ``grad`` is ``[B, D]`` with unit
    inner stride (a column slice of a wider buffer is fine)."""
    lib = _lib.load()
    _dev(table, "table", torch.float32)
    _dev(values, "values")
    _dev(grad, "grad", torch.float32)
    idt = _ids_dtype(values, "values")
    if combiner not in COMBINER:
        raise ValueError(f"combiner must be one of {sorted(COMBINER)}, got {combiner!r}")
    D = table.shape[1]
    if grad.dim() != 2 or grad.shape[1] != D or grad.stride(1) != 1:
        raise ValueError(f"grad must be [B, {D}] with unit inner stride")
    B = grad.shape[0]
    L = 0
    if offsets is None:
        if values.dim() == 3 and values.shape[-1] == 1:
            values = values.squeeze(-1)
        if values.dim() != 2 or values.shape[0] != B:
            raise ValueError("dense list values must be [B, L]")
        L = values.shape[1]
    else:
        _dev(offsets, "offsets")
        if offsets.dtype != values.dtype:
            raise TypeError("offsets and values must share one integer dtype")
        offsets = offsets.reshape(-1).contiguous()
        if offsets.shape[0] != B + 1:
            raise ValueError(f"offsets must have B + 1 = {B + 1} entries")
    values = values.reshape(-1).contiguous()
    nnz = values.shape[0]
    if B == 0 or nnz == 0:
        return
    if optimizer in ("adagrad", "adam", "lazy_adam") and state is None:
        raise ValueError(f"{optimizer} needs a state tensor")
    if optimizer in ("adam", "lazy_adam") and state2 is None:
        raise ValueError("adam needs a second-moment tensor")
    nbytes = lib.mh_embedding_bag_bwd_workspace_bytes(B, nnz, D)
    if nbytes < 0:
        raise _lib.MerlinHipError("mh_embedding_bag_bwd_workspace_bytes failed")
    ws = _workspace(nbytes, grad.device, "embedding_bag_bwd")
    with _timed("embedding_bag_bwd"):
        check(
            lib.mh_embedding_bag_bwd(_ptr(table), _ptr(state), _ptr(state2), table.shape[0], _ptr(values), nnz,
                                     _ptr(offsets), L, idt, B, D, COMBINER[combiner], _ptr(grad), grad.stride(0),
                                     _lib.OPT[optimizer], lr, eps, beta1, beta2, _ptr(lr_device), _ptr(ws), ws.numel(),
                                     _stream()),
            "mh_embedding_bag_bwd",
        )


def embedding_bag_backward_multi(tables: Sequence[torch.Tensor], states: Optional[Sequence[Optional[torch.Tensor]]],
                                 values: Sequence[torch.Tensor], offsets: Optional[Sequence[torch.Tensor]], grad: torch.Tensor,
                                 grad_offset: Sequence[int], combiner: str = "mean", optimizer: str = "sgd", lr: float = 0.01,
                                 eps: float = 1e-7, states2: Optional[Sequence[torch.Tensor]] = None, beta1: float = 0.9,
                                 beta2: float = 0.999, lr_device: Optional[torch.Tensor] = None) -> None:
    """``embedding_bag_backward`` of F list features over F distinct tables of one width in ONE sparse update
    (``mh_embedding_bag_bwd_multi``): ``grad`` is a contiguous ``[B, ...]`` buffer, feature f's ``[B, D]`` block starts
    ``grad_offset[f]`` floats into row b.  ``offsets=None``: every ``values[f]`` is a dense ``[B, L]`` list of one L.
    The same sums as F single-feature calls (a run of equal ids may be cut into partial sums at other places: a few ulp); no
    ``[nnz, D]`` expanded gradient is materialised."""
    lib = _lib.load()
    _sync_deterministic(lib)
    F = len(tables)
    if F == 0:
        return
    if F > _lib.MAX_FEATURES - 1:
        raise ValueError(f"at most {_lib.MAX_FEATURES - 1} features per backward call")
    if combiner not in COMBINER or combiner == "max":
        raise ValueError(f"combiner must be sum, mean or sqrtn, got {combiner!r}")
    if len(values) != F or len(grad_offset) != F or (offsets is not None and len(offsets) != F):
        raise ValueError("tables, values, offsets and grad_offset must have one entry per feature")
    _dev(grad, "grad", torch.float32)
    if not grad.is_contiguous():
        raise ValueError("grad must be contiguous")
    B = grad.shape[0]
    if B == 0:
        return
    D = tables[0].shape[1]
    row_stride = grad.numel() // B
    idt = _ids_dtype(values[0], "values[0]")
    for f, w in enumerate(tables):
        _dev(w, f"tables[{f}]", torch.float32)
        if w.shape[1] != D:
            raise ValueError("all tables of one call share the embedding width")
        if values[f].dtype != values[0].dtype:
            raise TypeError("all value lists share one integer dtype")
        if int(grad_offset[f]) < 0 or int(grad_offset[f]) + D > row_stride:
            raise ValueError(f"grad_offset[{f}] + D exceeds the gradient row")
    if len({w.data_ptr() for w in tables}) != F:
        raise ValueError("the features of one call use distinct tables (shared tables: embedding_bag_expand + one gather update)")
    L = 0
    if offsets is None:
        vals = []
        for v in values:
            if v.dim() == 3 and v.shape[-1] == 1:
                v = v.squeeze(-1)
            if v.dim() != 2 or v.shape[0] != B:
                raise ValueError("dense list values must be [B, L]")
            if L and v.shape[1] != L:
                raise ValueError("dense lists of one call share the list length")
            L = v.shape[1]
            vals.append(_dev(v, "values").reshape(-1).contiguous())
        offs = None
    else:
        vals = [_dev(v, "values").reshape(-1).contiguous() for v in values]
        offs = []
        for o in offsets:
            _dev(o, "offsets")
            if o.dtype != values[0].dtype:
                raise TypeError("offsets and values must share one integer dtype")
            o = o.reshape(-1).contiguous()
            if o.shape[0] != B + 1:
                raise ValueError(f"offsets must have B + 1 = {B + 1} entries")
            offs.append(o)
    st = st2 = None
    if optimizer in ("adagrad", "adam", "lazy_adam"):
        if states is None or any(s is None for s in states):
            raise ValueError(f"{optimizer} needs a state tensor per table")
        st = _host_ptr_array([s.data_ptr() for s in states])
    if optimizer in ("adam", "lazy_adam"):
        if states2 is None or any(s is None for s in states2):
            raise ValueError("adam needs a second-moment tensor per table")
        st2 = _host_ptr_array([s.data_ptr() for s in states2])
    nnz = [int(v.shape[0]) for v in vals]
    if max(nnz) == 0:
        return
    tab = _host_ptr_array([w.data_ptr() for w in tables])
    vp = _host_ptr_array([v.data_ptr() if v.numel() else 0 for v in vals])
    op = None if offs is None else _host_ptr_array([o.data_ptr() for o in offs])
    rows = (C.c_int64 * F)(*[w.shape[0] for w in tables])
    nz = (C.c_int64 * F)(*nnz)
    slot = (C.c_int64 * F)(*[int(s) for s in grad_offset])
    nbytes = lib.mh_embedding_bag_bwd_multi_workspace_bytes(B, max(nnz), F, D)
    if nbytes < 0:
        raise _lib.MerlinHipError("mh_embedding_bag_bwd_multi_workspace_bytes failed")
    ws = _workspace(nbytes, grad.device, "embedding_bag_bwd_multi")
    with _timed("embedding_bag_bwd_multi"):
        check(
            lib.mh_embedding_bag_bwd_multi(tab, st, st2, rows, vp, nz, op, L, idt, B, F, D, COMBINER[combiner], _ptr(grad),
                                           row_stride, slot, _lib.OPT[optimizer], lr, eps, beta1, beta2, _ptr(lr_device),
                                           _ptr(ws), ws.numel(), _stream()),
            "mh_embedding_bag_bwd_multi",
        )


def l2_batch_reg(out: torch.Tensor, grad: Optional[torch.Tensor], factor: float, loss_accum: torch.Tensor) -> None:
    """``loss_accum += factor * sum(out^2)`` and ``grad += 2 * factor * out`` for [B, D] views with unit inner stride
    (``mh_l2_batch_reg``: the l2_batch_regularization term of an embedding lookup)."""
    lib = _lib.load()
    _dev(out, "out", torch.float32)
    _dev(loss_accum, "loss_accum", torch.float32)
    if out.dim() != 2 or out.stride(1) != 1 or (grad is not None and (grad.shape != out.shape or grad.stride(1) != 1)):
        raise ValueError("l2_batch_reg: out / grad must be [B, D] views with unit inner stride")
    ws = _workspace(1024, out.device, "l2_batch_reg")
    check(lib.mh_l2_batch_reg(_ptr(out), out.stride(0), _ptr(grad), 0 if grad is None else grad.stride(0), out.shape[0],
                              out.shape[1], float(factor), _ptr(loss_accum), _ptr(ws), _stream()), "mh_l2_batch_reg")


def bce(p: torch.Tensor, label: torch.Tensor, need_grad: bool = True):
    """Mean binary cross-entropy of probabilities ``p`` (Keras semantics) and d(mean)/d(logit)."""
    lib = _lib.load()
    _dev(p, "p", torch.float32)
    _dev(label, "label", torch.float32)
    p, label = p.reshape(-1), label.reshape(-1)
    M = p.shape[0]
    dlogit = torch.empty_like(p) if need_grad else None
    mean = torch.empty(1, dtype=torch.float32, device=p.device)
    if M == 0:
        mean.fill_(float("nan"))
        return mean[0], (None if dlogit is None else dlogit.reshape(-1, 1))
    ws = _workspace(1024, p.device, "bce")
    if TAIL[0] is not None and not TIMER.enabled:
        check(lib.mh_bce_mean_partial(_ptr(p), _ptr(label), M, 1.0 / M, _ptr(dlogit), _ptr(ws), _stream()), "mh_bce_mean_partial")
        TAIL[0].append(lambda: check(lib.mh_bce_mean_finish(_ptr(ws), M, _ptr(mean), _stream()), "mh_bce_mean_finish"))
        return mean[0], (None if dlogit is None else dlogit.reshape(-1, 1))
    with _timed("bce"):
        check(lib.mh_bce_mean_fwd_bwd(_ptr(p), _ptr(label), M, 1.0 / M, _ptr(mean), _ptr(dlogit), _ptr(ws), _stream()),
              "mh_bce_mean_fwd_bwd")
    return mean[0], (None if dlogit is None else dlogit.reshape(-1, 1))


ACTX = {"tanh": 10, "elu": 11, "selu": 12, "softplus": 13, "swish": 14, "silu": 14, "gelu": 15, "leaky_relu": 16, "relu6": 17}


def activation(x: torch.Tensor, name: str, dy: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``f(x)`` (``dy`` None) or ``dy * f'(x)`` for the element-wise activations of ``ACTX`` (``mh_activation``); x is the
    layer's input (pre-activation) in both directions.  2-D row-major operands."""
    lib = _lib.load()
    if name not in ACTX:
        raise ValueError(f"unknown activation {name!r} (element-wise layer: {sorted(ACTX)})")
    _rowmajor_2d(x, "x")
    if dy is not None:
        _rowmajor_2d(dy, "dy")
        if dy.shape != x.shape:
            raise ValueError("activation: dy must have the shape of x")
    M, N = x.shape
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(lib.mh_activation(ACTX[name], _ptr(x), x.stride(0), _ptr(dy), 0 if dy is None else dy.stride(0), _ptr(out), N, M, N,
                            _stream()), "mh_activation")
    return out


def zero_pad_columns(buf: torch.Tensor, width: int) -> None:
    """``buf[:, width:] = 0`` for a row-major fp32 [M, ld] buffer (``mh_fill_columns``): a library launch, so a recorded step
    (graph.RecordedStep) holds it -- a torch fill would be missing from the replay."""
    ld = buf.stride(0) if buf.dim() == 2 and buf.shape[0] > 1 else buf.shape[-1]
    if buf.dim() != 2 or buf.dtype != torch.float32 or buf.stride(1) != 1 or width > buf.shape[1]:
        raise ValueError("zero_pad_columns wants a row-major fp32 [M, ld] buffer")
    if buf.shape[1] == width or buf.shape[0] == 0:
        return
    _dev(buf, "buf", torch.float32)
    check(_lib.load().mh_fill_columns(_ptr(buf), buf.shape[0], ld, width, buf.shape[1] - width, 0.0, _stream()), "mh_fill_columns")


def mean(x: torch.Tensor) -> torch.Tensor:
    """Mean of a contiguous fp32 tensor as a 0-d device tensor (``mh_mean``: deterministic two-stage sum; replaces the
    torch reduction kernel that used to run once per retrieval train step)."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    if not x.is_contiguous():
        raise ValueError("mean: x must be contiguous")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    if x.numel() == 0:
        out.fill_(float("nan"))
        return out[0]
    ws = _workspace(1024, x.device, "mean")
    check(lib.mh_mean(_ptr(x), x.numel(), _ptr(out), _ptr(ws), _stream()), "mh_mean")
    return out[0]


def bce_per_sample(p: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """Per-sample binary cross-entropy (``mh_bce_fwd_bwd`` without the gradient)."""
    lib = _lib.load()
    _dev(p, "p", torch.float32)
    _dev(label, "label", torch.float32)
    p, label = p.reshape(-1), label.reshape(-1)
    loss = torch.empty_like(p)
    if p.shape[0]:
        check(lib.mh_bce_fwd_bwd(_ptr(p), _ptr(label), p.shape[0], 1.0, _ptr(loss), None, _stream()), "mh_bce_fwd_bwd")
    return loss


def l2norm(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """tf.linalg.l2_normalize(x, axis=-1) (epsilon = 1e-12 on the squared norm)."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib.mh_l2norm_rows(_ptr(x), x.shape[0], x.shape[1], eps, _ptr(y), _stream()), "mh_l2norm_rows")
    return y


def l2norm_backward(x: torch.Tensor, dy: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Gradient of ``l2norm`` w.r.t. its input ``x``."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    _dev(dy, "dy", torch.float32)
    x, dy = x.contiguous(), dy.contiguous()
    if x.shape != dy.shape or x.dim() != 2:
        raise ValueError("l2norm_backward: x and dy must be [M, N]")
    dx = torch.empty_like(x)
    check(lib.mh_l2norm_rows_bwd(_ptr(x), _ptr(dy), x.shape[0], x.shape[1], eps, _ptr(dx), _stream()), "mh_l2norm_rows_bwd")
    return dx


def rowwise_dot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """sum(a * b, -1, keepdims=True)."""
    lib = _lib.load()
    _rowmajor_2d(a, "a")
    _rowmajor_2d(b, "b")
    if a.shape != b.shape:
        raise ValueError("rowwise_dot: shape mismatch")
    out = torch.empty((a.shape[0], 1), dtype=torch.float32, device=a.device)
    check(lib.mh_rowwise_dot(_ptr(a), a.stride(0), _ptr(b), b.stride(0), a.shape[0], a.shape[1], _ptr(out), _stream()),
          "mh_rowwise_dot")
    return out


def dense_optimizer_step(opt, p) -> None:
    """In-place SGD / Adagrad step on a dense Parameter (``p.grad`` must be set)."""
    lib = _lib.load()
    state = None
    if opt.name == "adagrad":
        state = p.state.get("accumulator")
        if state is None:
            state = torch.full_like(p.data, opt.initial_accumulator_value)
            p.state["accumulator"] = state
    state2 = None
    if opt.name == "adam":
        state, state2 = _adam_states(p)
    g = p.grad.contiguous()
    check(lib.mh_dense_optimizer_step(_ptr(p.data), _ptr(g), _ptr(state), p.data.numel(), _lib.OPT[opt.name],
                                      opt.learning_rate, opt.epsilon, _ptr(state2), opt.beta_1, opt.beta_2,
                                      _ptr(opt.lr_device), _stream()), "mh_dense_optimizer_step")


def _adam_states(p):
    if "m" not in p.state:
        p.state["m"] = torch.zeros_like(p.data)
        p.state["v"] = torch.zeros_like(p.data)
    return p.state["m"], p.state["v"]


def adam_tick(opt) -> None:
    """Advance Adam's on-device step counter and bias-corrected learning rate (one tiny launch)."""
    lib = _lib.load()
    check(lib.mh_adam_tick(_ptr(opt._step_dev), opt.learning_rate, opt.beta_1, opt.beta_2, _ptr(opt.lr_device), _stream()),
          "mh_adam_tick")


# --------------------------------------------------------------------------------------------
# retrieval: scorer, top-k, cross
# --------------------------------------------------------------------------------------------
class ScorerResult(NamedTuple):
    logits: Optional[torch.Tensor]  # [B, 1 + Nn] or None (fused mode)
    loss: torch.Tensor  # [B]   logsumexp(logits) - logits[:, 0]
    lse: torch.Tensor  # [B]


def _ids_pair(pos_ids, neg_ids):
    if pos_ids is None or neg_ids is None:
        return None, None, MH_I32
    _dev(pos_ids, "pos_ids")
    _dev(neg_ids, "neg_ids")
    pos_ids, neg_ids = pos_ids.reshape(-1).contiguous(), neg_ids.reshape(-1).contiguous()
    if neg_ids.dtype != pos_ids.dtype:
        neg_ids = neg_ids.to(pos_ids.dtype)  # reference casts positive ids to the negatives' dtype
    return pos_ids, neg_ids, _ids_dtype(pos_ids, "pos_ids")


def _logq_pair(pos_logq, neg_logq, B, Nn):
    """(pos_logq[B], neg_logq[Nn]) fp32 device vectors or (None, None); see include/merlin_hip.h."""
    if pos_logq is None and neg_logq is None:
        return None, None
    if pos_logq is None or neg_logq is None:
        raise ValueError("logQ correction needs the log-probabilities of BOTH the positive and the negative candidates")
    pos_logq = _dev(pos_logq, "pos_logq", torch.float32).reshape(-1).contiguous()
    neg_logq = _dev(neg_logq, "neg_logq", torch.float32).reshape(-1).contiguous()
    if pos_logq.shape[0] != B or neg_logq.shape[0] != Nn:
        raise ValueError(f"pos_logq / neg_logq must have {B} / {Nn} entries")
    return pos_logq, neg_logq


_ARITH = [None]


_SCORER_MODES = {"f32": 0, "bf16x3": 1, "bf16x6": 2}


def scorer_arith() -> str:
    """Arithmetic of the in-batch scorer's products at E = 128 and E = 64 (``MERLIN_HIP_SCORER_ARITH``; bf16x3: 128 only): "bf16x6" (default) -- the fp32-grade
    six-term split on the bf16 MFMA (every product from h h + h m + m h + h l + l h + m m, dropped terms <= 2^-25 of it); "f32" -- the
    exact fp32 MFMA chains; "bf16x3" -- the opt-in three-term split (2^-17 per operand, NOT fp32-grade, own dtype label)."""
    v = os.environ.get("MERLIN_HIP_SCORER_ARITH", "bf16x6")
    return v if v in _SCORER_MODES else "bf16x6"


def _sync_scorer_arith(lib) -> None:
    """The switch is read HERE (a dict lookup per call) and handed to the library when it changes (``mh_set_scorer_arith``)."""
    want = _SCORER_MODES[scorer_arith()]
    if _ARITH[0] != want:
        check(lib.mh_set_scorer_arith(want), "mh_set_scorer_arith")
        _ARITH[0] = want


def inbatch_softmax(q, item, neg_item, pos_ids=None, neg_ids=None, temperature: float = 1.0,
                    false_neg_score: float = -655.04, materialize: bool = True, pos_logq=None, neg_logq=None,
                    logq_after_mask: bool = False) -> ScorerResult:
    """Sampled-softmax scorer: positives ``<q, item>`` in column 0, negatives ``q @ neg_item^T``
    with false negatives rescored, temperature scaling and the softmax cross-entropy fused."""
    lib = _lib.load()
    for n_, t in (("q", q), ("item", item), ("neg_item", neg_item)):
        _dev(t, n_, torch.float32)
        if t.dim() != 2 or not t.is_contiguous():
            raise ValueError(f"{n_} must be contiguous 2-D")
    B, E = q.shape
    Nn = neg_item.shape[0]
    pos_ids, neg_ids, idt = _ids_pair(pos_ids, neg_ids)
    pos_logq, neg_logq = _logq_pair(pos_logq, neg_logq, B, Nn)
    logits = torch.empty((B, Nn + 1), dtype=torch.float32, device=q.device) if materialize else None
    loss = torch.empty((B,), dtype=torch.float32, device=q.device)
    lse = torch.empty((B,), dtype=torch.float32, device=q.device)
    ws = _workspace(lib.mh_inbatch_softmax_workspace_bytes(B, Nn, E, 0), q.device, "scorer_fwd")
    _sync_scorer_arith(lib)
    with _timed("inbatch_softmax_fwd", nbytes=4 * (2 * B + Nn) * E, flops=2 * B * Nn * E + 2 * B * E):
        check(
            lib.mh_inbatch_softmax_fwd(_ptr(q), _ptr(item), _ptr(neg_item), _ptr(pos_ids), _ptr(neg_ids), idt, B, Nn, E,
                                       temperature, false_neg_score, _ptr(pos_logq), _ptr(neg_logq), int(logq_after_mask),
                                       _ptr(logits), Nn + 1, _ptr(loss), _ptr(lse),
                                       _ptr(ws), ws.numel(), _stream()),
            "mh_inbatch_softmax_fwd",
        )
    return ScorerResult(logits, loss, lse)


def inbatch_softmax_train(q, item, neg_item, pos_ids=None, neg_ids=None, temperature: float = 1.0,
                          false_neg_score: float = -655.04, grad_scale: Optional[float] = None, pos_logq=None,
                          neg_logq=None, logq_after_mask: bool = False):
    """Training-mode forward (``mh_inbatch_softmax_fwd_dq``): one pass over the score tiles yields the per-row
    loss / lse AND dq, ditem (positive role) of ``grad_scale * sum_b loss[b]`` (default 1/B).  Returns
    ``(ScorerResult(None, loss, lse), dq, ditem)``; follow with ``inbatch_softmax_backward(..., need_dq=False)``
    for dneg.  Falls back to forward + full backward when E > 128."""
    lib = _lib.load()
    for n_, t in (("q", q), ("item", item), ("neg_item", neg_item)):
        _dev(t, n_, torch.float32)
        if t.dim() != 2 or not t.is_contiguous():
            raise ValueError(f"{n_} must be contiguous 2-D")
    B, E = q.shape
    Nn = neg_item.shape[0]
    if E > 128:
        return None
    pos_ids, neg_ids, idt = _ids_pair(pos_ids, neg_ids)
    pos_logq, neg_logq = _logq_pair(pos_logq, neg_logq, B, Nn)
    loss = torch.empty((B,), dtype=torch.float32, device=q.device)
    lse = torch.empty((B,), dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q)
    ditem = torch.empty_like(item)
    ws = _workspace(lib.mh_inbatch_softmax_workspace_bytes(B, Nn, E, 2), q.device, "scorer_fwd_dq")
    _sync_scorer_arith(lib)
    with _timed("inbatch_softmax_fwd_dq", nbytes=4 * (4 * B + Nn) * E, flops=4 * B * Nn * E):
        check(
            lib.mh_inbatch_softmax_fwd_dq(_ptr(q), _ptr(item), _ptr(neg_item), _ptr(pos_ids), _ptr(neg_ids), idt, B, Nn, E,
                                          temperature, false_neg_score, _ptr(pos_logq), _ptr(neg_logq), int(logq_after_mask),
                                          1.0 / B if grad_scale is None else grad_scale,
                                          _ptr(loss), _ptr(lse), _ptr(dq), _ptr(ditem), _ptr(ws), ws.numel(), _stream()),
            "mh_inbatch_softmax_fwd_dq",
        )
    return ScorerResult(None, loss, lse), dq, ditem


def inbatch_softmax_backward(q, item, neg_item, lse, pos_ids=None, neg_ids=None, temperature: float = 1.0,
                             false_neg_score: float = -655.04, grad_scale: Optional[float] = None, need_dq: bool = True,
                             pos_logq=None, neg_logq=None, logq_after_mask: bool = False):
    """Gradients of ``grad_scale * sum_b loss[b]`` (default 1/B: the Keras mean): (dq, ditem, dneg).
    ``need_dq=False`` runs the column pass only and returns ``(None, None, dneg)``."""
    lib = _lib.load()
    B, E = q.shape
    Nn = neg_item.shape[0]
    if not need_dq and E > 128:
        raise ValueError("need_dq=False requires E <= 128 (the streaming scorer)")
    pos_ids, neg_ids, idt = _ids_pair(pos_ids, neg_ids)
    pos_logq, neg_logq = _logq_pair(pos_logq, neg_logq, B, Nn)
    dq = torch.empty_like(q) if need_dq else None
    ditem = torch.empty_like(item) if need_dq else None
    dneg = torch.empty_like(neg_item)
    ws = _workspace(lib.mh_inbatch_softmax_workspace_bytes(B, Nn, E, 1), q.device, "scorer_bwd")
    _sync_scorer_arith(lib)
    with _timed("inbatch_softmax_bwd", nbytes=4 * (3 * B + 2 * Nn) * E, flops=(8 if dq is not None else 4) * B * Nn * E):
        check(
            lib.mh_inbatch_softmax_bwd(_ptr(q), _ptr(item), _ptr(neg_item), _ptr(pos_ids), _ptr(neg_ids), idt, B, Nn, E,
                                       temperature, false_neg_score, _ptr(pos_logq), _ptr(neg_logq), int(logq_after_mask),
                                       _ptr(lse), 1.0 / B if grad_scale is None else grad_scale,
                                       _ptr(dq), _ptr(ditem), _ptr(dneg), _ptr(ws), ws.numel(), _stream()),
            "mh_inbatch_softmax_bwd",
        )
    return dq, ditem, dneg


class TopKSplit:
    """The bf16 (hi, lo) split of a candidate matrix (``mh_topk_split``): what ``BruteForce.index`` keeps beside the fp32 rows so
    that ``mh_topk_dot_split`` can run its threshold filter on the bf16 matrix pipe.  ``norm2_max``: device float, max |row|^2."""

    def __init__(self, candidates: torch.Tensor):
        lib = _lib.load()
        _dev(candidates, "candidates", torch.float32)
        if candidates.dim() != 2 or not candidates.is_contiguous():
            raise ValueError("candidates must be contiguous 2-D")
        N, E = candidates.shape
        self.shape = (N, E)
        self.hi = torch.empty((N, E), dtype=torch.int16, device=candidates.device)
        self.lo = torch.empty((N, E), dtype=torch.int16, device=candidates.device)
        self.norm2_max = torch.zeros(1, dtype=torch.float32, device=candidates.device)
        if N:
            check(lib.mh_topk_split(_ptr(candidates), N, E, _ptr(self.hi), _ptr(self.lo), None, _ptr(self.norm2_max), _stream()),
                  "mh_topk_split")

    @staticmethod
    def supported(E: int) -> bool:
        return int(E) in (64, 128)  # the widths the filter kernel is instantiated for; others run the fp32 pipeline


def topk_mode() -> str:
    """MERLIN_HIP_TOPK = split (default: filter on the bf16 pipe where a split catalogue exists, result bit-identical) | f32."""
    import os

    return "f32" if os.environ.get("MERLIN_HIP_TOPK", "split") == "f32" else "split"


def topk_dot(q: torch.Tensor, candidates: torch.Tensor, cand_ids: Optional[torch.Tensor], k: int, split: Optional[TopKSplit] = None):
    """Brute-force retrieval: (scores[Bq,k] desc, ids[Bq,k] int32, idx[Bq,k] int32); ties -> lower index.  ``split``: the
    catalogue's ``TopKSplit`` (built once by ``BruteForce.index``): same result bit for bit, filter stages on the bf16 MFMA."""
    lib = _lib.load()
    _dev(q, "q", torch.float32)
    _dev(candidates, "candidates", torch.float32)
    q, candidates = q.contiguous(), candidates.contiguous()
    Bq, E = q.shape
    N = candidates.shape[0]
    if cand_ids is not None:
        _dev(cand_ids, "cand_ids", torch.int32)
        cand_ids = cand_ids.contiguous()
    scores = torch.empty((Bq, k), dtype=torch.float32, device=q.device)
    ids = torch.empty((Bq, k), dtype=torch.int32, device=q.device)
    idx = torch.empty((Bq, k), dtype=torch.int32, device=q.device)
    if Bq == 0:
        return scores, ids, idx
    if split is not None and topk_mode() == "split":
        if split.shape != (N, E):
            raise ValueError(f"split catalogue has shape {split.shape}, candidates {(N, E)}")
        ws = _workspace(lib.mh_topk_split_workspace_bytes(Bq, N, k, E), q.device, "topk")
        with _timed("topk_dot_split", nbytes=4 * (N + Bq) * E, flops=2 * Bq * N * E):
            check(
                lib.mh_topk_dot_split(_ptr(q), _ptr(candidates), _ptr(split.hi), _ptr(split.lo), _ptr(split.norm2_max), _ptr(cand_ids),
                                      Bq, N, E, k, _ptr(scores), _ptr(ids), _ptr(idx), _ptr(ws), ws.numel(), _stream()),
                "mh_topk_dot_split",
            )
        return scores, ids, idx
    ws = _workspace(lib.mh_topk_workspace_bytes(Bq, N, k), q.device, "topk")
    with _timed("topk_dot", nbytes=4 * (N + Bq) * E, flops=2 * Bq * N * E):
        check(
            lib.mh_topk_dot(_ptr(q), _ptr(candidates), _ptr(cand_ids), Bq, N, E, k, _ptr(scores), _ptr(ids), _ptr(idx),
                            _ptr(ws), ws.numel(), _stream()),
            "mh_topk_dot",
        )
    return scores, ids, idx


def gemm_arith() -> str:
    """The arithmetic of the GEMM-heavy layers that have a split-bf16 form -- the DCN-v2 cross layer's three GEMMs (``mh_cross_layer_fwd_split`` /
    ``_bwd_split``), wide Dense layers (``mh_linear_bias_act_fwd_split`` / ``_bwd_split``) and the tower layers (``mh_tower_split.hip``):
      unset / ``bf16x6``  the six-term split (x = h + m + l; dropped terms <= 2^-25 of a product): fp32-grade, the DEFAULT;
      ``f32``             the exact k-ascending fmaf-chain kernels everywhere (bit-reproducible against oracle/oracle_c.c);
      ``bf16x3``          the opt-in three-term split (2^-17 per operand; own dtype label) for the cross / wide layers, bf16x6 for the towers."""
    v = os.environ.get("MERLIN_HIP_GEMM_ARITH", "bf16x6")
    return v if v in ("f32", "bf16x3") else "bf16x6"


_GARITH = [None]


def _sync_gemm_arith(lib) -> None:
    """the *_split entry points compute in what ``mh_set_gemm_arith`` last recorded (1 = three terms, 2 = six): handed over when it changes"""
    want = 1 if gemm_arith() == "bf16x3" else 2
    if _GARITH[0] != want:
        check(lib.mh_set_gemm_arith(want), "mh_set_gemm_arith")
        _GARITH[0] = want


def _cross_split_ok(M: int, d: int, W: torch.Tensor) -> bool:
    return gemm_arith() != "f32" and d % 4 == 0 and d >= 64 and M >= 256 and tuple(W.shape) == (d, d)


def cross_layer(x0: torch.Tensor, x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], save_p: bool = False):
    """DCN-v2 cross layer ``x0 * (x @ W + b) + x`` (full-rank W [d, d]).  ``save_p``: also return ``p = x @ W + b``
    (what the backward multiplies the incoming gradient with), stored by the same kernel."""
    lib = _lib.load()
    for n_, t in (("x0", x0), ("x", x), ("W", W)):
        _dev(t, n_, torch.float32)
        if t.dim() != 2 or not t.is_contiguous():
            raise ValueError(f"{n_} must be contiguous 2-D")
    M, d = x.shape
    out = torch.empty_like(x)
    if _cross_split_ok(M, d, W):
        _sync_gemm_arith(lib)
        p = torch.empty_like(x) if save_p else None
        ws = _workspace(lib.mh_cross_layer_split_workspace_bytes(M, d), x.device, "cross_split")
        with _timed(f"cross_{d}", nbytes=4 * ((5 if save_p else 4) * M * d + d * d), flops=2 * M * d * d):
            check(lib.mh_cross_layer_fwd_split(_ptr(x0), _ptr(x), _ptr(W), _ptr(b), M, d, _ptr(out), _ptr(p), _ptr(ws), ws.numel(),
                                               _stream()), "mh_cross_layer_fwd_split")
        return (out, p) if save_p else out
    if save_p:
        p = torch.empty_like(x)
        with _timed(f"cross_{d}", nbytes=4 * (5 * M * d + d * d), flops=2 * M * d * d):
            check(lib.mh_cross_layer_fwd_save(_ptr(x0), _ptr(x), _ptr(W), _ptr(b), M, d, _ptr(out), _ptr(p), _stream()),
                  "mh_cross_layer_fwd_save")
        return out, p
    with _timed(f"cross_{d}", nbytes=4 * (4 * M * d + d * d), flops=2 * M * d * d):
        check(lib.mh_cross_layer_fwd(_ptr(x0), _ptr(x), _ptr(W), _ptr(b), M, d, _ptr(out), _stream()), "mh_cross_layer_fwd")
    return out


def cross_layer_lowrank(x0: torch.Tensor, x: torch.Tensor, h: torch.Tensor, V: torch.Tensor,
                        b: Optional[torch.Tensor]) -> torch.Tensor:
    """Second half of a low-rank cross layer: ``x0 * (h @ V + b) + x`` with ``h = x @ U`` [M, r], ``V`` [r, d]."""
    lib = _lib.load()
    for n_, t in (("x0", x0), ("x", x), ("h", h), ("V", V)):
        _dev(t, n_, torch.float32)
        if t.dim() != 2 or not t.is_contiguous():
            raise ValueError(f"{n_} must be contiguous 2-D")
    M, d = x.shape
    r = h.shape[1]
    if V.shape != (r, d) or x0.shape != x.shape or h.shape[0] != M:
        raise ValueError("cross_layer_lowrank: shapes must be x0, x [M, d], h [M, r], V [r, d]")
    out = torch.empty_like(x)
    with _timed(f"cross_lowrank_{d}x{r}"):
        check(lib.mh_cross_layer_lowrank_fwd(_ptr(x0), _ptr(x), _ptr(h), _ptr(V), _ptr(b), M, d, r, _ptr(out), _stream()),
              "mh_cross_layer_lowrank_fwd")
    return out


def cross_layer_backward(x0: torch.Tensor, x: torch.Tensor, p: torch.Tensor, dout: torch.Tensor, W: torch.Tensor,
                         dx0_acc: Optional[torch.Tensor] = None):
    """Backward of a full-rank cross layer ``out = x0 * (x W + b) + x`` (tf/blocks/cross.py:188-202 under the tape) through
    ``mh_cross_layer_bwd``: ONE element-wise pass (g = dout * x0, dx0_acc (+)= dout * p), dx = g W^T + dout with the residual add in
    the GEMM epilogue, dW = x^T g and db.  All operands contiguous [M, d4].  Returns ``(dx0_acc, dx, dW, db)``; ``dx0_acc`` is
    accumulated in place when given (the sum over the layers of a CrossBlock), created otherwise.  dW / db run on the "dw" side
    stream beside dX when side streams are active."""
    lib = _lib.load()
    M, d = x.shape
    for t, n in ((x0, "x0"), (x, "x"), (p, "p"), (dout, "dout")):
        _dev(t, n, torch.float32)
        if t.shape != (M, d) or not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous [{M}, {d}]")
    if d % 4 or tuple(W.shape) != (d, d) or not W.is_contiguous():
        raise ValueError("cross_layer_backward: W must be contiguous [d, d] with d % 4 == 0 (zero-padded layer)")
    accumulate = dx0_acc is not None
    if dx0_acc is None:
        dx0_acc = torch.empty_like(x)
    g = torch.empty_like(x)
    dx = torch.empty_like(x)
    dW = torch.empty((d, d), dtype=torch.float32, device=x.device)
    db = torch.empty((d,), dtype=torch.float32, device=x.device)
    if _cross_split_ok(M, d, W):  # split-bf16 GEMMs (six terms by default): the same three phases through mh_cross_layer_bwd_split
        _sync_gemm_arith(lib)
        nb = lib.mh_cross_layer_split_workspace_bytes(M, d)
        if SIDE.active("dw"):
            ws = _workspace(nb, x.device, "cross_split")
            check(lib.mh_cross_layer_bwd_split(_ptr(x0), None, _ptr(p), _ptr(dout), _ptr(W), M, d, _ptr(g), _ptr(dx0_acc),
                                               1 if accumulate else 0, _ptr(dx), None, None, _ptr(ws), ws.numel(), _stream()),
                  "mh_cross_layer_bwd_split")
            ws2 = _workspace(nb, x.device, "cross_split_side")
            with SIDE.on("dw", keep=(x, g)):
                check(lib.mh_cross_layer_bwd_split(None, _ptr(x), None, _ptr(dout), None, M, d, _ptr(g), None, 0, None, _ptr(dW), _ptr(db),
                                                   _ptr(ws2), ws2.numel(), _stream()), "mh_cross_layer_bwd_split")
            SIDE.maybe_join()
            return dx0_acc, dx, dW, db
        ws = _workspace(nb, x.device, "cross_split")
        with _timed(f"cross_bwd_{d}", nbytes=4 * M * d * (9 if accumulate else 8) + 8 * d * d, flops=4 * M * d * d):
            check(lib.mh_cross_layer_bwd_split(_ptr(x0), _ptr(x), _ptr(p), _ptr(dout), _ptr(W), M, d, _ptr(g), _ptr(dx0_acc),
                                               1 if accumulate else 0, _ptr(dx), _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                  "mh_cross_layer_bwd_split")
        return dx0_acc, dx, dW, db
    nbytes = lib.mh_linear_bwd_workspace_bytes(M, d, d)
    if SIDE.active("dw"):
        check(lib.mh_cross_layer_bwd(_ptr(x0), None, _ptr(p), _ptr(dout), _ptr(W), M, d, d, _ptr(g), _ptr(dx0_acc),
                                     1 if accumulate else 0, _ptr(dx), None, None, None, 0, _stream()), "mh_cross_layer_bwd")
        ws = _workspace(nbytes, x.device, "linear_bwd_side")
        with SIDE.on("dw", keep=(x, g)):
            check(lib.mh_cross_layer_bwd(None, _ptr(x), None, _ptr(dout), None, M, d, d, _ptr(g), None, 0, None, _ptr(dW), _ptr(db),
                                         _ptr(ws), ws.numel(), _stream()), "mh_cross_layer_bwd")
        SIDE.maybe_join()
        return dx0_acc, dx, dW, db
    ws = _workspace(nbytes, x.device, "linear_bwd")
    # element-wise pass: reads dout, x0, p (+ dx0_acc), writes g, dx0_acc; GEMMs: 2 x 2 M d^2 flops
    with _timed(f"cross_bwd_{d}", nbytes=4 * M * d * (9 if accumulate else 8) + 8 * d * d, flops=4 * M * d * d):
        check(lib.mh_cross_layer_bwd(_ptr(x0), _ptr(x), _ptr(p), _ptr(dout), _ptr(W), M, d, d, _ptr(g), _ptr(dx0_acc),
                                     1 if accumulate else 0, _ptr(dx), _ptr(dW), _ptr(db), _ptr(ws), ws.numel(), _stream()),
              "mh_cross_layer_bwd")
    return dx0_acc, dx, dW, db


def cross_lowrank_dx(dh: torch.Tensor, U: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    """dx = dh U^T + dout of a LOW-RANK cross layer (the last GEMM of its backward, residual add in the epilogue):
    dh [M, r4], U [d4, r4], dout [M, d4]."""
    lib = _lib.load()
    M, r = dh.shape
    d = U.shape[0]
    for t, n in ((dh, "dh"), (U, "U"), (dout, "dout")):
        _dev(t, n, torch.float32)
        if not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous")
    if U.shape[1] != r or dout.shape != (M, d) or r % 4 or d % 4:
        raise ValueError("cross_lowrank_dx: shapes must be dh [M, r], U [d, r], dout [M, d] with r, d multiples of 4")
    dx = torch.empty_like(dout)
    with _timed(f"cross_lowrank_dx_{d}x{r}", nbytes=4 * (M * r + d * r + 2 * M * d), flops=2 * M * d * r):
        check(lib.mh_cross_layer_bwd(None, None, None, _ptr(dout), _ptr(U), M, d, r, _ptr(dh), None, 0, _ptr(dx), None, None, None, 0,
                                     _stream()), "mh_cross_layer_bwd")
    return dx


def eltwise(op: str, a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``mul``: a*b, ``add``: a+b, ``fma``: a*b+c on contiguous fp32 tensors of one shape."""
    lib = _lib.load()
    code = {"mul": 0, "add": 1, "fma": 2}[op]
    for t in (a, b) + ((c,) if c is not None else ()):
        _dev(t, "operand", torch.float32)
        if not t.is_contiguous() or t.shape != a.shape:
            raise ValueError("eltwise operands must be contiguous and same-shaped")
    out = torch.empty_like(a)
    check(lib.mh_eltwise(code, _ptr(a), _ptr(b), _ptr(c), _ptr(out), a.numel(), _stream()), "mh_eltwise")
    return out


# --------------------------------------------------------------------------------------------
# fused DLRM segment: gather -> stack (LDS only) -> interaction (+ dense shortcut)
# --------------------------------------------------------------------------------------------
_SLOT_MEMO = [None]  # (slot_tables, slot_ids, arrays) of the last call: the backward of a step passes the forward's list objects


def _fused_slot_arrays(slot_tables, slot_ids):
    m = _SLOT_MEMO[0]
    if m is not None and m[0] is slot_tables and m[1] is slot_ids:
        return m[2]  # same lists (held alive by the memo, so the identity test cannot be fooled by a recycled id): 27 reshapes and
        # three ctypes arrays less per step on the host
    res = _fused_slot_arrays_build(slot_tables, slot_ids)
    _SLOT_MEMO[0] = (slot_tables, slot_ids, res)
    return res


def _fused_slot_arrays_build(slot_tables, slot_ids):
    F = len(slot_tables)
    idt = None
    tabs, ids, rows = [], [], []
    for t, i in zip(slot_tables, slot_ids):
        if t is None:
            tabs.append(0)
            ids.append(0)
            rows.append(0)
            continue
        _dev(t, "table", torch.float32)
        _dev(i, "ids")
        i = i.reshape(-1)
        d = _ids_dtype(i, "ids")
        if idt is not None and d != idt:
            raise TypeError("fused DLRM segment: all ids must share one dtype")
        idt = d
        tabs.append(t.data_ptr())
        ids.append(i.data_ptr())
        rows.append(t.shape[0])
    return F, idt if idt is not None else MH_I32, _host_ptr_array(tabs), (C.c_int64 * F)(*rows), _host_ptr_array(ids)


def dlrm_interaction_fused(slot_tables, slot_ids, dense: Optional[torch.Tensor], append_dense: bool = True,
                           out: Optional[torch.Tensor] = None, tail_first: bool = True) -> torch.Tensor:
    """``slot_tables[s]`` / ``slot_ids[s]`` per stack slot in sorted feature order; ``None`` marks the
    dense slot fed by ``dense[B, D]`` (the bottom-MLP output).  Returns [B, (D +) P]: ``[dense | pairs]`` with
    ``tail_first`` (the reference's ``[bottom_block | interactions]``), ``[pairs | dense]`` without."""
    lib = _lib.load()
    F, idt, tab, rows, idp = _fused_slot_arrays(slot_tables, slot_ids)
    first = next(t for t in slot_tables if t is not None)
    D = first.shape[1]
    B = next(i for i in slot_ids if i is not None).reshape(-1).shape[0]
    if dense is not None:
        _rowmajor_2d(dense, "dense")
    P = F * (F - 1) // 2
    T = D if (append_dense and dense is not None) else 0
    if out is None:
        out = torch.empty((B, P + T), dtype=torch.float32, device=first.device)
    with _timed("dlrm_fused_fwd", nbytes=B * ((F - 1) * (D * 4 + 4) + (D * 4 if dense is not None else 0) + (P + T) * 4),
                flops=2 * B * D * F * (F - 1) // 2):
        check(lib.mh_dlrm_interaction_fused_fwd(tab, rows, idp, idt, _ptr(dense), 0 if dense is None else dense.stride(0),
                                                B, F, D, int(bool(append_dense)), 1 if tail_first else 0, _ptr(out),
                                                out.stride(0), _stream()),
              "mh_dlrm_interaction_fused_fwd")
    return out


def dlrm_interaction_fused_backward(slot_tables, slot_ids, dense: Optional[torch.Tensor], dout: torch.Tensor,
                                    tail_to_dense: bool = True, tail_first: bool = True) -> torch.Tensor:
    """dX [B, F, D] of the fused segment (rows re-gathered from the not-yet-updated tables)."""
    lib = _lib.load()
    F, idt, tab, rows, idp = _fused_slot_arrays(slot_tables, slot_ids)
    first = next(t for t in slot_tables if t is not None)
    D = first.shape[1]
    _rowmajor_2d(dout, "dout")
    B = dout.shape[0]
    dx = torch.empty((B, F, D), dtype=torch.float32, device=first.device)
    with _timed("dlrm_fused_bwd", nbytes=B * ((F - 1) * (D * 4 + 4) + (D * 4 if dense is not None else 0) + dout.shape[1] * 4 + F * D * 4),
                flops=2 * B * D * F * (F - 1)):
        check(lib.mh_dlrm_interaction_fused_bwd(tab, rows, idp, idt, _ptr(dense), 0 if dense is None else dense.stride(0),
                                                _ptr(dout), dout.stride(0), B, F, D, int(bool(tail_to_dense)),
                                                1 if tail_first else 0, _ptr(dx), _stream()),
              "mh_dlrm_interaction_fused_bwd")
    return dx


def dense_optimizer_step_multi(opt, params) -> None:
    """SGD / Adagrad step on many dense Parameters in one launch (chunks of 64 tensors)."""
    lib = _lib.load()
    params = [p for p in params if p.grad is not None]
    for start in range(0, len(params), _lib.MAX_FEATURES):
        chunk = params[start:start + _lib.MAX_FEATURES]
        grads = [p.grad.contiguous() for p in chunk]
        states = states2 = None
        if opt.name == "adagrad":
            for p in chunk:
                if "accumulator" not in p.state:
                    p.state["accumulator"] = torch.full_like(p.data, opt.initial_accumulator_value)
            states = _host_ptr_array([p.state["accumulator"].data_ptr() for p in chunk])
        elif opt.name == "adam":
            mv = [_adam_states(p) for p in chunk]
            states = _host_ptr_array([m.data_ptr() for m, _ in mv])
            states2 = _host_ptr_array([v.data_ptr() for _, v in mv])
        n = len(chunk)
        check(lib.mh_dense_optimizer_step_multi(_host_ptr_array([p.data.data_ptr() for p in chunk]),
                                                _host_ptr_array([g.data_ptr() for g in grads]), states,
                                                (C.c_int64 * n)(*[p.data.numel() for p in chunk]), n, _lib.OPT[opt.name],
                                                opt.learning_rate, opt.epsilon, states2, opt.beta_1, opt.beta_2,
                                                _ptr(opt.lr_device), _stream()), "mh_dense_optimizer_step_multi")
    for p in params:
        p.grad = None


TOPK_METRIC_NAMES = ("recall", "precision", "map", "dcg", "ndcg", "mrr")


def dropout(x: torch.Tensor, rate: float, rng_state: torch.Tensor, backward: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``tf.nn.dropout(x, rate)`` with a counter-based mask (``rng_state``: device int64 ``[3]`` = seed, calls, last).
    ``backward=True`` applies the mask of the LAST forward to ``x`` (a gradient)."""
    lib = _lib.load()
    _dev(x, "x", torch.float32)
    _dev(rng_state, "rng_state", torch.int64)
    if not x.is_contiguous() or rng_state.numel() != 3:
        raise ValueError("dropout: contiguous input and a 3-entry int64 rng_state required")
    out = torch.empty_like(x) if out is None else out
    check(lib.mh_dropout(_ptr(x), _ptr(out), x.numel(), float(rate), _ptr(rng_state), int(bool(backward)), _stream()), "mh_dropout")
    return out


def batchnorm(x: torch.Tensor, gamma, beta, moving_mean: torch.Tensor, moving_var: torch.Tensor, eps: float = 1e-3,
              momentum: float = 0.99, training: bool = False):
    """Keras BatchNormalization over the last axis of ``[M, N]``: returns ``(y, save_mean, save_invstd)``; ``training``
    normalises with the batch statistics and updates the moving ones in place."""
    lib = _lib.load()
    _rowmajor_2d(x, "x")
    M, N = x.shape
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    save_mean = torch.empty(N, dtype=torch.float32, device=x.device)
    save_invstd = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = _workspace(lib.mh_batchnorm_workspace_bytes(M, N), x.device, "batchnorm")
    check(lib.mh_batchnorm_fwd(_ptr(x), x.stride(0), M, N, _ptr(gamma), _ptr(beta), float(eps), float(momentum), int(bool(training)),
                               _ptr(moving_mean), _ptr(moving_var), _ptr(save_mean), _ptr(save_invstd), _ptr(y), N, _ptr(ws),
                               ws.numel(), _stream()), "mh_batchnorm_fwd")
    return y, save_mean, save_invstd


def batchnorm_backward(x: torch.Tensor, dy: torch.Tensor, gamma, save_mean: torch.Tensor, save_invstd: torch.Tensor,
                       training: bool = True):
    """``(dx, dgamma, dbeta)`` of ``batchnorm``."""
    lib = _lib.load()
    _rowmajor_2d(x, "x")
    _rowmajor_2d(dy, "dy")
    M, N = x.shape
    dx = torch.empty((M, N), dtype=torch.float32, device=x.device)
    dgamma = torch.empty(N, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = _workspace(lib.mh_batchnorm_workspace_bytes(M, N), x.device, "batchnorm")
    check(lib.mh_batchnorm_bwd(_ptr(x), x.stride(0), _ptr(dy), dy.stride(0), M, N, _ptr(gamma), _ptr(save_mean), _ptr(save_invstd),
                               int(bool(training)), _ptr(dx), N, _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), _stream()),
          "mh_batchnorm_bwd")
    return dx, dgamma, dbeta


def copy_many(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor]) -> None:
    """``dsts[i][:] = srcs[i]`` for a list of small contiguous device tensors in ONE launch (the columns of a batch into the
    static inputs of a captured step).  Pairs the kernel cannot take (other dtypes / shapes, odd byte counts) use ``copy_``."""
    lib = _lib.load()
    ps, pd, nb = [], [], []
    for s_, d_ in zip(srcs, dsts):
        n = s_.numel() * s_.element_size()
        if (s_.is_cuda and d_.is_cuda and s_.dtype == d_.dtype and s_.shape == d_.shape and s_.is_contiguous() and d_.is_contiguous()
                and n % 4 == 0 and s_.data_ptr() % 4 == 0 and d_.data_ptr() % 4 == 0):
            ps.append(s_.data_ptr())
            pd.append(d_.data_ptr())
            nb.append(n)
        else:
            d_.copy_(s_, non_blocking=True)
    if ps:
        check(lib.mh_copy_many(_host_ptr_array(ps), _host_ptr_array(pd), (C.c_int64 * len(nb))(*nb), len(ps), _stream()), "mh_copy_many")


def concat_columns(cols: Sequence[torch.Tensor], pad_to: int = 0) -> torch.Tensor:
    """``ConcatFeatures`` of narrow fp32 device columns (``[B]`` / ``[B, w]``) in ONE launch: returns the ``[B, sum w]`` view
    of a ``[B, ld]`` buffer, ``ld`` = the width rounded up to a multiple of ``pad_to`` (0: no padding) with the padding
    columns zeroed -- 16-byte aligned rows for the vector paths of the Dense kernels."""
    lib = _lib.load()
    two_d = []
    for i, c in enumerate(cols):
        _dev(c, f"cols[{i}]", torch.float32)
        c2 = c.reshape(c.shape[0], -1) if c.dim() != 2 else c
        if c2.stride(1) != 1 and c2.shape[1] > 1:
            c2 = c2.contiguous()
        two_d.append(c2)
    B = two_d[0].shape[0]
    W = sum(c.shape[1] for c in two_d)
    ld = (W + pad_to - 1) // pad_to * pad_to if pad_to > 0 else W
    out = torch.empty((B, ld), dtype=torch.float32, device=two_d[0].device)
    if B:
        n = len(two_d)
        check(lib.mh_concat_columns(_host_ptr_array([c.data_ptr() for c in two_d]), (C.c_int64 * n)(*[int(c.stride(0)) if c.shape[0] > 1 else c.shape[1] for c in two_d]),
                                    (C.c_int32 * n)(*[int(c.shape[1]) for c in two_d]), n, B, _ptr(out), ld, ld, _stream()), "mh_concat_columns")
    return out[:, :W]


def stream_copy(src: torch.Tensor, dst: torch.Tensor) -> None:
    """``dst[:] = src`` with the library's float4 copy kernel (measurement probe: the box's achievable streaming rate)."""
    lib = _lib.load()
    _dev(src, "src")
    _dev(dst, "dst")
    nbytes = src.numel() * src.element_size()
    if not (src.is_contiguous() and dst.is_contiguous()) or dst.numel() * dst.element_size() != nbytes:
        raise ValueError("stream_copy: contiguous tensors of equal byte size required")
    check(lib.mh_stream_copy(_ptr(src), _ptr(dst), nbytes, _stream()), "mh_stream_copy")


def log_uniform_sample(range_max: int, n: int, unique: bool, rng_state: torch.Tensor, min_id: int = 0) -> torch.Tensor:
    """``n`` classes of the log-uniform (Zipfian) candidate sampler over ``[min_id, min_id + range_max)`` as int64 ``[n]``.
    ``rng_state``: device int64 ``[2]`` = (seed, calls); the kernel advances ``calls`` (no host state, no host sync)."""
    lib = _lib.load()
    _dev(rng_state, "rng_state", torch.int64)
    if rng_state.numel() != 2 or not rng_state.is_contiguous():
        raise ValueError("rng_state must be a contiguous int64 tensor with 2 entries (seed, calls)")
    out = torch.empty((int(n),), dtype=torch.int64, device=rng_state.device)
    ws = _workspace(lib.mh_log_uniform_sample_workspace_bytes(int(n), int(bool(unique))), rng_state.device, "log_uniform")
    check(lib.mh_log_uniform_sample(int(range_max), int(min_id), int(n), int(bool(unique)), _ptr(rng_state), _ptr(out), _ptr(ws),
                                    ws.numel(), _stream()), "mh_log_uniform_sample")
    return out


def log_uniform_sample_status(device) -> int:
    """The status word the LAST ``mh_log_uniform_sample`` call on ``device`` left in its workspace (0 = every requested class
    was drawn; 1 = the unique draw gave up before reaching ``n`` classes: a broken argument combination).  One host read --
    call it outside captured / timed regions (``PopularityBasedSamplerV2.check_status``: epoch ends)."""
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.type == "cuda" and dev.index is None:  # "cuda" and "cuda:<current>" are one device; the workspace is keyed by the
        dev = torch.device("cuda", torch.cuda.current_device())  # device of the rng-state TENSOR, which always carries its index
    buf = _WS.get((str(dev), "log_uniform"))
    if buf is None:
        return 0
    return int(buf[:4].view(torch.int32).item())


def topk_metrics(labels_sorted: torch.Tensor, k: int, relevant_counts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-query ranking metrics @k on pre-sorted labels -> [B, 6] (TOPK_METRIC_NAMES order)."""
    lib = _lib.load()
    _rowmajor_2d(labels_sorted, "labels_sorted")
    B = labels_sorted.shape[0]
    if k > labels_sorted.shape[1]:
        raise ValueError("k exceeds the number of sorted labels per row")
    if relevant_counts is not None:
        _dev(relevant_counts, "relevant_counts", torch.float32)
        relevant_counts = relevant_counts.reshape(-1).contiguous()
    out = torch.empty((B, 6), dtype=torch.float32, device=labels_sorted.device)
    check(lib.mh_topk_metrics(_ptr(labels_sorted), labels_sorted.stride(0), _ptr(relevant_counts), B, k, _ptr(out),
                              _stream()), "mh_topk_metrics")
    return out


# --- multi-GPU: row-sharded embedding exchange --------------------------------------------------
def route_build(ids: Sequence[torch.Tensor], world_size: int, slots: Optional[Sequence[int]] = None,
                n_slots: Optional[int] = None, capacity: int = 0, overflow: Optional[torch.Tensor] = None, dedup: bool = False):
    """Send order of the row-sharded lookup (``mh_route_build``): stable counting sort of the F*B requests
    by owner = id % W.  Returns ``(send_keys int64, pos_of [F, B] int64, src_row int64, counts [W] int64)``; see
    include/merlin_hip.h for the meaning of each.  ``capacity`` > 0: fixed windows of that many slots per owner
    (``send_keys`` / ``src_row`` have ``W * capacity`` entries, padding = -1; dropped requests have ``pos_of`` -1 and
    set the int32 device flag ``overflow``) -- equal splits, so the exchange needs no host sync."""
    lib = _lib.load()
    F = len(ids)
    if F == 0:
        raise ValueError("route_build: need at least one id column")
    idt = _ids_dtype(ids[0], "ids[0]")
    B = ids[0].numel()
    flat = []
    for f, i in enumerate(ids):
        _dev(i, f"ids[{f}]")
        if _ids_dtype(i, f"ids[{f}]") != idt:
            raise TypeError("route_build: all ids must share one dtype")
        i = i.reshape(-1)
        if i.shape[0] != B or not i.is_contiguous():
            raise ValueError(f"ids[{f}] must be contiguous with {B} entries")
        flat.append(i)
    slots = list(range(F)) if slots is None else [int(s_) for s_ in slots]
    n_slots = (max(slots) + 1) if n_slots is None else int(n_slots)
    dev = flat[0].device
    n = F * B
    n_send = world_size * int(capacity) if capacity else n
    send_keys = torch.empty(n_send, dtype=torch.int64, device=dev)
    pos_of = torch.empty((F, B), dtype=torch.int64, device=dev)
    counts = torch.empty(world_size, dtype=torch.int64, device=dev)
    if overflow is not None:
        _dev(overflow, "overflow", torch.int32)
    if dedup:
        # every distinct (feature, id) once per owner (``mh_route_build_dedup``): dense mode fills the first sum(counts) entries of
        # ``send_keys``; ``pos_of`` maps equal requests to one slot; there is no ``src_row`` (the backward is a segment sum)
        nbytes = lib.mh_route_dedup_workspace_bytes(n, world_size)
        if nbytes < 0:
            raise _lib.MerlinHipError("mh_route_dedup_workspace_bytes failed")
        ws = _workspace(nbytes, dev, "route_dedup")
        check(lib.mh_route_build_dedup(_host_ptr_array([i.data_ptr() for i in flat]), idt, F, B, world_size, int(capacity),
                                       _ptr(send_keys), _ptr(pos_of), _ptr(counts), _ptr(overflow), _ptr(ws), ws.numel(),
                                       _stream()), "mh_route_build_dedup")
        return send_keys, pos_of, None, counts
    src_row = torch.empty(n_send, dtype=torch.int64, device=dev)
    nbytes = lib.mh_route_workspace_bytes(n, world_size)
    if nbytes < 0:
        raise _lib.MerlinHipError("mh_route_workspace_bytes failed")
    ws = _workspace(nbytes, dev, "route")
    check(lib.mh_route_build(_host_ptr_array([i.data_ptr() for i in flat]), idt, F, B, world_size,
                             (C.c_int32 * F)(*slots), n_slots, int(capacity), _ptr(send_keys), _ptr(pos_of), _ptr(src_row),
                             _ptr(counts), _ptr(overflow), _ptr(ws), ws.numel(), _stream()), "mh_route_build")
    return send_keys, pos_of, src_row, counts


def route_local_rows(recv_keys: torch.Tensor, base: torch.Tensor, shard_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rows[i] = base[key >> 40] + (key & (2^40 - 1)), or -1 for padding keys and rows outside the feature's shard
    (``mh_route_local_rows``)."""
    lib = _lib.load()
    _dev(recv_keys, "recv_keys", torch.int64)
    _dev(base, "base", torch.int64)
    if shard_rows is not None:
        _dev(shard_rows, "shard_rows", torch.int64)
    rows = torch.empty_like(recv_keys)
    if recv_keys.numel():
        check(lib.mh_route_local_rows(_ptr(recv_keys), recv_keys.numel(), _ptr(base), _ptr(shard_rows), base.numel(),
                                      _ptr(rows), _stream()), "mh_route_local_rows")
    return rows
