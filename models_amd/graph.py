"""hipGraph capture of a whole hot-path step: as ONE graph (``GraphedStep``) or as per-stream graph SEGMENTS replayed on
several streams (``SegmentedStep``: the zero-host-work replay that keeps the eager step's side-stream overlap).

The DLRM forward at batch 64 K is ~15 kernel launches of 10-150 us each: launched eagerly from
Python the step is host-bound by an order of magnitude.  The step is a fixed launch sequence
with static shapes, so it is captured ONCE into a hipGraph (through torch's CUDAGraph wrapper:
plumbing) and replayed; new batches are copied into the static input buffers.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch


def _stage(static: Dict[str, torch.Tensor], new: Dict[str, torch.Tensor]) -> None:
    """The next batch into the static inputs of a captured step: ONE launch for all columns (``ops.copy_many``)."""
    from . import ops

    keys = list(new)
    if keys and all(new[k].is_cuda for k in keys):
        ops.copy_many([new[k] for k in keys], [static[k] for k in keys])
    else:
        for k in keys:
            static[k].copy_(new[k], non_blocking=True)


class PackedBatch:
    """A batch (dict of tensors) stored as views of ONE contiguous buffer per dtype, so that loading the next batch
    into the static inputs of a captured step is one device copy per dtype (two for ids + floats) instead of one per
    column -- 40 columns would otherwise cost 40 copy launches per 1.5 ms step."""

    def __init__(self, tensors: Dict[str, torch.Tensor]):
        self.layout = []  # (name, dtype, offset, shape)
        sizes: Dict[torch.dtype, int] = {}
        for k, v in tensors.items():
            n = (v.numel() + 63) // 64 * 64  # 256-byte aligned columns
            self.layout.append((k, v.dtype, sizes.get(v.dtype, 0), tuple(v.shape)))
            sizes[v.dtype] = sizes.get(v.dtype, 0) + n
        dev = next(iter(tensors.values())).device
        self.buffers = {dt: torch.empty(n, dtype=dt, device=dev) for dt, n in sizes.items()}
        self.tensors: Dict[str, torch.Tensor] = {}
        for k, dt, off, shape in self.layout:
            n = 1
            for d in shape:
                n *= d
            view = self.buffers[dt][off:off + n].view(shape)
            view.copy_(tensors[k])
            self.tensors[k] = view

    def copy_from(self, other: "PackedBatch") -> None:
        for dt, buf in self.buffers.items():
            buf.copy_(other.buffers[dt], non_blocking=True)

    def nbytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self.buffers.values())


class GraphedStep:
    """Capture ``fn(static_inputs)`` once; ``replay(new_inputs)`` copies + replays.  ``inputs`` is a dict of tensors
    or a ``PackedBatch`` (then ``new_inputs`` must be a ``PackedBatch`` of the same layout)."""

    def __init__(self, fn: Callable, inputs, warmup: int = 3):
        self.packed = None
        if isinstance(inputs, PackedBatch):
            self.packed = PackedBatch(inputs.tensors)
            self.inputs = self.packed.tensors
        else:
            self.inputs = {k: v.clone() for k, v in inputs.items()}
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # builds lazy layers, sets LDS attributes, warms the allocator
                fn(self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        from . import ops

        self.graph = torch.cuda.CUDAGraph()
        ops.CAPTURING[0] += 1  # persistent buffers touched now are addressed by this graph for good (ops.note_captured / park_replaced)
        try:
            with torch.cuda.graph(self.graph):
                self.output = fn(self.inputs)
        finally:
            ops.CAPTURING[0] -= 1

    def replay(self, new_inputs=None):
        if new_inputs is not None:
            if self.packed is not None:
                self.packed.copy_from(new_inputs)
            else:
                _stage(self.inputs, new_inputs)
        self.graph.replay()
        return self.output

    __call__ = replay


class _KeepAllocations:
    """While active, every CUDA tensor an aten op returns is kept alive (a TorchDispatchMode): the storage a recorded step
    allocated stays allocated -- and in place -- for as long as the recording that addresses it (the role of a graph's private
    memory pool, without depending on which allocator pool a block came from or on ``empty_cache`` leaving it alone)."""

    def __init__(self):
        from torch.utils._python_dispatch import TorchDispatchMode

        keep = self.keep = []
        impure = self.impure = []  # aten ops that launch kernels of their own: a recorded step would not replay them
        pure = ("empty", "view", "as_strided", "slice", "select", "reshape", "unsqueeze", "squeeze", "expand", "permute",
                "transpose", "detach", "alias", "t.default", "split", "unbind", "narrow", "_unsafe_view", "lift_fresh",
                "is_pinned", "stride", "size", "numel", "storage_offset", "sym_")
        # NOT in the list on purpose: _to_copy of a device tensor (a conversion / copy kernel) and _local_scalar_dense (``.item()``:
        # a host read the replay would not repeat -- a step that branches on one cannot be recorded)

        class _Mode(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                out = func(*args, **(kwargs or {}))
                cuda = False
                for t in (out if isinstance(out, (tuple, list)) else (out,)):
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        keep.append(t)
                        cuda = True
                if not cuda:
                    cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
                name = str(func)
                if cuda and not any(p_ in name for p_ in pure):
                    impure.append(name)
                return out

        self._mode = _Mode()

    def __enter__(self):
        self._mode.__enter__()
        return self

    def __exit__(self, *exc):
        return self._mode.__exit__(*exc)


class RecordedStep:
    """``fn(static_inputs)`` executed ONCE while ``libmerlin_hip.so`` records its launch sequence (``mh_record_begin / _end``: every
    kernel launch with its arguments by value, on the stream it was issued on, plus the event hand-offs between the streams, which
    ``ops.SIDE`` issues through the library while ``crec`` is set) and replayed by ONE C call per step (``mh_record_replay``).

    What the three other launch modes cost: eager launches need ~0.75 ms of Python per step (~35 launches and a dozen stream /
    event operations); ONE hipGraph replays on one hardware queue (no overlap); per-stream graph segments (``SegmentedStep``) keep
    the overlap but pay a graph launch + hand-off per segment.  The recorded sequence is the eager step's own sequence -- same
    streams, same overlap -- issued from C.  Measured (round 4, ``profiles/r4_notes.md``): 0.957-0.960 ms against 0.945-0.950 for
    the eager step on the same box -- on that host the step is bound by its GPU critical path, so eager stays the default and this
    class is the host-free mode for slower hosts.  Same contract as a captured graph: static inputs (every batch is
    staged into them), fixed shapes, no host-side decision inside the step; the step must consist of library launches only (a torch
    kernel inside it would not be replayed: ``assert_pure`` checks the recording against a kernel-free torch dispatch).
    Same interface as ``GraphedStep`` / ``SegmentedStep``."""

    def __init__(self, fn: Callable, inputs, warmup: int = 3, assert_pure: bool = True):
        import ctypes as C

        from . import _lib, ops

        self.packed = None
        if isinstance(inputs, PackedBatch):
            self.packed = PackedBatch(inputs.tensors)
            self.inputs = self.packed.tensors
        else:
            self.inputs = {k: v.clone() for k, v in inputs.items()}
        self.fn = fn
        for _ in range(warmup):  # eager: builds lazy layers, sizes every workspace, sets the kernels' LDS attributes
            fn(self.inputs)
        torch.cuda.synchronize()
        if ops.SIDE.recorder is not None or ops.SIDE.crec:
            raise RuntimeError("a step is already being recorded")
        lib = _lib.load()
        self._lib = lib
        self._keep = _KeepAllocations()
        _lib.check(lib.mh_record_begin(), "mh_record_begin")
        ops.SIDE.crec = True
        ops.CAPTURING[0] += 1  # persistent buffers touched now are addressed by the recording for good (note_captured / park_replaced)
        ok = False
        try:
            with self._keep:
                self.output = fn(self.inputs)  # executes for real AND is recorded
            ok = True
        finally:
            ops.SIDE.crec = False
            ops.CAPTURING[0] -= 1
            if not ok:
                lib.mh_record_abort()
        h = C.c_void_p()
        _lib.check(lib.mh_record_end(C.byref(h)), "mh_record_end")
        self.handle = h
        self.impure_ops = sorted(set(self._keep.impure))
        if assert_pure and self.impure_ops:
            lib.mh_record_free(h)
            self.handle = None
            raise RuntimeError("the step launches kernels outside libmerlin_hip.so, which a recorded launch sequence would not "
                               f"replay: {self.impure_ops}")
        n, e = C.c_int64(), C.c_int64()
        _lib.check(lib.mh_record_info(h, C.byref(n), C.byref(e)), "mh_record_info")
        self.n_launches, self.n_hand_offs = int(n.value), int(e.value)
        torch.cuda.synchronize()
        # side streams the recording issues on: they must outlive it (ops.SIDE keeps them), and so must the launch stream
        self._stream = torch.cuda.current_stream()

    def replay(self, new_inputs=None):
        if new_inputs is not None:
            if self.packed is not None:
                self.packed.copy_from(new_inputs)
            else:
                _stage(self.inputs, new_inputs)
        if torch.cuda.current_stream() != self._stream:
            raise RuntimeError("a recorded step replays on the stream it was recorded on")
        from . import _lib

        _lib.check(self._lib.mh_record_replay(self.handle), "mh_record_replay")
        return self.output

    __call__ = replay

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.mh_record_free(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class SegmentedStep:
    """``fn(static_inputs)`` recorded ONCE as a list of per-stream hipGraph segments (``ops.StepRecorder``: every
    ``SIDE.on / mark / wait / join`` of the step cuts a segment and notes a dependency edge) and replayed by launching the
    segments on real streams with events on the edges.

    One graph of the whole step replays on ONE hardware queue (ROCm 7.2): no host work, no overlap.  Eager launches overlap
    (sort beside the top MLP, dW beside dX, sparse apply beside the bottom-MLP backward) but need ~0.75 ms of Python per step.
    A segmented step has both: ~10 graph launches + a few event operations of host work per step, the eager step's overlap.
    Same interface as ``GraphedStep``."""

    def __init__(self, fn: Callable, inputs, warmup: int = 3):
        from . import ops

        self.packed = None
        if isinstance(inputs, PackedBatch):
            self.packed = PackedBatch(inputs.tensors)
            self.inputs = self.packed.tensors
        else:
            self.inputs = {k: v.clone() for k, v in inputs.items()}
        self.fn = fn
        for _ in range(warmup):  # eager, with the real side streams: builds lazy layers, sizes every workspace
            fn(self.inputs)
        torch.cuda.synchronize()
        if ops.SIDE.recorder is not None:
            raise RuntimeError("a step is already being recorded")
        rec = ops.StepRecorder()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        ops.SIDE.recorder = rec
        ops.CAPTURING[0] += 1  # persistent buffers touched now are addressed by the segments for good (ops.note_captured / park_replaced)
        try:
            with torch.cuda.stream(cap):
                rec.begin()
                try:
                    self.output = fn(self.inputs)
                finally:
                    rec.finish()
        finally:
            ops.SIDE.recorder = None
            ops.CAPTURING[0] -= 1
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        self.segments = rec.segments
        needed = {d for sg in self.segments for d in sg["deps"]}
        self._events = {i: torch.cuda.Event() for i in needed}
        self._side = {sg["stream"]: ops.SIDE.stream(sg["stream"]) for sg in self.segments if sg["stream"] != "main"}
        self._skip = {i for i, sg in enumerate(self.segments) if sg.get("empty")}  # nothing captured: only their edges matter
        self._first = True

    @property
    def n_segments(self) -> int:
        return len(self.segments) - len(self._skip)

    def replay(self, new_inputs=None):
        if new_inputs is not None:
            if self.packed is not None:
                self.packed.copy_from(new_inputs)
            else:
                _stage(self.inputs, new_inputs)
        cur = torch.cuda.current_stream()
        evs = self._events
        for i, sg in enumerate(self.segments):
            st = cur if sg["stream"] == "main" else self._side[sg["stream"]]
            for d in sg["deps"]:
                st.wait_event(evs[d])
            if i not in self._skip:
                if st is cur:
                    self._launch(i, sg)
                else:
                    with torch.cuda.stream(st):
                        self._launch(i, sg)
            if i in evs:
                evs[i].record(st)
        self._first = False
        # every side segment was joined by a later main segment of the step (SIDE.deferred() exits join): the launch stream
        # is ordered after the whole step
        return self.output

    def _launch(self, i, sg) -> None:
        # segments that captured nothing (two cuts in a row) were flagged at capture time and are in _skip; a launch failure of
        # any OTHER segment is a real error -- dropping it would silently remove a part of the train step
        try:
            sg["graph"].replay()
        except RuntimeError as e:
            raise RuntimeError(f"segmented step: segment {i} (stream {sg['stream']!r}) failed to launch: {e}") from e

    __call__ = replay
