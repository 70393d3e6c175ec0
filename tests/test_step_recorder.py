"""Host logic of the segmented step replay (ops.StepRecorder behind ops.SIDE): how a step's side-stream blocks are cut into per-stream
graph segments with dependency edges.  No GPU: the graph objects are stand-ins that only count captures."""
import pytest
import torch

from models_amd import ops


class _FakeGraph:
    live = 0

    def __init__(self):
        self.nodes = 0

    def capture_begin(self):
        assert _FakeGraph.live == 0, "one capture at a time"
        _FakeGraph.live += 1

    def capture_end(self):
        _FakeGraph.live -= 1


@pytest.fixture
def recorder(monkeypatch):
    monkeypatch.setattr(torch.cuda, "CUDAGraph", _FakeGraph)
    rec = ops.StepRecorder()
    assert ops.SIDE.recorder is None
    ops.SIDE.recorder = rec
    rec.begin()
    yield rec
    ops.SIDE.recorder = None


def _finish(rec):
    rec.finish()
    segs = rec.segments
    assert _FakeGraph.live == 0
    assert all(d < i for i, sg in enumerate(segs) for d in sg["deps"]), "edges point backwards in launch order"
    return segs


def test_a_step_without_side_work_is_one_main_segment(recorder):
    segs = _finish(recorder)
    assert [(s["stream"], s["deps"]) for s in segs] == [("main", [])]


def test_side_blocks_cut_segments_and_the_join_orders_main_behind_them(recorder):
    with ops.SIDE.on("sort"):          # forked behind what main has enqueued so far
        pass
    m = ops.SIDE.mark()                # an ordering point on main
    with ops.SIDE.on("sparse", after=[m]):
        pass
    ops.SIDE.join()
    segs = _finish(recorder)
    streams = [s["stream"] for s in segs]
    # the default configuration runs every kind of side work on ONE logical side stream
    assert set(streams) == {"main", "sort"}
    first_side = streams.index("sort")
    assert segs[first_side]["deps"] == [first_side - 1]          # behind the main segment that preceded the fork
    second_side = len(streams) - 1 - streams[::-1].index("sort")
    assert second_side > first_side
    assert all(segs[d]["stream"] == "main" for d in segs[second_side]["deps"])  # its `after` marker; the earlier side segment
    # is ordered before it by the stream itself (same logical stream: no edge needed)
    last = segs[-1]
    assert last["stream"] == "main" and second_side in last["deps"]            # the join


def test_wait_on_a_marker_adds_an_edge_only_across_streams(recorder):
    m = ops.SIDE.mark()
    ops.SIDE.wait(m)                   # main waiting for main: a cut, but no edge
    with ops.SIDE.on("dw"):
        inner = ops.SIDE.mark()        # an ordering point INSIDE the side block
    ops.SIDE.wait(inner)               # main waits for that point of the side stream
    segs = _finish(recorder)
    main_after_wait = segs[-1]
    assert main_after_wait["stream"] == "main"
    assert any(segs[d]["stream"] == "sort" for d in main_after_wait["deps"])
    assert all(segs[d]["stream"] != sg["stream"] for sg in segs for d in sg["deps"])


def test_join_stream_joins_one_kind_and_leaves_nothing_pending(recorder):
    with ops.SIDE.on("dw"):
        pass
    ops.SIDE.join_stream("dw")
    with ops.SIDE.deferred():
        with ops.SIDE.on("sparse"):
            pass
        ops.SIDE.maybe_join()          # deferred: no join yet
        n_before = len(recorder.segments)
    # leaving the deferred context joins
    assert len(recorder.segments) > n_before
    segs = _finish(recorder)
    assert not recorder._pending


def test_alias_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("MERLIN_HIP_SIDE_ALIAS", "none")
    s = ops._SideStreams()
    assert s._alias == {}
    monkeypatch.setenv("MERLIN_HIP_SIDE_ALIAS", "dw=sort")
    assert ops._SideStreams()._alias == {"dw": "sort"}
    monkeypatch.delenv("MERLIN_HIP_SIDE_ALIAS")
    assert ops._SideStreams()._alias == {"dw": "sort", "sparse": "sort"}


def test_replaced_buffers_are_parked_only_if_a_capture_touched_them():
    a, b = torch.zeros(4), torch.zeros(4)
    n = len(ops._PARKED)
    ops.note_captured(a)               # outside a capture: not marked
    ops.park_replaced(a)
    assert len(ops._PARKED) == n
    ops.CAPTURING[0] += 1
    try:
        ops.note_captured(b)
    finally:
        ops.CAPTURING[0] -= 1
    ops.park_replaced(b)
    assert len(ops._PARKED) == n + 1 and ops._PARKED[-1] is b
    ops._PARKED.pop()


def test_concat_features_takes_the_hip_kernel_only_for_device_columns():
    """ConcatFeatures (tf/core/aggregation.py:38-66): narrow fp32 DEVICE columns go through mh_concat_columns; host tensors are plain
    buffer plumbing (torch.cat) -- and the HIP op itself refuses host tensors loudly instead of falling back."""
    from models_amd import _lib
    from models_amd.core import ConcatFeatures

    cols = {"b": torch.arange(6, dtype=torch.float32).reshape(3, 2), "a": torch.ones(3)}
    out = ConcatFeatures()(cols)
    assert out.shape == (3, 3) and torch.equal(out[:, 0], torch.ones(3)) and torch.equal(out[:, 1:], cols["b"])  # sorted-key order
    with pytest.raises(_lib.MerlinHipError):
        ops.concat_columns([torch.ones(3)])


def test_join_of_one_kind_does_not_wait_for_other_kinds_on_the_shared_stream(recorder):
    """optim.py joins the dW GEMMs before the dense update while the sparse apply is already queued on the same side stream: the
    edge must point at the dW segment, not at the apply's."""
    with ops.SIDE.on("dw"):
        pass
    with ops.SIDE.on("sparse"):
        pass
    ops.SIDE.join_stream("dw")
    segs_now = len(recorder.segments)
    dense = recorder.segments[-1]
    side = [i for i, s in enumerate(recorder.segments) if s["stream"] == "sort"]
    assert len(side) == 2 and dense["stream"] == "main" and dense["deps"] == [side[0]]
    ops.SIDE.join()
    segs = _finish(recorder)
    assert side[1] in segs[-1]["deps"] and len(segs) > segs_now
