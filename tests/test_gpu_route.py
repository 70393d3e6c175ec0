"""mh_route_build / mh_route_local_rows (row-sharded exchange) against the
framework-op statement ``distributed.route_build_torch`` -- bit-exact (index work)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.mark.parametrize("W", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("F,B", [(1, 1), (3, 257), (8, 4099), (26, 2048)])
def test_route_build_matches_stable_sort(W, dtype, F, B):
    from models_amd import ops
    from models_amd.distributed import route_build_torch

    dev = _dev()
    g = torch.Generator().manual_seed(F * 1000 + B + W)
    ids = [torch.randint(0, 1 << 20, (B,), generator=g).to(dtype).to(dev) for _ in range(F)]
    slots = torch.randperm(F + 2, generator=g)[:F].tolist()
    got = ops.route_build(ids, W, slots, F + 2)
    want = route_build_torch(ids, W, slots, F + 2)
    for a, b, name in zip(got, want, ("send_keys", "pos_of", "src_row", "counts")):
        assert torch.equal(a, b), name


def test_route_build_skewed_and_local_rows():
    from models_amd import ops
    from models_amd.distributed import route_build_torch, route_local_rows_torch

    dev = _dev()
    W = 8
    ids = [(torch.arange(5000, device=dev) * W), torch.full((5000,), 7, device=dev, dtype=torch.int64)]  # owners 0 / 7 only
    got = ops.route_build(ids, W)
    want = route_build_torch(ids, W, [0, 1], 2)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert got[3].tolist() == [5000, 0, 0, 0, 0, 0, 0, 5000]
    base = torch.tensor([0, 123456], dtype=torch.int64, device=dev)
    assert torch.equal(ops.route_local_rows(got[0], base), route_local_rows_torch(got[0], base))


def test_route_build_empty_and_errors():
    from models_amd import ops
    from models_amd._lib import MerlinHipError

    dev = _dev()
    out = ops.route_build([torch.empty(0, dtype=torch.int64, device=dev)], 4)
    assert out[3].tolist() == [0, 0, 0, 0] and out[0].numel() == 0
    with pytest.raises(MerlinHipError):
        ops.route_build([torch.zeros(4, dtype=torch.int64, device=dev)], 65)
    with pytest.raises(MerlinHipError):
        ops.route_build([torch.zeros(4, dtype=torch.int64)], 2)  # CPU tensor: no fallback


@pytest.mark.parametrize("W,cap", [(2, 4096), (8, 600), (8, 64)])
def test_route_build_fixed_capacity_matches_statement(W, cap):
    """Fixed windows (capacity > 0): padding keys / source rows are -1, requests beyond a window are dropped
    (pos_of -1) and raise the overflow flag -- bit-exact against the framework-op statement."""
    from models_amd import ops
    from models_amd.distributed import route_build_torch, route_local_rows_torch

    dev = _dev()
    g = torch.Generator().manual_seed(W * 7 + cap)
    F, B = 5, 1000
    ids = [torch.randint(0, 1 << 18, (B,), generator=g).to(torch.int32).to(dev) for _ in range(F)]
    slots = [4, 0, 2, 6, 1]
    of_hip = torch.zeros(1, dtype=torch.int32, device=dev)
    of_ref = torch.zeros(1, dtype=torch.int32, device=dev)
    got = ops.route_build(ids, W, slots, 7, capacity=cap, overflow=of_hip)
    want = route_build_torch(ids, W, slots, 7, cap, of_ref)
    for a, b, name in zip(got, want, ("send_keys", "pos_of", "src_row", "counts")):
        assert torch.equal(a, b), name
    assert int(of_hip.item()) == int(of_ref.item()) == int(F * B / W > cap)
    base = torch.tensor([0, 1 << 15, 1 << 16, 3 << 15, 1 << 17], dtype=torch.int64, device=dev)
    shard_rows = torch.tensor([20000, 1 << 15, 100, 5, 1 << 15], dtype=torch.int64, device=dev)  # some rows out of range
    r_hip = ops.route_local_rows(got[0], base, shard_rows)
    r_ref = route_local_rows_torch(got[0], base, shard_rows)
    assert torch.equal(r_hip, r_ref) and bool((r_hip[got[0] < 0] == -1).all()) and bool((r_hip == -1).any())


@pytest.mark.parametrize("rccl", [False, True])
@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_comm_sharded_lookup_single_rank_matches_direct_path(opt, rccl, monkeypatch):
    """mh_comm_* / mh_sharded_lookup_fwd / _bwd with a world of one: the result must equal the unsharded gather and the
    unsharded fused update (same kernels, same order of the duplicate rows: the route keeps request order within an owner).
    rccl=False: the exchange degenerates to device copies.  rccl=True: the communicator is a REAL one-rank RCCL communicator
    (mh_comm_unique_id -> mh_comm_init with the id: dlopen of librccl, ncclGetUniqueId, ncclCommInitRank with the 128-byte id
    BY VALUE) and every exchange runs ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, the bucket reduction
    ncclReduceScatter + ncclAllGather, and with MERLIN_HIP_ALLREDUCE=plain ncclAllReduce -- the code an N-rank job runs, on
    one GPU (reference role: sok.lookup_sparse, tf/distributed/embedding.py:117-149; hvd.DistributedOptimizer,
    tf/models/base.py:476-508)."""
    from models_amd import comm as mc
    from models_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(11)
    rows, D, B = [5000, 300, 70000], 16, 3000
    tabs = [torch.randn(r, D, generator=g).to(dev) for r in rows]
    ids = [torch.randint(0, r, (B,), generator=g).to(torch.int32).to(dev) for r in rows]
    ids[1][:50] = 7  # duplicates
    F = len(rows)
    c = mc.Comm.create(force_rccl=rccl)
    assert (c.rank, c.world) == (0, 1)
    local = torch.cat(tabs).contiguous()
    base = torch.tensor([0, rows[0], rows[0] + rows[1]], dtype=torch.int64, device=dev)
    srows = torch.tensor(rows, dtype=torch.int64, device=dev)
    look = mc.ShardedLookup(c, local.clone(), base, srows, capacity=F * B)
    out = torch.zeros(B, F * D + 8, device=dev)
    look.forward(ids, out, [0, D, 2 * D])
    want = torch.cat([t[i.long()] for t, i in zip(tabs, ids)], dim=1)
    assert torch.equal(out[:, : F * D], want) and int(look.overflow.item()) == 0 and bool((out[:, F * D:] == 0).all())

    grad = torch.randn(B, F, D, generator=g).to(dev)
    state = torch.full_like(local, 0.1) if opt == "adagrad" else None
    look.backward(grad, optimizer=opt, lr=0.05, state=state)
    ref_t = [t.clone() for t in tabs]
    ref_s = [torch.full_like(t, 0.1) for t in tabs] if opt == "adagrad" else None
    gflat = grad.reshape(B, F * D)
    ops.embedding_gather_backward(ref_t, ref_s, ids, gflat, [0, D, 2 * D], optimizer=opt, lr=0.05)
    got = torch.split(look.local, rows)
    for a, b in zip(got, ref_t):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)  # summation order of duplicates may differ between the two sorts
    if opt == "adagrad":
        for a, b in zip(torch.split(state, rows), ref_s):
            torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)

    # a too-small window drops requests and raises the flag instead of faulting
    small = mc.ShardedLookup(c, local.clone(), base, srows, capacity=B)
    small.forward(ids, torch.zeros(B, F * D, device=dev), [0, D, 2 * D])
    assert int(small.overflow.item()) == 1

    flat = torch.randn(1000, device=dev)
    keep = flat.clone()
    assert torch.equal(c.allreduce_(flat), keep)  # world of one: identity (reduce-scatter + all-gather under rccl=True)
    monkeypatch.setenv("MERLIN_HIP_ALLREDUCE", "plain")
    assert torch.equal(c.allreduce_(flat), keep)  # one ncclAllReduce under rccl=True
    monkeypatch.delenv("MERLIN_HIP_ALLREDUCE")
    a, b = torch.arange(64, device=dev, dtype=torch.float32), torch.empty(64, device=dev)
    c.alltoall(a, b)
    assert torch.equal(a, b)
    # the asynchronous forms the sharded step uses (communicator's own stream, event hand-off)
    b.zero_()
    w1 = c.alltoall_async(a, b)
    w2 = c.allreduce_async(flat)
    w1.wait()
    w2.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(flat, keep)
    c.destroy()


@pytest.mark.parametrize("W", [1, 2, 8, 64])
@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("F,B,card", [(1, 1, 5), (3, 257, 40), (8, 4099, 1 << 20), (26, 2048, 300), (26, 65536, 20000)])
def test_route_build_dedup_matches_statement(W, dtype, F, B, card):
    """``mh_route_build_dedup`` (hash leader election + counting sort of the leaders) against the framework-op statement, bit for
    bit: the send slot of a key is the rank of its FIRST occurrence among its owner's distinct keys, whatever order the atomics
    ran in.  card = 40 / 300: nearly every request is a duplicate; 2^20: nearly none."""
    from models_amd import ops
    from models_amd.distributed import route_build_torch

    dev = _dev()
    g = torch.Generator().manual_seed(F * 1000 + B + W)
    ids = [torch.randint(0, card, (B,), generator=g).to(dtype).to(dev) for _ in range(F)]
    if B > 4:
        ids[0][3] = -5  # a negative id: no owner has it (pos_of -1), and it must not poison the table
    keys, pos, src, counts = ops.route_build(ids, W, dedup=True)
    wk, wp, ws, wc = route_build_torch(ids, W, list(range(F)), F, dedup=True)
    assert src is None and ws is None
    assert torch.equal(counts, wc)
    total = int(wc.sum())
    assert keys.numel() == F * B and torch.equal(keys[:total], wk)
    assert torch.equal(pos, wp)
    for run in range(2):  # same answer again (atomics race differently; the workspace is reused)
        k2, p2, _, c2 = ops.route_build(ids, W, dedup=True)
        assert torch.equal(k2[:total], wk) and torch.equal(p2, wp) and torch.equal(c2, wc)


@pytest.mark.parametrize("W,cap,card", [(2, 4096, 3000), (8, 64, 5000), (8, 128, 600)])
def test_route_build_dedup_fixed_windows_match_statement(W, cap, card):
    from models_amd import ops
    from models_amd.distributed import route_build_torch

    dev = _dev()
    g = torch.Generator().manual_seed(W * 7 + cap)
    F, B = 6, 3001
    ids = [torch.randint(0, card, (B,), generator=g).to(dev) for _ in range(F)]
    over, wover = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    keys, pos, _, counts = ops.route_build(ids, W, capacity=cap, overflow=over, dedup=True)
    wk, wp, _, wc = route_build_torch(ids, W, list(range(F)), F, capacity=cap, overflow=wover, dedup=True)
    assert torch.equal(keys, wk) and torch.equal(pos, wp) and torch.equal(counts, wc) and int(over) == int(wover)
    assert int(over) == int(bool((wc > cap).any()))


def test_dedup_segment_sum_through_the_fused_update():
    """The sender side of a de-duplicated backward: gradient rows of equal requests summed into their send slot by
    ``mh_embedding_gather_bwd`` (SGD, lr = -1, onto zeros) == the framework-op statement."""
    from models_amd import ops
    from models_amd.distributed import _hip_segment_sum, route_build_torch, segment_sum_torch

    dev = _dev()
    g = torch.Generator().manual_seed(4)
    F, B, D, W = 5, 3000, 16, 4
    ids = [torch.randint(0, 500, (B,), generator=g).to(dev) for _ in range(F)]
    keys, pos, _, counts = ops.route_build(ids, W, capacity=640, overflow=torch.zeros(1, dtype=torch.int32, device=dev), dedup=True)
    dstack = torch.randn(B, F + 2, D, generator=g).to(dev)
    slots = [6, 0, 3, 1, 4]
    got = _hip_segment_sum(dstack, slots, pos, keys.numel())
    want = segment_sum_torch(dstack.cpu().double(), slots, pos.cpu(), keys.numel())
    torch.testing.assert_close(got.cpu().double(), want, atol=1e-5, rtol=1e-5)
    assert bool((got[keys < 0] == 0).all())  # padding slots carry no gradient


@pytest.mark.parametrize("rccl", [False, True])
def test_recorded_collectives_are_replayed(rccl):
    """A collective of the C-ABI communicator issued while the library records a step's launch sequence is part of the recording
    (``mh_record_*``): the replay moves the data again -- through RCCL on the one-rank communicator, a device copy without one."""
    import ctypes as C

    from models_amd import _lib, comm as mc

    dev = _dev()
    lib = _lib.load()
    c = mc.Comm.create(force_rccl=rccl)
    send = torch.arange(4096, dtype=torch.float32, device=dev)
    recv = torch.zeros_like(send)
    flat = torch.full((1024,), 3.0, device=dev)
    _lib.check(lib.mh_record_begin(), "mh_record_begin")
    try:
        c.alltoall(send, recv)
        c.allreduce_(flat)
    except Exception:
        lib.mh_record_abort()
        raise
    h = C.c_void_p()
    _lib.check(lib.mh_record_end(C.byref(h)), "mh_record_end")
    n, e = C.c_int64(), C.c_int64()
    _lib.check(lib.mh_record_info(h, C.byref(n), C.byref(e)), "mh_record_info")
    assert n.value == (2 if rccl else 1)  # without RCCL a one-rank all-reduce is nothing at all
    torch.cuda.synchronize()
    assert torch.equal(recv, send)
    send.mul_(2.0)
    recv.zero_()
    _lib.check(lib.mh_record_replay(h), "mh_record_replay")
    torch.cuda.synchronize()
    assert torch.equal(recv, send) and bool((flat == 3.0).all())  # a one-rank sum leaves the bucket as it is
    lib.mh_record_free(h)


@pytest.mark.parametrize("rccl", [False, True])
def test_async_collectives_on_the_communicator_stream(rccl):
    """``Comm.alltoall_async`` / ``allreduce_async`` (what the N-rank exchange calls): issued on the communicator's own stream
    behind the caller's work so far, ``wait()`` orders the caller's stream behind them."""
    from models_amd import comm as mc

    dev = _dev()
    c = mc.Comm.create(force_rccl=rccl)
    send = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
    recv = torch.empty_like(send)
    send.add_(5.0)                      # enqueued on the caller's stream BEFORE the collective: it must see the 5s
    w = c.alltoall_async(send, recv)
    w.wait()
    got = recv.clone()                  # on the caller's stream, behind the wait
    flat = torch.full((4096,), 2.0, device=dev)
    w2 = c.allreduce_async(flat)
    w2.wait()
    torch.cuda.synchronize()
    assert bool((got == 5.0).all()) and bool((flat == 2.0).all())
