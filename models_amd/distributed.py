"""Multi-GPU execution of the hot path: one process per GPU, RCCL over xGMI via torch.distributed.

What the reference does: Horovod data parallelism (`tf/models/base.py:476-508`: gradient
all-reduce, averaged; `:1472-1473` rank-0 broadcast) and, optionally, SOK model-parallel embeddings
(`tf/distributed/embedding.py:47-149`: rows spread over all GPUs, ids exchanged, vectors returned).

MI355X-first redesign:
  * the batch is sharded by rank (pure DP through MLPs / interaction / scorer);
  * large tables are ROW-SHARDED: ``owner = row % W``, ``local_row = row // W`` (balanced under skew);
    lookup = all-to-all(ids) -> local HIP gather -> all-to-all(rows); backward = all-to-all(row grads)
    -> the owner's fused dedup + optimizer update.  Embedding gradients are never all-reduced.
    xGMI is point-to-point (7 links per GPU), so an all-to-all uses every link at once;
  * small tables are replicated; their gradient is accumulated into a dense [V, D] buffer by the same
    fused backward kernel (SGD with lr = -1 on a zeroed buffer), summed across ranks with the dense
    bucket, then applied as a dense update -- a few MB instead of an all-gather of B x F rows;
  * dense gradients travel in ONE flat bucket: reduce-scatter + all-gather on RCCL (all links busy),
    plain all-reduce on gloo (CPU tests).

Every collective lives in this file; the compute callables are injected so that the routing logic is
covered by world-size-2 ``gloo`` tests on CPU (tests/test_distributed.py) with oracle kernels, while the
product path passes the HIP ops.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def world() -> Tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if is_initialized() else (0, 1)


# ------------------------------------------------------------------------------------------------
# row sharding
# ------------------------------------------------------------------------------------------------
def local_rows(global_rows: int, rank: int, world_size: int) -> int:
    """Number of rows owned by ``rank`` under ``owner = row % W``."""
    return (global_rows - rank + world_size - 1) // world_size if global_rows > rank else 0


def shard_table(full: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """Rows ``rank, rank + W, rank + 2W, ...`` of a full table (``local_row = row // W``)."""
    return full[rank::world_size].contiguous()


class sharded_tables:
    """``with distributed.sharded_tables(threshold): model = mm.DLRMModel(...)``: every EmbeddingTable with at least
    ``threshold`` rows is allocated as this rank's row shard only (rows ``rank, rank + W, ...``; same values as the
    unsharded table would hold).  Wrap the model in ``DistributedDLRM`` / ``DistributedModel`` afterwards."""

    def __init__(self, threshold: int = 200_000, force: bool = False):
        self.threshold, self.force = int(threshold), force

    def __enter__(self):
        from . import inputs

        rank, W = world()
        self._prev = inputs._SHARD_CTX
        inputs._SHARD_CTX = (rank, W, self.threshold) if (W > 1 or self.force) else None
        return self

    def __exit__(self, *exc):
        from . import inputs

        inputs._SHARD_CTX = self._prev
        return False


class Route:
    """Send ``payload[i]`` (int64) to rank ``owner[i]``: permutation into owner order + per-peer counts.
    ONE host sync per route (RCCL needs host-side split sizes): send and receive counts travel together."""

    def __init__(self, owner: torch.Tensor, payload: torch.Tensor, world_size: int, group=None):
        owner, payload = owner.reshape(-1), payload.reshape(-1)
        self.n = owner.shape[0]
        self.world_size = world_size
        self.group = group
        self.order = torch.argsort(owner, stable=True)
        send_counts = torch.bincount(owner, minlength=world_size)
        recv_counts = torch.empty_like(send_counts)
        if world_size > 1:
            dist.all_to_all_single(recv_counts, send_counts, group=group)
        else:
            recv_counts.copy_(send_counts)
        both = torch.stack([send_counts, recv_counts]).tolist()  # the one host sync
        self.send_counts: List[int] = both[0]
        self.recv_counts: List[int] = both[1]
        send = payload[self.order].contiguous()
        self.recv_payload = torch.empty(sum(self.recv_counts), dtype=payload.dtype, device=payload.device)
        self._a2a(self.recv_payload, send, self.recv_counts, self.send_counts)

    @classmethod
    def for_rows(cls, ids: torch.Tensor, world_size: int, group=None) -> "Route":
        """owner = row % W, payload = local row = row // W."""
        ids = ids.reshape(-1)
        r = cls(torch.remainder(ids, world_size), torch.div(ids, world_size, rounding_mode="floor"), world_size, group)
        r.recv_rows = r.recv_payload
        return r

    def _a2a(self, out, inp, out_splits, in_splits):
        if self.world_size > 1:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        else:
            out.copy_(inp)

    def return_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """Owner -> requester: ``rows`` is [sum(recv_counts), D]; result is [n, D] in the ORIGINAL order."""
        D = rows.shape[1]
        back = torch.empty((self.n, D), dtype=rows.dtype, device=rows.device)
        self._a2a(back, rows.contiguous(), self.send_counts, self.recv_counts)
        out = torch.empty_like(back)
        out[self.order] = back
        return out

    def send_grads(self, grad: torch.Tensor) -> torch.Tensor:
        """Requester -> owner: ``grad`` [n, D] in original order; result [sum(recv_counts), D] aligned
        with ``recv_payload``."""
        D = grad.shape[1]
        g = grad[self.order].contiguous()
        out = torch.empty((sum(self.recv_counts), D), dtype=grad.dtype, device=grad.device)
        self._a2a(out, g, self.recv_counts, self.send_counts)
        return out


class ShardedEmbeddingTable:
    """One row-sharded table.  ``gather_fn(local_table, local_rows) -> [n, D]`` and
    ``update_fn(local_table, state, local_rows, grads)`` are the HIP ops in production."""

    def __init__(self, local_table: torch.Tensor, global_rows: int, gather_fn: Callable, update_fn: Callable,
                 group=None):
        self.rank, self.world_size = world()
        self.table = local_table
        self.global_rows = global_rows
        self.gather_fn, self.update_fn = gather_fn, update_fn
        self.group = group
        self.state: Optional[torch.Tensor] = None
        self._route: Optional[Route] = None

    def lookup(self, ids: torch.Tensor) -> torch.Tensor:
        self._route = Route.for_rows(ids, self.world_size, self.group)
        rows = self.gather_fn(self.table, self._route.recv_rows)
        return self._route.return_rows(rows)

    def backward_update(self, grad: torch.Tensor) -> None:
        g = self._route.send_grads(grad)
        self.update_fn(self.table, self.state, self._route.recv_rows, g)


def _route_build_dedup_torch(idm: torch.Tensor, owner: torch.Tensor, W: int, capacity: int, overflow):
    """Framework-op statement of ``mh_route_build_dedup``: every distinct (feature, id) once per owner, in the order of first
    occurrence (entry order e = f*B + b) within the owner; ``pos_of`` maps equal requests to one slot; negative ids -> -1."""
    F, B = idm.shape
    n = F * B
    dev = idm.device
    feat = torch.arange(F, device=dev, dtype=torch.int64).unsqueeze(1)
    flat = idm.reshape(-1)
    valid = flat >= 0
    gkey = ((feat << 40) | idm.clamp(min=0)).reshape(-1)
    e = torch.arange(n, device=dev, dtype=torch.int64)
    uniq, inv = torch.unique(gkey, return_inverse=True)
    first = torch.full((uniq.numel(),), n, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, inv[valid], e[valid], "amin")
    leader = valid & (first[inv] == e)
    lead_e = e[leader]                                   # ascending entry order
    lo = owner[lead_e]
    order = torch.argsort(lo, stable=True)
    sel, so = lead_e[order], lo[order]                   # leaders grouped by owner, first occurrence first
    counts = torch.bincount(lo, minlength=W)
    key = ((feat << 40) | torch.div(idm, W, rounding_mode="floor")).reshape(-1)
    m = sel.numel()
    if not capacity:
        p = torch.arange(m, device=dev, dtype=torch.int64)
        send_keys = key[sel]
    else:
        start = torch.cumsum(counts, 0) - counts
        rank_in = torch.arange(m, device=dev, dtype=torch.int64) - start[so]
        keep = rank_in < capacity
        p = torch.where(keep, so * capacity + rank_in, torch.full_like(rank_in, -1))
        send_keys = torch.full((W * capacity,), -1, dtype=torch.int64, device=dev)
        send_keys[p[keep]] = key[sel][keep]
        if overflow is not None and m:
            overflow |= (~keep).any().to(overflow.dtype)
    slot_of_key = torch.full((uniq.numel(),), -1, dtype=torch.int64, device=dev)
    slot_of_key[inv[sel]] = p
    pos_of = torch.where(valid, slot_of_key[inv], torch.full_like(e, -1))
    return send_keys, pos_of.reshape(F, B), None, counts


def segment_sum_torch(dstack: torch.Tensor, slots: Sequence[int], pos_of: torch.Tensor, n_send: int) -> torch.Tensor:
    """send[p] = sum over {(f, b): pos_of[f, b] == p} of dstack[b, slots[f]]: the gradient rows of a de-duplicated route."""
    B, F, D = dstack.shape
    send = torch.zeros((n_send, D), dtype=dstack.dtype, device=dstack.device)
    g = dstack[:, torch.tensor(list(slots), dtype=torch.int64, device=dstack.device)].transpose(0, 1).reshape(-1, D)
    pos = pos_of.reshape(-1)
    keep = pos >= 0
    send.index_add_(0, pos[keep], g[keep])
    return send


def route_build_torch(ids: Sequence[torch.Tensor], world_size: int, slots: Sequence[int], n_slots: int,
                      capacity: int = 0, overflow: Optional[torch.Tensor] = None, dedup: bool = False):
    """Framework-op statement of ``mh_route_build`` (same contract, any device): what the gloo tests inject and
    what the HIP kernel is checked against.  Entry e = f*B + b; stable order within an owner.  ``capacity`` > 0:
    fixed windows of that many slots per owner, padding -1, dropped requests ``pos_of`` = -1 (``overflow`` |= 1)."""
    idm = torch.stack([i.reshape(-1).to(torch.int64) for i in ids])  # [F, B]
    F, B = idm.shape
    owner = torch.remainder(idm, world_size).reshape(-1)
    if dedup:
        return _route_build_dedup_torch(idm, owner, world_size, capacity, overflow)
    order = torch.argsort(owner, stable=True)
    feat = torch.arange(F, device=idm.device, dtype=torch.int64).unsqueeze(1)
    key = ((feat << 40) | torch.div(idm, world_size, rounding_mode="floor")).reshape(-1)
    b = torch.arange(B, device=idm.device, dtype=torch.int64).unsqueeze(0)
    src = (b * n_slots + torch.tensor(list(slots), device=idm.device, dtype=torch.int64).unsqueeze(1)).reshape(-1)
    counts = torch.bincount(owner, minlength=world_size)
    n = order.numel()
    if not capacity:
        pos_of = torch.empty_like(order)
        pos_of[order] = torch.arange(n, device=order.device, dtype=order.dtype)
        return key[order], pos_of.reshape(F, B), src[order], counts
    first = torch.cumsum(counts, 0) - counts                       # first dense slot of every owner
    so = owner[order]
    rank_in = torch.arange(n, device=order.device, dtype=torch.int64) - first[so]
    keep = rank_in < capacity
    p = so * capacity + rank_in
    send_keys = torch.full((world_size * capacity,), -1, dtype=torch.int64, device=idm.device)
    src_row = torch.full((world_size * capacity,), -1, dtype=torch.int64, device=idm.device)
    send_keys[p[keep]] = key[order][keep]
    src_row[p[keep]] = src[order][keep]
    pos_of = torch.empty_like(order)
    pos_of[order] = torch.where(keep, p, torch.full_like(p, -1))
    if overflow is not None and n:
        overflow |= (~keep).any().to(overflow.dtype)
    return send_keys, pos_of.reshape(F, B), src_row, counts


def route_local_rows_torch(recv_keys: torch.Tensor, base: torch.Tensor, shard_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    f = (recv_keys >> 40).clamp(0, base.numel() - 1)
    r = recv_keys & ((1 << 40) - 1)
    ok = (recv_keys >= 0) & ((recv_keys >> 40) < base.numel())
    if shard_rows is not None:
        ok = ok & (r < shard_rows[f])
    return torch.where(ok, base[f] + r, torch.full_like(r, -1))


class ShardedEmbeddingGroup:
    """ALL row-sharded features of a model behind ONE route per step: the local shards of the DISTINCT tables live
    back to back in one [sum V_local, D] buffer, a request is the int64 key (feature << 40 | local_row), the owner
    turns it into a row of the concatenated buffer -> one gather launch, one fused update launch and three
    all-to-alls (ids, rows, row-gradients) per step regardless of the number of sharded tables.

    Two exchange modes:
      * dense (``capacity`` None): every rank sends exactly its requests; RCCL needs the per-peer counts on the host:
        ONE host sync per step.  Used for the first ``calibration`` steps, which record the largest per-owner count;
      * fixed capacity (after ``freeze_capacity``): every (sender, owner) window has ``capacity`` slots, padded with
        key -1 -> equal, host-known splits: no host sync, the whole step is a fixed launch sequence (hipGraph replay).
        A request beyond its window is dropped and sets the device flag ``overflow`` (``check_overflow()`` raises).

    ``tables``: the distinct tables, full (``presharded=False``: sliced here) or already the local shards;
    ``feature_table[f]``: index into ``tables`` of feature f (features sharing a table share its shard).
    ``route_fn`` / ``rows_fn``: ``ops.route_build`` / ``ops.route_local_rows`` (HIP) in production, the framework-op
    statements above in the gloo tests."""

    def __init__(self, tables: Sequence[torch.Tensor], gather_fn: Callable, update_fn: Callable, group=None,
                 route_fn: Optional[Callable] = None, rows_fn: Optional[Callable] = None,
                 feature_table: Optional[Sequence[int]] = None, presharded: bool = False,
                 global_rows: Optional[Sequence[int]] = None, capacity_factor: Optional[float] = None, calibration: int = 2,
                 dedup="auto", reduce_fn: Optional[Callable] = None):
        self.rank, self.world_size = world()
        self.group = group
        self.gather_fn, self.update_fn = gather_fn, update_fn
        # Per-(sender, owner) de-duplication of the requests (SparseOperationKit does it inside sok.lookup_sparse,
        # tf/distributed/embedding.py:144-148): True / False / "auto" = on when there is somebody to send to, and switched off at
        # the end of calibration if the batches turn out to hold (almost) no duplicates -- the ranks agree on that figure
        # (all-reduce), the hash pass and the segment sum of the gradient rows then cost more than the bytes they save.
        self.dedup = (self.world_size > 1) if dedup == "auto" else bool(dedup)
        self._dedup_auto = dedup == "auto"
        self.dedup_min_saving = 0.15              # "auto": keep de-duplicating if >= 15 % of the requests are duplicates
        self._seen_requests = self._seen_unique = 0
        self.reduce_fn = reduce_fn or segment_sum_torch
        self.route_fn = route_fn or route_build_torch
        self.rows_fn = rows_fn or route_local_rows_torch
        W = self.world_size
        shards = list(tables) if presharded else [shard_table(t, self.rank, W) for t in tables]
        self.global_rows = list(global_rows) if global_rows is not None else [t.shape[0] for t in tables]
        sizes = [sh.shape[0] for sh in shards]
        D = tables[0].shape[1]
        dev = tables[0].device
        self.local = torch.empty((max(sum(sizes), 1), D), dtype=torch.float32, device=dev)
        tbase, o = [], 0
        self.views: List[torch.Tensor] = []
        for sh, n in zip(shards, sizes):
            self.local[o:o + n] = sh
            self.views.append(self.local[o:o + n])
            tbase.append(o)
            o += n
        ft = list(range(len(tables))) if feature_table is None else list(feature_table)
        self.feature_table = ft
        self.base = torch.tensor([tbase[t] for t in ft], dtype=torch.int64, device=dev)          # per FEATURE
        self.shard_rows = torch.tensor([sizes[t] for t in ft], dtype=torch.int64, device=dev)    # per FEATURE
        self.state: Optional[torch.Tensor] = None
        self.state2: Optional[torch.Tensor] = None
        self._rows: Optional[torch.Tensor] = None
        # Slack of a fixed window over the largest per-owner count the calibration saw.  None (default): a statistical margin --
        # count + 8 sqrt(count) + 3 % -- because padding slots travel like requests (at 1.25 x a quarter of the row and
        # row-gradient bytes on the wire is padding) and an eager step that overflows its window loses nothing: the call is
        # served by the dense exchange and the window re-derived (`lossless`).  Steps captured into a hipGraph cannot branch on
        # the host: where those are asked for (MERLIN_HIP_GRAPH_DISTRIBUTED=1) the default stays the generous 1.25 x.
        if capacity_factor is None:
            import os as _os

            capacity_factor = 1.25 if _os.environ.get("MERLIN_HIP_GRAPH_DISTRIBUTED") == "1" else 0.0
        self.capacity_factor = float(capacity_factor)  # 0.0 = the statistical margin
        self.calibration = int(calibration)
        self.capacity: Optional[int] = None          # slots per (sender, owner) window once frozen
        self._capacity_n = 0                         # request count the window was derived from
        self._steps, self._max_count = 0, 0
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        self.check_every = 64                        # CAPTURED steps only: fixed-window calls between two automatic overflow checks
        # Eager calls never lose a request: right behind the route kernel -- before anything is exchanged -- the ranks agree
        # (all-reduce MAX of the overflow flag, one host read) whether ANY window overflowed; if one did, THIS call takes the dense
        # exchange (exact per-peer counts) and the window is re-derived from the counts it sees.  SparseOperationKit never drops
        # (tf/distributed/embedding.py:144-148); a step replayed from a captured graph cannot branch on the host, so it keeps the
        # flag and raises at the next check.
        self.lossless = True
        self.spills = 0                              # calls that overflowed their window and were served by the dense exchange
        self._since_check = 0
        self._parent: Optional["ShardedEmbeddingGroup"] = None
        self._deferred: List = []                    # (rows, gradient rows) handed over by aliases: ONE update per step
        self._pending_bwd = None

    def alias(self) -> "ShardedEmbeddingGroup":
        """A second route over the SAME local shards (own exchange state, own windows): list / ragged features request one row
        per VALUE, a different request count than the one-hot features of the group.  Its gradient rows are not applied by
        itself: ``backward_end`` hands them to this group, which applies everything that arrived for its shards in ONE fused
        update (Keras sums all IndexedSlices of a variable before one optimizer apply)."""
        import copy

        a = copy.copy(self)
        a._parent = self
        a._deferred = []
        a._pending_bwd = a._pending_lookup = None
        a._rows = None
        a.capacity, a._capacity_n, a._steps, a._max_count, a._since_check = None, 0, 0, 0, 0
        a.overflow = torch.zeros_like(self.overflow)
        # The request count of a ragged route is this rank's nnz: it differs from rank to rank and from step to step, so neither
        # "n <= the count the window was sized for" nor a window derived from a local n is a decision every rank takes alike --
        # ranks on different branches issue mismatched collectives and hang (round-3 advisor finding, reproduced on two gloo
        # ranks).  Alias routes therefore stay on the dense exchange (exact per-peer counts, one host read per step) for good.
        a._never_fixed = True
        a.dedup, a._dedup_auto = False, False  # one request per VALUE of a list: the owner-side update de-duplicates
        a._fmap_cache = {}
        return a

    # ---- capacity management ------------------------------------------------------------------------------------
    def freeze_capacity(self, n_requests: int, capacity: Optional[int] = None) -> int:
        """Switch to the fixed-capacity exchange.  Default window: the largest per-owner count seen during calibration
        (maximum over the ranks) plus the slack of ``capacity_factor`` (see __init__); one rank needs none (its window holds every
        request)."""
        W = self.world_size
        if capacity is None:
            if W == 1:
                capacity = n_requests
            else:
                seen = self._max_count if self._max_count else (n_requests + W - 1) // W
                slack = (int(seen * self.capacity_factor) if self.capacity_factor > 0 else
                         seen + int(8.0 * seen ** 0.5) + int(0.03 * seen))
                capacity = min(n_requests, slack + 1)
        self.capacity = (max(int(capacity), 1) + 63) // 64 * 64
        self._capacity_n = int(n_requests)
        return self.capacity

    def check_overflow(self) -> None:
        """One host read of the overflow flag (call it outside timed regions / at epoch ends)."""
        if int(self.overflow.item()) != 0:
            raise RuntimeError(f"row-sharded exchange: a per-owner window of {self.capacity} requests overflowed "
                               "(requests were dropped); raise capacity_factor or recalibrate")

    @property
    def graph_capturable(self) -> bool:
        return self.capacity is not None

    def _exchange(self, inp: torch.Tensor, n_out: int, out_splits=None, in_splits=None, async_op: bool = False):
        """all-to-all of the leading dimension; returns (received tensor, work handle or None).  One rank: the exchange
        is the identity -- the input is returned as is (no copy of the 134 MB row / gradient buffers)."""
        if self.world_size == 1:
            return inp, None
        out = torch.empty((n_out,) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
        if out_splits is None and inp.is_cuda and self.group is None:
            # fixed windows on GPUs: the collective behind the C ABI (mh_comm_alltoall: grouped ncclSend / ncclRecv)
            from . import comm as _comm

            c = _comm.default()
            if c is not None:
                work = c.alltoall_async(inp.contiguous(), out)
                if not async_op:
                    work.wait()
                    work = None
                return out, work
        work = dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=async_op)
        return out, work

    def lookup(self, ids: Sequence[torch.Tensor], scatter_into=None, features=None) -> Optional[torch.Tensor]:
        self.lookup_begin(ids, scatter_into, features=features)
        return self.lookup_end()

    def lookup_begin(self, ids: Sequence[torch.Tensor], scatter_into=None, layout=None, features=None) -> None:
        """``ids[f]`` is [B] for sharded feature f; returns [F_sh, B, D], or, with
        ``scatter_into = (stacked [B, F, D], slots, scatter_fn)``, writes feature f into ``stacked[:, slots[f]]``.
        ``layout = (slots, n_slots)``: the [B, n_slots, D] stack the GRADIENT will arrive in (``backward_begin(...,
        from_stacked=...)``) when the forward places the rows itself (``lookup_end(scatter=...)``)."""
        W = self.world_size
        F_sh, B = len(ids), ids[0].numel()
        n = F_sh * B
        if scatter_into is not None:
            stacked, slots, scatter_fn = scatter_into
            n_slots = stacked.shape[1]
        elif layout is not None:
            slots, n_slots = list(layout[0]), int(layout[1])
        else:
            slots, n_slots = list(range(F_sh)), F_sh
        never_fixed = getattr(self, "_never_fixed", False)
        if self.capacity is None and self.calibration <= 0 and not never_fixed:
            self.freeze_capacity(n)
        # The window was sized for _capacity_n requests.  A call with MORE requests (evaluate / predict with a larger batch,
        # a bigger train batch) would overflow it and silently drop requests: such a call takes the dense exchange (host-side
        # counts, exact).  For the one-hot routes that reach this point n = F x B is the same on every rank (equal per-rank
        # batches -- the fixed all-to-all needs that anyway), so all ranks take the same branch; routes whose n is rank-local
        # (ragged aliases) never leave the dense exchange.
        fixed = self.capacity is not None and n <= self._capacity_n and not never_fixed
        capturing = send_capturing()
        if fixed and self.check_every > 0 and (capturing or not self.lossless):
            self._since_check += 1
            if self._since_check >= self.check_every and not capturing:
                self._since_check = 0
                self.check_overflow()  # one host read every check_every steps: a dropped request never goes unnoticed for long
        if fixed and capturing:
            self._captured_once = True
        elif fixed and self.lossless and W > 1 and getattr(self, "_captured_once", False):
            # Replays of a captured step cannot branch on the host: they leave the flag set and go on.  An eager call that
            # follows must not mistake that flag for its own overflow (it would zero it, re-derive the window, and the requests
            # dropped inside the replays would never be reported -- round-4 advisor finding): read it FIRST and raise.
            self.check_overflow()
        if fixed:                       # ---- fixed windows ----
            cap = self.capacity
            send_keys, pos_of, src_row, _ = self._route(ids, W, slots, n_slots, cap, self.overflow)
            if self.lossless and W > 1 and not capturing:
                # nothing has been exchanged yet: agree on the outcome of the route (every rank takes the same branch)
                dist.all_reduce(self.overflow, op=dist.ReduceOp.MAX, group=self.group)
                if int(self.overflow.item()) != 0:
                    self.overflow.zero_()
                    self.spills += 1
                    # this call is served exactly by the dense exchange below, which also re-derives the window from the
                    # counts it sees (max over the ranks x capacity_factor): the skew that overflowed it is the new normal
                    self.capacity, self._capacity_n = None, 0
                    self._steps = max(self.calibration - 1, 0)
                    fixed = False
        if fixed:
            self._send_counts = self._recv_counts = None
            n_recv = W * cap
        else:                           # ---- dense: per-peer counts on the host (calibration steps) ----
            send_keys, pos_of, src_row, send_counts = self._route(ids, W, slots, n_slots)
            recv_counts = torch.empty_like(send_counts)
            if W > 1:
                dist.all_to_all_single(recv_counts, send_counts, group=self.group)
            else:
                recv_counts.copy_(send_counts)
            both = torch.stack([send_counts, recv_counts]).tolist()  # the one host sync of a calibration step
            self._send_counts, self._recv_counts = both[0], both[1]
            n_recv = sum(self._recv_counts)
            if self.dedup:  # the route filled the first sum(counts) slots of an n-entry buffer
                send_keys = send_keys[:sum(self._send_counts)]
            self._seen_requests += n
            self._seen_unique += sum(self._send_counts)
            self._max_count = max(self._max_count, max(self._send_counts), max(self._recv_counts))
            self._steps += 1
            if self.capacity is None and self._steps >= self.calibration and not never_fixed:
                if W > 1:  # every rank must choose the same window (and the same answer to "does de-duplication pay")
                    m = torch.tensor([self._max_count, self._seen_unique, self._seen_requests], dtype=torch.int64,
                                     device=send_keys.device)
                    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)  # the rank whose batches held the most distinct keys
                    both = m.tolist()
                    self._max_count, uniq, req = int(both[0]), int(both[1]), int(both[2])
                else:
                    uniq, req = self._seen_unique, self._seen_requests
                if self.dedup and self._dedup_auto and req > 0 and 1.0 - uniq / req < self.dedup_min_saving:
                    # (almost) no duplicates: calibrate again without the de-duplication (its windows hold requests, not keys)
                    self.dedup = False
                    self._steps, self._max_count = 0, 0
                else:
                    self._freeze_after = n
                self._seen_requests = self._seen_unique = 0
        if features is not None and list(features) != list(range(F_sh)):
            # the route numbers the id columns 0 .. F_sh - 1; ``features`` says which features of the GROUP they are
            # one tensor per (features, device), built once: a per-lookup torch.tensor(list) is a pageable host-to-device copy
            # and a synchronisation in every step, and illegal while a step is being captured
            cache = self.__dict__.setdefault("_fmap_cache", {})
            fkey = (tuple(int(f) for f in features), str(send_keys.device))
            fmap = cache.get(fkey)
            if fmap is None:
                fmap = cache[fkey] = torch.tensor(list(fkey[0]), dtype=torch.int64, device=send_keys.device)
            low = (1 << 40) - 1
            send_keys = torch.where(send_keys >= 0, (fmap[(send_keys >> 40).clamp(min=0)] << 40) | (send_keys & low), send_keys)
        self._pos_of, self._src_row, self._n_send = pos_of, src_row, send_keys.numel()
        self._dedup_call = src_row is None  # this call's route was de-duplicated: the backward is a segment sum over pos_of
        self._fwd_layout = (list(slots), int(n_slots))
        recv_keys, _ = self._exchange(send_keys, n_recv, self._recv_counts, self._send_counts)
        self._rows = self.rows_fn(recv_keys, self.base, self.shard_rows)  # rows of the concatenated local buffer (-1: none)
        rows = self.gather_fn(self.local, self._rows).contiguous()
        # the row exchange runs on RCCL's stream: whatever the caller enqueues before lookup_end() overlaps it
        back, work = self._exchange(rows, self._n_send, self._send_counts, self._recv_counts, async_op=True)
        self._pending_lookup = (work, back, rows, scatter_into, F_sh, B)

    def _route(self, ids, W, slots, n_slots, cap: int = 0, overflow=None):
        if self.dedup:
            return self.route_fn(ids, W, slots, n_slots, cap, overflow, dedup=True)
        return self.route_fn(ids, W, slots, n_slots, cap, overflow)

    def prepare_update(self) -> None:
        """Training step, HIP path: the owner-side rows of this step's requests are known since ``lookup_begin`` -- start the
        id-only half of their fused update (sort + piece list) on the "sort" side stream now, beside the rest of the forward."""
        from . import ops

        self._prepared_update = None
        if self._rows is None or not self._rows.is_cuda or not ops.SIDE.active("sort"):
            return
        with ops.SIDE.on("sort_sharded", keep=(self._rows,)):
            self._prepared_update = ops.embedding_gather_backward_prepare([self.local], [self._rows], tag=":sharded")

    def lookup_end(self, scatter: Optional[Callable] = None) -> Optional[torch.Tensor]:
        """``scatter(back, [pos_of[f] ...])``: the caller places the returned rows itself (e.g. into a concat buffer)."""
        work, back, _rows_alive, scatter_into, F_sh, B = self._pending_lookup
        self._pending_lookup = None
        if work is not None:
            work.wait()
        pos_of, D = self._pos_of, back.shape[1]
        if scatter is not None:
            scatter(back, [pos_of[f] for f in range(F_sh)])
            return None
        if scatter_into is None:
            out = self.gather_fn(back, pos_of.reshape(-1))  # dropped requests (pos_of -1) read as zero rows
            return out.reshape(F_sh, B, D)
        stacked, slots, scatter_fn = scatter_into
        # returned rows arrive in owner order; request (f, b) sits at pos_of[f, b]: ONE multi-"table" gather
        # writes them straight into their stack slots (no un-permute pass, no index_put)
        scatter_fn([back] * F_sh, [pos_of[f] for f in range(F_sh)], stacked, slots)
        return None

    def backward_update(self, grad: torch.Tensor, from_stacked=None) -> None:
        self.backward_begin(grad, from_stacked)
        self.backward_end()

    def backward_begin(self, grad: torch.Tensor, from_stacked=None) -> None:
        """``grad`` [F_sh, B, D] in the order of ``lookup``; or ``from_stacked = (dstack [B, F, D], slots,
        gather_fn)``: the gradient rows are pulled out of dstack already in owner order by one gather launch
        (``src_row`` of the route; padding slots read a zero row).  Starts the gradient exchange; ``backward_end``
        applies the fused update."""
        if from_stacked is None:
            D = grad.shape[-1]
            flat = grad.reshape(-1, D)
            send = torch.zeros((self._n_send, D), dtype=grad.dtype, device=grad.device)
            pos = self._pos_of.reshape(-1)
            keep = pos >= 0
            if self._dedup_call:
                send.index_add_(0, pos[keep], flat[keep])  # equal requests share a slot: their gradient rows are summed here
            else:
                send[pos[keep]] = flat[keep]
        elif self._dedup_call:
            dstack, slots, _ = from_stacked
            send = self.reduce_fn(dstack, list(slots), self._pos_of, self._n_send)
        else:
            dstack, slots, gather_fn = from_stacked
            B, F, D = dstack.shape
            src = self._src_row
            fslots, fn = self._fwd_layout
            if list(slots) != fslots or F != fn:
                # the gradient stack is laid out differently from the forward stack the route was built for:
                # src = b * fn + forward slot  ->  b * F + backward slot (padding stays -1)
                remap = torch.full((fn,), -1, dtype=torch.int64, device=src.device)
                remap[torch.tensor(fslots, dtype=torch.int64, device=src.device)] = torch.tensor(
                    list(slots), dtype=torch.int64, device=src.device)
                safe = src.clamp(min=0)
                src = torch.where(src >= 0, torch.div(safe, fn, rounding_mode="floor") * F + remap[safe % fn], src)
            send = gather_fn(dstack.reshape(B * F, D), src)  # [n_send, D] in owner order
        g, work = self._exchange(send.contiguous(), self._rows.numel(), self._recv_counts, self._send_counts, async_op=True)
        self._pending_bwd = (work, g, send)

    def backward_end(self) -> None:
        parts = []
        if self._pending_bwd is not None:
            work, g, _send_alive = self._pending_bwd
            self._pending_bwd = None
            if work is not None:
                work.wait()
            parts.append((self._rows, g))
        if self._parent is not None:      # an alias: the owner group applies everything in one update
            self._parent._deferred.extend(parts)
            parts = []
        else:
            parts += self._deferred
            self._deferred = []
        if len(parts) == 1:
            self.update_fn(self.local, self.state, parts[0][0], parts[0][1])
        elif parts:
            self.update_fn(self.local, self.state, torch.cat([r for r, _ in parts]), torch.cat([g for _, g in parts]))
        fa = getattr(self, "_freeze_after", None)
        if fa is not None:  # calibration done: the next step runs on fixed windows
            self._freeze_after = None
            self.freeze_capacity(fa)


# ------------------------------------------------------------------------------------------------
# dense gradients
# ------------------------------------------------------------------------------------------------
class _Phases:
    """Time split of a sharded step (SURVEY 8e "Reported": gather, a2a, MLP, interaction, allreduce).  Off by default
    (``with phase(...)`` is then a no-op); ``PHASES.start()`` makes every phase synchronise the device on entry and exit
    and accumulate its wall time -- phases then run one after the other, so the split shows what each part COSTS, not the
    overlapped step time.  bench.py runs a few profiled steps after the timed region at N > 1."""

    def __init__(self):
        self.acc = None
        self.steps = 0

    def start(self) -> None:
        self.acc, self.steps = {}, 0

    def stop(self) -> Dict[str, float]:
        acc, n = self.acc or {}, max(self.steps, 1)
        self.acc = None
        return {k: v / n * 1e3 for k, v in acc.items()}  # ms per step

    class _Ctx:
        def __init__(self, owner, name):
            self.owner, self.name = owner, name

        def __enter__(self):
            import time

            torch.cuda.synchronize()
            self.t0 = time.perf_counter()

        def __exit__(self, *exc):
            import time

            torch.cuda.synchronize()
            a = self.owner.acc
            a[self.name] = a.get(self.name, 0.0) + time.perf_counter() - self.t0
            return False

    class _Null:
        def __enter__(self):
            return None

        def __exit__(self, *exc):
            return False

    _NULL = _Null()

    def __call__(self, name: str):
        if self.acc is None or not torch.cuda.is_available():
            return self._NULL
        return _Phases._Ctx(self, name)


PHASES = _Phases()
phase = PHASES


def send_capturing() -> bool:
    """True while the current HIP stream is being captured into a graph (no host reads allowed then)."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def allreduce_flat_(flat: torch.Tensor, group=None, async_op: bool = False):
    """In-place SUM of one flat fp32 bucket across ranks.  On RCCL: reduce-scatter + all-gather, so every xGMI
    link carries 1/W of the bucket per phase (needs ``numel % W == 0``; callers pad the bucket).
    ``async_op``: returns the work handles (wait on all of them before reading ``flat``)."""
    rank, W = world()
    if W == 1 or flat.numel() == 0:
        return []
    n = flat.numel()
    if flat.is_cuda and group is None:  # GPUs: mh_allreduce_dense behind the C ABI (reduce-scatter + all-gather over RCCL)
        from . import comm as _comm

        c = _comm.default()
        if c is not None:
            work = c.allreduce_async(flat)
            if async_op:
                return [work]
            work.wait()
            return []
    if dist.get_backend(group) == "nccl" and n % W == 0:
        shard = torch.empty(n // W, dtype=flat.dtype, device=flat.device)
        works = [dist.reduce_scatter_tensor(shard, flat, group=group, async_op=async_op),
                 dist.all_gather_into_tensor(flat, shard, group=group, async_op=async_op)]
    else:
        works = [dist.all_reduce(flat, group=group, async_op=async_op)]
    return [w for w in works if w is not None] if async_op else []


def allreduce_sum_(tensors: Sequence[torch.Tensor], group=None) -> None:
    """Sum a list of tensors across ranks through ONE flat bucket (in place)."""
    rank, W = world()
    if W == 1 or not tensors:
        return
    n = sum(t.numel() for t in tensors)
    flat = torch.zeros((n + W - 1) // W * W, dtype=tensors[0].dtype, device=tensors[0].device)
    torch.cat([t.reshape(-1) for t in tensors], out=flat[:n])
    allreduce_flat_(flat, group)
    o = 0
    for t in tensors:
        t.copy_(flat[o:o + t.numel()].reshape(t.shape))
        o += t.numel()


def broadcast_parameters(params: Sequence[torch.Tensor], src: int = 0, group=None) -> None:
    """BroadcastGlobalVariablesCallback(0) (`tf/models/base.py:1472-1473`)."""
    if world()[1] == 1:
        return
    for p in params:
        dist.broadcast(p, src, group=group)


# ------------------------------------------------------------------------------------------------
# DLRM data-parallel / model-parallel hybrid step
# ------------------------------------------------------------------------------------------------
def _hip_group_fns(owner):
    """gather / fused-update callables over the concatenated local shards (HIP ops); ``owner.model.optimizer`` and
    ``owner.group_sh`` are read at call time."""
    from . import ops

    def gather_fn(table, rows):
        return ops.embedding_gather([table], [rows])[:, 0]

    def update_fn(table, state, rows, grads):
        opt = owner.model.optimizer
        D = table.shape[1]
        g3 = grads.reshape(grads.shape[0], 1, D).contiguous()
        st2 = owner.group_sh.state2  # LazyAdam second moment of the local shards
        prep = getattr(owner.group_sh, "_prepared_update", None)  # id-only half already done beside the forward
        owner.group_sh._prepared_update = None
        ops.embedding_gather_backward([table], None if state is None else [state], [rows], g3, [0], opt.name,
                                      opt.learning_rate, opt.epsilon, None if st2 is None else [st2],
                                      opt.beta_1, opt.beta_2, opt.lr_device, prepared=prep)

    return gather_fn, update_fn


def _hip_segment_sum(dstack, slots, pos_of, n_send):
    """Gradient rows of a de-duplicated route: ``send[p] = sum of dstack[b, slots[f]] over pos_of[f, b] == p`` by the fused sparse
    update itself (SGD, lr = -1, onto zeros: its sort + piece reduction is the segment sum; slots < 0 are skipped)."""
    from . import ops

    B, F, D = dstack.shape
    send = torch.zeros((n_send, D), dtype=dstack.dtype, device=dstack.device)
    if n_send and B:
        dstack = dstack.contiguous()
        step = _lib_max_features()
        for f0 in range(0, len(slots), step):  # one launch holds at most that many id columns; the launches add into `send`
            fs = range(f0, min(f0 + step, len(slots)))
            ops.embedding_gather_backward([send] * len(fs), None, [pos_of[f] for f in fs], dstack,
                                          [int(slots[f]) * D for f in fs], "sgd", -1.0, 0.0)
    return send


def _lib_max_features() -> int:
    from . import _lib

    return _lib.MAX_FEATURES - 1


def _build_group(owner, emb, names, group, capacity_factor, calibration, dedup="auto"):
    """One ShardedEmbeddingGroup over the DISTINCT tables of the sharded features ``names`` of an EmbeddingsBlock
    (features sharing a table -- same Parameter -- share one shard); rebinds every table to a view of its shard."""
    from . import ops

    tabs, index, ft = [], {}, []
    for n in names:
        t = emb.feature_table[n]
        if id(t) not in index:
            index[id(t)] = len(tabs)
            tabs.append(t)
        ft.append(index[id(t)])
    gather_fn, update_fn = _hip_group_fns(owner)
    pre = [getattr(t, "shard", None) is not None for t in tabs]
    if any(pre) and not all(pre):
        raise ValueError("either every sharded table is built as a shard (distributed.sharded_tables) or none is")
    grp = ShardedEmbeddingGroup([t.table.data for t in tabs], gather_fn, update_fn, group, route_fn=ops.route_build,
                                rows_fn=ops.route_local_rows, feature_table=ft, presharded=all(pre) and len(pre) > 0,
                                global_rows=[t.input_dim for t in tabs], capacity_factor=capacity_factor,
                                calibration=calibration, reduce_fn=_hip_segment_sum, dedup=dedup)
    for t, view in zip(tabs, grp.views):
        t.table.data = view  # drop the replicated copy; keep a view of the local shard
        t.shard = (grp.rank, grp.world_size)
    return grp, tabs


class DistributedDLRM:
    """Wraps an ``mm.DLRMModel`` built on every rank: tables with >= ``shard_threshold`` rows keep only
    their local row shard (all-to-all lookup), the rest stay replicated.  Build the model under
    ``distributed.sharded_tables(threshold)`` and the large tables are ALLOCATED as shards (a 100M-row table is never
    materialised on one GPU).  After ``calibration`` steps the exchange runs on fixed windows with no host sync
    (``graph_capturable``); one rank (``force_shard``) uses them from the first step."""

    def __init__(self, model, shard_threshold: int = 200_000, group=None, force_shard: bool = False,
                 capacity_factor: Optional[float] = None, calibration: int = 2, dedup="auto"):
        self.model = model
        self.body = model.body
        self.group = group
        self.rank, self.world_size = world()
        emb = self.body.embeddings
        active = self.world_size > 1 or force_shard
        names = [n for n in self.body.cat_names if active and (emb.feature_table[n].input_dim >= shard_threshold
                                                               or getattr(emb.feature_table[n], "shard", None) is not None)]
        self.group_sh: Optional[ShardedEmbeddingGroup] = None
        self.sharded_tables: List = []
        if names:
            self.group_sh, self.sharded_tables = _build_group(self, emb, names, group, capacity_factor,
                                                              0 if self.world_size == 1 else calibration, dedup)
        self.sharded_names = names
        self._bucket: Optional[torch.Tensor] = None
        self.replicated = [n for n in self.body.cat_names if n not in names]
        rep_tabs, seen = [], set()
        for n in self.replicated:  # distinct replicated tables (shared tables appear once)
            t = emb.feature_table[n].table
            if id(t) not in seen:
                seen.add(id(t))
                rep_tabs.append(t)
        self.rep_tabs = rep_tabs
        dense = [p.data for p in model.parameters() if not p.sparse]
        broadcast_parameters(dense + [t.data for t in rep_tabs], 0, group)

    @property
    def sharded(self) -> Dict[str, torch.Tensor]:
        """name -> local shard of every sharded feature's table."""
        emb = self.body.embeddings
        return {n: emb.feature_table[n].table.data for n in self.sharded_names}

    @property
    def graph_capturable(self) -> bool:
        """True once every per-step collective has host-known sizes (fixed windows).  At more than one rank the capture
        of RCCL collectives into a hipGraph is opt-in (MERLIN_HIP_GRAPH_DISTRIBUTED=1)."""
        import os

        ok = self.group_sh is None or self.group_sh.graph_capturable
        return ok and (self.world_size == 1 or os.environ.get("MERLIN_HIP_GRAPH_DISTRIBUTED") == "1")

    def check_overflow(self) -> None:
        if self.group_sh is not None:
            self.group_sh.check_overflow()

    # ---- checkpoint / resume: the row shards and their optimizer state are per rank -----------------------------------
    def save_weights(self, path) -> None:
        """Rank 0 writes the replicated part (``Model.save_weights`` layout, sharded tables left out) to ``path``; every
        rank writes its shards and their optimizer state to ``path.shard<rank>of<W>.npz``."""
        import numpy as np

        path = str(path)
        shard_ids = {id(t.table) for t in self.sharded_tables}
        if self.rank == 0:
            self.model.save_weights(path, skip=shard_ids)
        arrays = {}
        gs = self.group_sh
        if gs is not None:
            arrays["local"] = gs.local.detach().cpu().numpy()
            if gs.state is not None:
                arrays["state"] = gs.state.detach().cpu().numpy()
            if gs.state2 is not None:
                arrays["state2"] = gs.state2.detach().cpu().numpy()
        np.savez(f"{path}.shard{self.rank}of{self.world_size}.npz", **arrays)
        if self.world_size > 1:
            dist.barrier(group=self.group)

    def load_weights(self, path) -> None:
        """Inverse of ``save_weights`` at the SAME world size; values are copied INTO the live tensors, so a captured
        hipGraph keeps replaying on the restored state."""
        import numpy as np

        path = str(path)
        shard_ids = {id(t.table) for t in self.sharded_tables}
        self.model.load_weights(path if path.endswith(".npz") else path + ".npz", skip=shard_ids)
        z = np.load(f"{path}.shard{self.rank}of{self.world_size}.npz")
        gs = self.group_sh
        if gs is not None:
            gs.local.copy_(torch.from_numpy(z["local"]).to(gs.local.device))
            for key in ("state", "state2"):
                if key in z.files:
                    cur = getattr(gs, key)
                    val = torch.from_numpy(z[key]).to(gs.local.device)
                    if cur is None:
                        setattr(gs, key, val)
                    else:
                        cur.copy_(val)

    # forward of the DLRM body with sharded lookups
    def forward_body(self, inputs, head=None, training: bool = False):
        from . import ops

        body = self.body
        for n in self.sharded_names:
            if not body.embeddings._is_onehot(inputs[n]):
                raise NotImplementedError(f"DistributedDLRM looks row-sharded tables up one id per sample; {n!r} is a list / ragged "
                                          "feature: wrap the model in distributed.DistributedModel (alias routes per list feature)")
        B = inputs[body.cat_names[0]].shape[0]
        dev = inputs[body.cat_names[0]].device
        F, D = body.num_features, body.dim
        if dev.type == "cuda" and body._fusable(inputs):
            return self._forward_body_fused(inputs, head, training)
        # 1. route FIRST: its one host sync then waits only for the tiny bucketing kernels, and everything
        #    below is enqueued back to back; the row all-to-all overlaps the bottom MLP / replicated gather
        stacked = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        if self.group_sh is not None:
            def scatter_fn(tabs, idx, out, slots):
                ops.embedding_gather(tabs, idx, out=out, out_slot=slots)

            self.group_sh.lookup_begin([inputs[n] for n in self.sharded_names],
                                       scatter_into=(stacked, [body.slots[n] for n in self.sharded_names], scatter_fn))
        x = body.continuous(inputs)
        from .blocks import mlp_forward

        tail = stacked[:, body.slots["bottom_block"]]
        mlp_forward(body.bottom_block.layers, x, out_last=tail)  # small Dense runs fused; the last layer writes its stack slot
        emb = body.embeddings
        if self.replicated:
            ops.embedding_gather([emb.feature_table[n].table.data for n in self.replicated],
                                 [inputs[n] for n in self.replicated], out=stacked,
                                 out_slot=[body.slots[n] for n in self.replicated])
        if self.group_sh is not None:
            self.group_sh.lookup_end()  # rows are in: scatter them into their stack slots
        emb._last = {n: inputs[n] for n in body.cat_names}
        body._stacked = stacked
        body._fused = False  # the sharded path materialises the stacked tensor
        P = F * (F - 1) // 2
        width = P + D
        ld = (width + 3) // 4 * 4
        buf = torch.empty((B, ld), dtype=torch.float32, device=dev)
        if ld != width:
            ops.zero_pad_columns(buf, width)
        top_in = buf[:, :width]
        body.interaction.forward(stacked, tail, out=top_in)
        body._top_in = top_in
        return body._top(top_in, head)  # the Dense(1, sigmoid) head rides on the top MLP's fused chain

    def _forward_body_fused(self, inputs, head=None, training: bool = False):
        """The fused gather -> interaction kernel on the sharded path: a sharded feature's slot is the buffer of rows that
        came back from the owners, indexed by the position of request (f, b) in it (``pos_of``; a dropped request is -1 and
        reads as a zero row) -- no stacked [B, F, D] tensor, no scatter of the returned rows, and the backward re-gathers
        from the same buffer.  Replicated features are ordinary (table, ids) slots."""
        from . import ops
        from .blocks import mlp_forward

        body = self.body
        emb = body.embeddings
        B = inputs[body.cat_names[0]].shape[0]
        dev = inputs[body.cat_names[0]].device
        F, D = body.num_features, body.dim
        gs = self.group_sh
        if gs is not None:  # route first: the row all-to-all overlaps the bottom MLP
            with phase("a2a_ids_route_owner_gather"):  # route build, ids all-to-all, owner-side gather, rows all-to-all issued
                gs.lookup_begin([inputs[n] for n in self.sharded_names],
                                layout=([body.slots[n] for n in self.sharded_names], F))
        with phase("mlp_bottom_fwd"):
            dense = mlp_forward(body.bottom_block.layers, body.continuous(inputs))
        got = {}
        if gs is not None:
            with phase("a2a_rows_wait"):
                gs.lookup_end(scatter=lambda back, pos: got.update(back=back, pos=pos))
        idt = inputs[body.cat_names[0]].dtype
        sh_index = {n: i for i, n in enumerate(self.sharded_names)}
        # positions of the sharded features' requests in the returned-row buffer, in the id dtype of the other slots: ONE
        # conversion of the [F_sh, B] tensor (per feature it was eight 5 us launches in front of the fused kernel)
        pos_all = None
        if gs is not None:
            pos_all = gs._pos_of if gs._pos_of.dtype == idt else gs._pos_of.to(idt)  # [F_sh, B]
        slot_tables, slot_ids = [], []
        for k in body.stack_order:
            if k == "bottom_block":
                slot_tables.append(None)
                slot_ids.append(None)
            elif k in sh_index:
                slot_tables.append(got["back"])
                slot_ids.append(pos_all[sh_index[k]])
            else:
                slot_tables.append(emb.feature_table[k].table.data)
                slot_ids.append(inputs[k])
        emb._last = {n: inputs[n] for n in body.cat_names}
        P = F * (F - 1) // 2
        width = P + D
        ld = (width + 3) // 4 * 4
        # persistent per wrapper, like the one-GPU block's: the alignment column behind `width` is zeroed ONCE, not in every step
        buf = getattr(self, "_top_buf", None)
        if buf is None or buf.shape != (B, ld) or buf.device != dev:
            ops.park_replaced(buf)  # a captured step may still address the old one
            buf = self._top_buf = torch.zeros((B, ld), dtype=torch.float32, device=dev)
        ops.note_captured(buf)
        top_in = buf[:, :width]
        with phase("gather_interaction_fwd"):
            ops.dlrm_interaction_fused(slot_tables, slot_ids, dense, append_dense=True, out=top_in)
        self._prep_rep = None
        if training and ops.SIDE.active("sort"):
            # the id-only halves (sort + piece list) of BOTH sparse updates of the step start here, behind the HBM-bound
            # gather -> interaction kernel, beside the MFMA-bound top MLP (same placement as the one-GPU step)
            if gs is not None:
                gs.prepare_update()
            if self.replicated:
                with ops.SIDE.on("sort", keep=tuple(inputs[n] for n in self.replicated)):
                    self._prep_rep = ops.embedding_gather_backward_prepare(
                        [emb.feature_table[n].table.data for n in self.replicated], [inputs[n] for n in self.replicated], tag=":rep")
        body._fused = True
        body._slots_ctx = (slot_tables, slot_ids, dense)  # keeps the returned rows alive for the backward's re-gather
        body._top_in = top_in
        with phase("mlp_top_fwd"):
            return body._top(top_in, head)

    def __call__(self, inputs):
        from .models import prepare_features

        return self.forward_body(prepare_features(inputs), head=self.model.output.to_call)

    def train_step(self, inputs, targets):
        """Gradients are partial sums of the GLOBAL-mean loss (scale 1/(B*W) at the loss), so every
        cross-rank reduction is a plain SUM."""
        from . import ops
        from .models import prepare_features

        model, body = self.model, self.body
        if model.optimizer is None:
            model.compile()
        opt = model.optimizer
        x = prepare_features(inputs)
        p = self.forward_body(x, head=model.output.to_call, training=True)
        B = p.shape[0]
        loss, dlogit = ops.bce(p, targets, need_grad=True)
        if self.world_size > 1:
            dlogit = dlogit / self.world_size
        if PHASES.acc is not None:
            PHASES.steps += 1
        with ops.SIDE.deferred():  # side work (dW GEMMs) is joined below, right before the bucket reads the gradients
            with phase("mlp_interaction_bwd"):
                body.backward(dlogit)  # head + top MLP + interaction; leaves (dstack, offsets) pending on the embeddings block
            dstack, offsets = body.embeddings._pending
            body.embeddings._pending = None
            D = body.dim
            emb = body.embeddings
            opt.begin_step(dstack.device)  # Adam: advance the on-device step / bias-corrected lr once per step
            # 1. sharded tables: route the gradient rows to their owners (fused update there, in step 3)
            if self.group_sh is not None:
                gs = self.group_sh
                if opt.name == "adagrad" and gs.state is None:
                    gs.state = torch.full_like(gs.local, opt.initial_accumulator_value)
                if opt.name == "adam" and gs.state is None:
                    gs.state, gs.state2 = torch.zeros_like(gs.local), torch.zeros_like(gs.local)
                # starts the gradient all-to-all; it overlaps the replicated-table gradient pass below
                with phase("a2a_grads"):
                    gs.backward_begin(None, from_stacked=(dstack, [body.slots[n] for n in self.sharded_names],
                                                          lambda tab, idx: ops.embedding_gather([tab], [idx])[:, 0]))
            # 2 + 3. ONE persistent flat bucket [MLP / head gradients | dense [V, D] gradients of the replicated
            #    tables]: the table part is zeroed by one fill and accumulated by the fused backward (SGD, lr = -1),
            #    the MLP part is packed by one cat; after the in-place reduction the gradients are VIEWS of the bucket
            rep_tabs = self.rep_tabs  # DISTINCT replicated tables: a shared table has one gradient and one update
            dense = [q for q in model.parameters() if not q.sparse and q.grad is not None]
            n_dense = sum(q.grad.numel() for q in dense)
            n_rep = sum(t.data.numel() for t in rep_tabs)
            n_head = (n_dense + 1 + 63) // 64 * 64  # + one slot for the loss; table gradients start 256-byte aligned
            total = (n_head + n_rep + 63) // 64 * 64
            if self._bucket is None or self._bucket.numel() != total:
                from . import ops as _ops

                _ops.park_replaced(self._bucket)
                self._bucket = torch.zeros(total, dtype=torch.float32, device=dstack.device)
            bucket = self._bucket
            if bucket.is_cuda:
                from . import ops as _ops2

                _ops2.note_captured(bucket)
            rep_grads, o, grad_of = [], n_head, {}
            for t in rep_tabs:
                rep_grads.append(bucket[o:o + t.data.numel()].view_as(t.data))
                grad_of[id(t)] = rep_grads[-1]
                o += t.data.numel()
            if rep_tabs:
                bucket[n_head:n_head + n_rep].zero_()
                # per FEATURE: features sharing a table pass the same gradient buffer and are summed into it
                prep, self._prep_rep = getattr(self, "_prep_rep", None), None
                with phase("replicated_table_grads"):
                    ops.embedding_gather_backward([grad_of[id(emb.feature_table[n].table)] for n in self.replicated], None,
                                                  [x[n] for n in self.replicated], dstack,
                                                  [offsets[n] for n in self.replicated], "sgd", -1.0, 0.0, prepared=prep)
            ops.SIDE.join()  # the dW / db GEMMs ran on their side stream: the bucket below reads them
        # (packed at every world size, so that the single-GPU parity test walks the same code as an 8-GPU job)
        torch.cat([q.grad.reshape(-1) for q in dense] + [loss.detach().reshape(1)], out=bucket[:n_dense + 1])
        with phase("allreduce_issue"):
            works = allreduce_flat_(bucket, self.group, async_op=True)
        if self.group_sh is not None:
            with phase("a2a_grads_wait_sharded_update"):
                self.group_sh.backward_end()  # fused update of the local shards overlaps the bucket reduction
        with phase("allreduce_wait"):
            for w in works:
                w.wait()
        loss = bucket[n_dense] / self.world_size
        o = 0
        for q in dense:
            q.grad = bucket[o:o + q.data.numel()].view_as(q.data)
            o += q.data.numel()
        for t, g in zip(rep_tabs, rep_grads):
            t.grad = g
        with phase("dense_optimizer"):
            ops.dense_optimizer_step_multi(opt, dense + rep_tabs)  # one launch per 64 tensors
        return loss

    def exchange_bytes_per_step(self, batch: int) -> Dict[str, int]:
        """Bytes this rank SENDS per train step: ids all-to-all, rows all-to-all, row-gradient all-to-all over the
        fixed windows (padding included), and the dense bucket (reduce-scatter + all-gather: 2 (W - 1) / W of it)."""
        W, gs = self.world_size, self.group_sh
        out = {"a2a_ids": 0, "a2a_rows": 0, "a2a_grads": 0, "allreduce": 0}
        if gs is not None and W > 1:
            cap = gs.capacity if gs.capacity is not None else (len(self.sharded_names) * batch + W - 1) // W
            D = gs.local.shape[1]
            out["a2a_ids"] = (W - 1) * cap * 8
            out["a2a_rows"] = out["a2a_grads"] = (W - 1) * cap * D * 4
            # de-duplicated route: a window holds the distinct keys of a (sender, owner) pair, not its requests
            out["dedup"] = bool(gs.dedup)
            out["window_slots"] = int(cap)
            out["requests_per_owner"] = (len(self.sharded_names) * batch + W - 1) // W
        if self._bucket is not None and W > 1:
            out["allreduce"] = int(2 * (W - 1) / W * self._bucket.numel() * 4)
        return out


# ------------------------------------------------------------------------------------------------
# generic data-parallel wrapper: any RankingModel / RetrievalModel over EmbeddingsBlocks (TwoTower C3, DCN-v2 C5)
# ------------------------------------------------------------------------------------------------
class _ShardedEmbeddings:
    """Installs the row-sharded exchange on ONE EmbeddingsBlock by replacing two of its methods on the instance:
    ``gather_into`` (forward: sharded features arrive through the group's all-to-all, the rest by the local gather)
    and ``_apply_sparse_now`` (backward: sharded gradient rows go to their owners; gradients of replicated tables are
    accumulated as dense [V, D] buffers inside the owner's flat bucket)."""

    def __init__(self, owner, emb, names, group, capacity_factor, calibration, dedup="auto"):
        from .inputs import EmbeddingsBlock

        self.owner, self.emb = owner, emb
        self.groups: Dict[int, Tuple[ShardedEmbeddingGroup, List[str]]] = {}
        by_dim: Dict[int, List[str]] = {}
        for n in names:
            by_dim.setdefault(emb.feature_table[n].dim, []).append(n)
        self.sharded_tables = []
        for d, ns in by_dim.items():
            holder = type("_G", (), {})()  # update_fn reads .model / .group_sh of its owner: one holder per group
            holder.model = owner.model
            grp, tabs = _build_group(holder, emb, ns, group, capacity_factor, calibration, dedup)
            holder.group_sh = grp
            self.groups[d] = (grp, ns)
            self.sharded_tables += tabs
        self.sharded_names = set(names)
        self.replicated = [n for n in emb.feature_table if n not in self.sharded_names]
        self._orig_gather_into = EmbeddingsBlock.gather_into
        self._orig_gather_concat = EmbeddingsBlock.gather_concat
        emb.gather_into = self.gather_into
        emb.gather_concat = self.gather_concat
        emb._apply_sparse_now = self.apply_sparse_now
        self._active: List[ShardedEmbeddingGroup] = []
        self._aliases: Dict[str, ShardedEmbeddingGroup] = {}  # list feature -> its own route over the group's shards
        self._list_ctx: Dict[str, tuple] = {}

    # ---- list / ragged features of a row-sharded table (SOK's lookup takes sparse ids with a combiner,
    #      tf/distributed/embedding.py:144-148): one request per VALUE through an alias route, combined on the requesting rank
    def _list_lookup(self, grp, gnames, n, x, out_view) -> None:
        from . import ops
        from .inputs import Ragged

        ft = self.emb.feature_table[n]
        comb = ft.sequence_combiner
        if not comb:
            raise ValueError("list inputs need a str sequence_combiner")
        if isinstance(x, Ragged):
            vals, offs, shape = x.values.reshape(-1), x.offsets, None
            # pruned ids (< 0, safe_embedding_lookup_sparse) are requested as row `input_dim`: beyond the table, so the owner
            # answers with a zero row and its update skips the request -- exactly "not looked up"
            req = torch.where(vals >= 0, vals, torch.full_like(vals, ft.input_dim))
        else:
            if x.dim() == 3 and x.shape[-1] == 1:
                x = x.squeeze(-1)
            if comb == "sqrtn":
                raise ValueError("Only 'mean', 'sum', and 'max' str combiners is implemented for dense list/multi-hot embedded features.")
            vals, offs, shape = x.reshape(-1), None, tuple(x.shape)
            req = vals
        alias = self._aliases.get(n)
        if alias is None:
            alias = self._aliases[n] = grp.alias()
        rows = alias.lookup([req], features=[gnames.index(n)])[0]            # [nnz, D] in value order
        local = torch.arange(vals.numel(), dtype=vals.dtype, device=vals.device)
        if offs is not None:
            local = torch.where(vals >= 0, local, torch.full_like(local, -1))
            ops.embedding_bag(rows, local, offs, comb, out=out_view)
        else:
            ops.embedding_dense_list(rows, local.reshape(shape), comb, out=out_view)
        self._list_ctx[n] = (alias, rows, local if offs is not None else local.reshape(shape), offs)

    def _list_backward(self, n, gcols, comb) -> None:
        from . import ops

        alias, rows, local, offs = self._list_ctx.pop(n)
        _, gexp = ops.embedding_bag_expand(rows, local, offs, gcols, comb)   # one gradient row per value
        alias.backward_begin(gexp.reshape(1, gexp.shape[0], gexp.shape[1]))
        self._active.insert(0, alias)  # aliases hand their rows over BEFORE the owner groups apply

    def gather_into(self, inputs, out, slots) -> None:
        from . import ops

        emb = self.emb
        names = [n for n in emb.feature_names if n in inputs]
        D = out.shape[2]
        grp, gnames = self.groups.get(D, (None, []))
        sharded_here = [n for n in gnames if n in names]
        mine = [n for n in sharded_here if emb._is_onehot(inputs[n])]  # the GROUP's feature order (keys carry the feature index)
        lists = [n for n in sharded_here if n not in mine]
        if mine:
            def scatter_fn(tabs, idx, o, sl):
                ops.embedding_gather(tabs, idx, out=o, out_slot=sl)

            grp.lookup_begin([inputs[n].reshape(-1) for n in mine], scatter_into=(out, [slots[n] for n in mine], scatter_fn),
                             features=[gnames.index(n) for n in mine])
        rest = {n: inputs[n] for n in names if n not in sharded_here}
        if rest:
            self._orig_gather_into(emb, rest, out, slots)  # the replicated-table gather overlaps the row all-to-all
        if mine:
            grp.lookup_end()
        for n in lists:
            self._list_lookup(grp, gnames, n, inputs[n], out[:, slots[n]])
        last = dict(getattr(emb, "_last_all", {})) if getattr(emb, "_last_step", None) is self.owner._step_token else {}
        last.update({n: inputs[n] for n in names})
        emb._last_all, emb._last_step = last, self.owner._step_token
        emb._last = last
        emb._record_fwd_out(names, lambda n: out[:, slots[n]])

    def gather_concat(self, inputs, names, buf, offsets) -> None:
        """Concat layout (InputBlockV2): sharded features through their groups, the rest by the local gather."""
        from . import ops

        emb = self.emb
        begun = []
        for d, (grp, gnames) in self.groups.items():
            mine = [n for n in gnames if n in names]  # the GROUP's feature order (gather_concat takes one-hot features only)
            if mine:
                grp.lookup_begin([inputs[n].reshape(-1) for n in mine], features=[gnames.index(n) for n in mine])
                begun.append((grp, mine))
        taken = {n for _, mine in begun for n in mine}
        rest = [n for n in names if n not in taken]
        if rest:
            self._orig_gather_concat(emb, inputs, rest, buf, offsets)
        for grp, mine in begun:
            grp.lookup_end(scatter=lambda back, pos, mine=mine: ops.embedding_gather(
                [back] * len(mine), pos, out=buf, out_offset=[offsets[n] for n in mine]))
        emb._last = {n: inputs[n] for n in names}
        emb._record_fwd_out(names, lambda n: buf[:, offsets[n]:offsets[n] + emb.feature_table[n].dim])

    def apply_sparse_now(self, opt, grad, offsets, reset_reg: bool = True) -> None:
        from . import ops

        emb = self.emb
        g2 = grad.reshape(grad.shape[0], -1)
        emb._apply_batch_regularization(g2, offsets, reset=reset_reg)
        B, stride = g2.shape
        present = [n for n in offsets if n in emb._last and emb.feature_table[n].table.trainable]
        for d, (grp, gnames) in self.groups.items():
            here = [n for n in gnames if n in present]
            if not here:
                continue
            if opt.name == "adagrad" and grp.state is None:
                grp.state = torch.full_like(grp.local, opt.initial_accumulator_value)
            if opt.name == "adam" and grp.state is None:
                grp.state, grp.state2 = torch.zeros_like(grp.local), torch.zeros_like(grp.local)
            mine = [n for n in here if n not in self._list_ctx]  # one-hot lookups, same order as the lookup
            for n in here:
                if n in self._list_ctx:
                    self._list_backward(n, g2[:, offsets[n]:offsets[n] + d], emb.feature_table[n].sequence_combiner)
            if grp not in self._active:
                self._active.append(grp)  # applies what its aliases hand over even when it has no one-hot lookups itself
            if not mine:
                continue
            pull = lambda tab, idx: ops.embedding_gather([tab], [idx])[:, 0]
            if stride % d == 0 and all(offsets[n] % d == 0 for n in mine):
                # gradient rows sit at multiples of d: the buffer IS a [B, stride / d, d] stack, rows pulled in owner order
                grp.backward_begin(None, from_stacked=(g2.reshape(B, stride // d, d), [offsets[n] // d for n in mine], pull))
            else:  # e.g. a concat row of 26 x 128 + 13 floats: compact the sharded features' columns first (one copy)
                compact = torch.stack([g2[:, offsets[n]:offsets[n] + d] for n in mine], dim=1).contiguous()
                grp.backward_begin(None, from_stacked=(compact, list(range(len(mine))), pull))
        rep = [n for n in present if n not in self.sharded_names]
        onehot = [n for n in rep if emb._is_onehot(emb._last[n])]
        grad_of = self.owner._rep_grad
        for d in sorted({emb.feature_table[n].dim for n in onehot}):
            grp_n = [n for n in onehot if emb.feature_table[n].dim == d]
            ops.embedding_gather_backward([grad_of[id(emb.feature_table[n].table)] for n in grp_n], None,
                                          [emb._last[n] for n in grp_n], grad, [offsets[n] for n in grp_n], "sgd", -1.0, 0.0)
        for n in rep:
            if n in onehot:
                continue
            from .inputs import Ragged

            ft, x = emb.feature_table[n], emb._last[n]
            vals, offs = (x.values, x.offsets) if isinstance(x, Ragged) else (x, None)
            ops.embedding_bag_backward(grad_of[id(ft.table)], None, vals, offs, g2[:, offsets[n]:offsets[n] + ft.dim],
                                       ft.sequence_combiner, "sgd", -1.0, 0.0)

    def finish(self) -> None:
        for grp in self._active:
            grp.backward_end()
        self._active = []


class DistributedModel:
    """Data-parallel execution of ANY ``mm`` ranking / retrieval model whose categorical inputs go through
    EmbeddingsBlocks -- ``mm.TwoTowerModel`` (BASELINE configs[2]) and ``mm.DCNModel`` (configs[4]) in the bench --
    the role of ``hvd.DistributedOptimizer`` around the model's optimizer in the reference (tf/models/base.py:476-508,
    1472-1476):
      * the batch is sharded by rank; in-batch negatives stay rank-local (tf/blocks/retrieval/base.py:329-375);
      * tables with >= ``shard_threshold`` rows are row-sharded (``ShardedEmbeddingGroup``: all-to-all lookup, fused
        update on the owner; never all-reduced), the rest are replicated;
      * every dense gradient (MLP / cross / head tensors and the dense [V, D] gradients of the replicated tables) lives
        in ONE flat bucket, summed in place by reduce-scatter + all-gather while the owners apply the sharded updates;
      * the loss gradient is scaled by 1/W at the source, so every reduction is a plain SUM; parameters are broadcast
        from rank 0 at construction.
    Replicated tables take the DENSE optimizer step on their summed gradient (Adam: every row's moments decay; the
    single-GPU path steps touched rows only, LazyAdam) -- identical for SGD / Adagrad."""

    def __init__(self, model, shard_threshold: int = 200_000, group=None, force_shard: bool = False,
                 capacity_factor: Optional[float] = None, calibration: int = 2, dedup="auto"):
        from .inputs import EmbeddingsBlock

        self.model = model
        self.group = group
        self.rank, self.world_size = world()
        W = self.world_size
        active = W > 1 or force_shard
        self._step_token = object()
        self.shards: List[_ShardedEmbeddings] = []
        rep_tabs, seen, shard_ids = [], set(), set()
        for emb in model.blocks_of_type(EmbeddingsBlock):
            names = [n for n, t in emb.feature_table.items()
                     if active and (t.input_dim >= shard_threshold or getattr(t, "shard", None) is not None)]
            sh = _ShardedEmbeddings(self, emb, names, group, capacity_factor, 0 if W == 1 else calibration, dedup)
            self.shards.append(sh)
            shard_ids |= {id(t.table) for t in sh.sharded_tables}
        for emb in model.blocks_of_type(EmbeddingsBlock):
            for t in emb.feature_table.values():
                if id(t.table) not in shard_ids and id(t.table) not in seen and t.table.trainable:
                    seen.add(id(t.table))
                    rep_tabs.append(t.table)
        self.rep_tabs = rep_tabs
        self._shard_param_ids = shard_ids
        self._rep_grad: Dict[int, torch.Tensor] = {}
        self._bucket: Optional[torch.Tensor] = None
        model.loss_grad_divisor = W
        if model.optimizer is None:
            model.compile()
        model.optimizer.apply = self._apply  # the DistributedOptimizer: same call site, reductions inside
        dense = [p.data for p in model.parameters() if not p.sparse]
        broadcast_parameters(dense + [t.data for t in rep_tabs], 0, group)

    @property
    def graph_capturable(self) -> bool:
        import os

        ok = all(g.graph_capturable for sh in self.shards for g, _ in sh.groups.values())
        return ok and getattr(self.model, "graph_capturable", True) and (
            self.world_size == 1 or os.environ.get("MERLIN_HIP_GRAPH_DISTRIBUTED") == "1")

    def check_overflow(self) -> None:
        for sh in self.shards:
            for g, _ in sh.groups.values():
                g.check_overflow()

    def __call__(self, inputs, **kwargs):
        self._step_token = object()
        return self.model(inputs, **kwargs)

    def _apply(self, model=None) -> None:
        from . import ops
        from .inputs import EmbeddingsBlock

        model, opt = self.model, self.model.optimizer
        params = model.parameters()
        if params:
            opt.ensure_begun(params[0].data.device)
        dense = [q for q in params if not q.sparse and q.trainable and q.grad is not None]
        n_dense = sum(q.grad.numel() for q in dense)
        n_rep = sum(t.data.numel() for t in self.rep_tabs)
        n_head = (n_dense + 63) // 64 * 64  # table gradients start 256-byte aligned
        W = self.world_size
        total = (n_head + n_rep + 64 * W - 1) // (64 * W) * (64 * W)
        dev = params[0].data.device
        if self._bucket is None or self._bucket.numel() != total:
            from . import ops as _ops

            _ops.park_replaced(self._bucket)
            self._bucket = torch.zeros(total, dtype=torch.float32, device=dev)
        bucket = self._bucket
        if bucket.is_cuda:
            from . import ops as _ops2

            _ops2.note_captured(bucket)
        o = n_head
        for t in self.rep_tabs:
            self._rep_grad[id(t)] = bucket[o:o + t.data.numel()].view_as(t.data)
            o += t.data.numel()
        with ops.SIDE.deferred():
            if self.rep_tabs:
                bucket[n_head:n_head + n_rep].zero_()
            for emb in model.blocks_of_type(EmbeddingsBlock):
                pending = getattr(emb, "_pending", None)
                if pending is None:
                    continue
                emb._pending, emb._pending_event = None, None
                emb._apply_sparse_now(opt, *pending)  # sharded: gradient all-to-all started; replicated: into the bucket
            ops.SIDE.join()  # dW / db GEMMs ran on their side stream: the bucket reads them
        ops.run_tail()  # parked chain slab reductions: the bucket reads their dW / db too (and the BCE sum)
        if dense:
            torch.cat([q.grad.reshape(-1) for q in dense], out=bucket[:n_dense])
        works = allreduce_flat_(bucket, self.group, async_op=True)
        for sh in self.shards:
            sh.finish()  # fused updates of the local shards overlap the bucket reduction
        for w in works:
            w.wait()
        o = 0
        for q in dense:
            q.grad = bucket[o:o + q.data.numel()].view_as(q.data)
            o += q.data.numel()
        for t in self.rep_tabs:
            t.grad = self._rep_grad[id(t)]
        ops.dense_optimizer_step_multi(opt, dense + self.rep_tabs)
        opt._begun = False

    def train_step(self, inputs, targets=None):
        self._step_token = object()
        loss = self.model.train_step(inputs, targets)
        if self.world_size > 1:  # mean of the ranks' batch-mean losses = the global-batch mean (equal shards)
            loss = loss.detach().clone()
            dist.all_reduce(loss, group=self.group)
            loss = loss / self.world_size
        return loss

    # checkpoint / resume: replicated part by rank 0, row shards + their optimizer state per rank
    def save_weights(self, path) -> None:
        import numpy as np

        path = str(path)
        if self.rank == 0:
            self.model.save_weights(path, skip=self._shard_param_ids)
        arrays = {}
        for i, sh in enumerate(self.shards):
            for d, (g, _) in sh.groups.items():
                arrays[f"local_{i}_{d}"] = g.local.detach().cpu().numpy()
                if g.state is not None:
                    arrays[f"state_{i}_{d}"] = g.state.detach().cpu().numpy()
                if g.state2 is not None:
                    arrays[f"state2_{i}_{d}"] = g.state2.detach().cpu().numpy()
        np.savez(f"{path}.shard{self.rank}of{self.world_size}.npz", **arrays)
        if self.world_size > 1:
            dist.barrier(group=self.group)

    def load_weights(self, path) -> None:
        import numpy as np

        path = str(path)
        self.model.load_weights(path, skip=self._shard_param_ids)
        z = np.load(f"{path}.shard{self.rank}of{self.world_size}.npz")
        for i, sh in enumerate(self.shards):
            for d, (g, _) in sh.groups.items():
                g.local.copy_(torch.from_numpy(z[f"local_{i}_{d}"]).to(g.local.device))
                for key in ("state", "state2"):
                    name = f"{key}_{i}_{d}"
                    if name in z.files:
                        val = torch.from_numpy(z[name]).to(g.local.device)
                        cur = getattr(g, key)
                        if cur is None:
                            setattr(g, key, val)
                        else:
                            cur.copy_(val)


DistributedTwoTower = DistributedModel  # BASELINE configs[2]: row-sharded user / item id tables, rank-local negatives
DataParallel = DistributedModel         # BASELINE configs[4]: DCN-v2, replicated cross / deep weights in one bucket
