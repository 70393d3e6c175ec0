"""Negative samplers of the retrieval path and the logQ sampling correction.

Reference: ``merlin/models/tf/outputs/sampling/base.py`` (Candidate, CandidateSampler),
``outputs/sampling/in_batch.py:25-114`` (InBatchSamplerV2), ``outputs/sampling/popularity.py:24-199``
(PopularityBasedSamplerV2: log-uniform candidate sampler + its sampling probabilities),
``blocks/sampling/queue.py:22-360`` (FIFOQueue: fixed-capacity ring storage for cross-batch negatives) and
``transforms/bias.py:77-290`` (PopularityLogitsCorrection).  The samplers only produce ids / embeddings / log
sampling probabilities; scoring, the correction itself and the loss run in the fused scorer kernels
(``ops.inbatch_softmax*`` with ``pos_logq`` / ``neg_logq``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch

from .core import Block
from .schema import Schema, Tags

EMBEDDING_KEY = "__embedding__"
LOGQ_EPS = 1e-16  # outputs/contrastive.py:316, transforms/bias.py:252


class Candidate:
    """Candidate ids with their metadata (outputs/sampling/base.py:26-100)."""

    def __init__(self, id: torch.Tensor, metadata: Optional[Dict[str, torch.Tensor]] = None,
                 sampling_prob: Optional[torch.Tensor] = None):
        self.id = id
        self.metadata = {} if metadata is None else metadata
        self.sampling_prob = sampling_prob

    @property
    def embedding(self) -> torch.Tensor:
        return self.metadata[EMBEDDING_KEY]

    @property
    def has_embedding(self) -> bool:
        return EMBEDDING_KEY in self.metadata

    def with_embedding(self, embedding: torch.Tensor) -> "Candidate":
        self.metadata[EMBEDDING_KEY] = embedding
        return self

    def with_sampling_prob(self, sampling_prob: torch.Tensor) -> "Candidate":
        return Candidate(self.id, self.metadata, sampling_prob)

    def __add__(self, other: "Candidate") -> "Candidate":
        md = {k: torch.cat([v, other.metadata[k]], 0) for k, v in self.metadata.items() if k in other.metadata}
        prob = None
        if self.sampling_prob is not None and other.sampling_prob is not None:
            prob = torch.cat([self.sampling_prob.reshape(-1), other.sampling_prob.reshape(-1)], 0)
        return Candidate(torch.cat([self.id.reshape(-1), other.id.reshape(-1)], 0), md, prob)


class CandidateSampler(Block):
    """Base class of the negative samplers (outputs/sampling/base.py:103-190)."""

    def __init__(self, max_num_samples: Optional[int] = None, name: Optional[str] = None):
        super().__init__(name)
        self.set_max_num_samples(max_num_samples)

    def set_max_num_samples(self, value) -> None:
        self._max_num_samples = value

    @property
    def max_num_samples(self) -> int:
        return self._max_num_samples

    def forward(self, items: Candidate, features=None, targets=None, training: bool = False, testing: bool = False) -> Candidate:
        if training:
            self.add(items)
        return self.sample()

    def with_sampling_probs(self, items: Candidate) -> Candidate:
        return items

    @property
    def graph_capturable(self) -> bool:
        """True when a step using this sampler is a fixed launch sequence with fixed shapes (no host-side state that a
        replayed hipGraph would miss)."""
        return False

    def add(self, items: Candidate) -> None:
        raise NotImplementedError()

    def sample(self) -> Candidate:
        raise NotImplementedError()


class InBatchSamplerV2(CandidateSampler):
    """The batch's own items are the negatives (outputs/sampling/in_batch.py:25-114)."""

    def __init__(self, batch_size: Optional[int] = None, name: Optional[str] = None):
        super().__init__(batch_size, name)
        self._last_batch: Optional[Candidate] = None

    graph_capturable = True

    def add(self, items: Candidate) -> None:
        self._last_batch = items

    def forward(self, items: Candidate, features=None, targets=None, training: bool = False, testing: bool = False) -> Candidate:
        self.add(items)
        return self.sample()

    def sample(self) -> Candidate:
        return self._last_batch


class PopularityBasedSamplerV2(CandidateSampler):
    """Log-uniform (Zipfian) sampling of ``max_num_samples`` ids from ``[min_id, max_id)`` -- ids are assumed to be
    sorted by decreasing frequency -- with the sampling probabilities the logQ correction needs
    (outputs/sampling/popularity.py:24-199; the sampler is TensorFlow's ``log_uniform_candidate_sampler``:
    P(k) = (log(k + 2) - log(k + 1)) / log(range_max + 1), drawn as floor(exp(u log(range_max + 1))) - 1).

    The draw runs on the device (``mh_log_uniform_sample``: counter-based Philox draws, first-appearance dedup in a hash
    set; the call counter is device state): no host loop, no host synchronisation, replayable from a hipGraph."""

    def __init__(self, max_id: int, min_id: int = 0, max_num_samples: int = 10, unique: bool = True,
                 seed: Optional[int] = None, name: Optional[str] = None):
        super().__init__(max_num_samples, name)
        self.max_id, self.min_id, self.seed, self.unique = int(max_id), int(min_id), seed, bool(unique)
        assert self.max_num_samples <= self.max_id, (
            f"Number of items to sample `{self.max_num_samples}` should be less than total number of ids `{self.max_id}`")
        if self.unique and self.max_num_samples > self.max_id - self.min_id:
            raise ValueError(f"cannot draw {self.max_num_samples} unique ids from the {self.max_id - self.min_id} ids of "
                             f"[{self.min_id}, {self.max_id})")
        if seed is None:  # like an unseeded TF op: a fresh stream per sampler
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._seed = int(seed)
        self._rng_state: Dict[torch.device, torch.Tensor] = {}  # device int64 [2] = (seed, calls so far)
        self.sampling_dist = self.get_sampling_distribution()

    graph_capturable = True  # the draw is one kernel, its call counter is device state

    def add(self, items: Candidate) -> None:
        pass

    def forward(self, positive_items: Candidate = None, features=None, targets=None, training: bool = False,
                testing: bool = False) -> Candidate:
        return self.sample(device=None if positive_items is None else positive_items.id.device)

    def sample(self, device=None) -> Candidate:
        from . import ops
        from .inputs import default_device

        device = torch.device(device) if device is not None else default_device()
        if device.type != "cuda":
            raise RuntimeError("PopularityBasedSamplerV2 draws on the GPU (mh_log_uniform_sample); there is no CPU path")
        st = self._rng_state.get(device)
        if st is None:
            st = self._rng_state[device] = torch.tensor([self._seed, 0], dtype=torch.int64, device=device)
        ids = ops.log_uniform_sample(self.max_id - self.min_id, self.max_num_samples, self.unique, st, self.min_id)
        return Candidate(ids.reshape(-1, 1), {})

    def check_status(self) -> None:
        """Host read of the sampler kernel's status word on every device it drew on (the kernel cannot raise, and a captured step
        cannot read it): call at epoch ends / after evaluation (``Model.fit`` does)."""
        from . import ops

        for dev in self._rng_state:
            if ops.log_uniform_sample_status(dev) != 0:
                raise RuntimeError(f"PopularityBasedSamplerV2: the unique draw of {self.max_num_samples} ids from "
                                   f"[{self.min_id}, {self.max_id}) did not complete on {dev}")

    def get_sampling_distribution(self) -> torch.Tensor:
        """Probability of every id under the sampler (popularity.py:139-166); with ``unique`` the probability of being
        drawn at least once in ``max_num_samples`` trials, 1 - (1 - p)^n computed as -expm1(n log1p(-p))."""
        log_idx = torch.log(torch.arange(1.0, self.max_id - self.min_id + 3.0, 1.0, dtype=torch.float32))
        p = (log_idx[1:] - log_idx[:-1]) / log_idx[-1]
        if self.unique:
            p = -torch.expm1(self.max_num_samples * torch.log1p(-p))
        if self.min_id > 0:
            p = torch.cat([torch.zeros(self.min_id, dtype=p.dtype), p], 0)
        return p

    def with_sampling_probs(self, items: Candidate) -> Candidate:
        dist = self.sampling_dist
        if dist.device != items.id.device:
            dist = self.sampling_dist = dist.to(items.id.device)
        return items.with_sampling_prob(dist[items.id.reshape(-1).long()])


class FIFOQueue(Block):
    """Fixed-capacity first-in-first-out store of tensors across batches (the role of blocks/sampling/queue.py:22-360:
    cached item embeddings / ids for cross-batch negatives; when full, the oldest entries are overwritten).

    Design: ``storage`` is a device-resident ring.  The slot of the oldest entry lives ON THE DEVICE (``_head``, an int64
    scalar tensor) and every data movement is ONE indexed copy through slot numbers computed on the device,
    ``(head + offset + arange(n)) % capacity`` -- wrap-around is arithmetic, not a case distinction.  The host keeps only the
    entry COUNT, which moves by amounts it knows (the row count of what is enqueued / dequeued): no operation reads anything
    back from the device, and once the queue is full every operation is a fixed launch sequence with fixed shapes -- a train
    step that feeds the queue replays from a hipGraph (the head advances on the device with every replay)."""

    def __init__(self, capacity: int, dtype: torch.dtype, dims: Sequence[int] = (), queue_name: str = "",
                 initialize_tensor: Optional[torch.Tensor] = None, device=None, name: Optional[str] = None):
        assert capacity > 0
        super().__init__(name)
        self.capacity, self.queue_dtype, self.dims, self.queue_name = int(capacity), dtype, list(dims), queue_name
        if initialize_tensor is None:
            # -1 is never a valid categorical value: index_of() cannot match a slot that was never written
            initialize_tensor = torch.full([self.capacity] + self.dims, -1, dtype=dtype, device=device)
        self.storage = initialize_tensor.clone()
        dev = self.storage.device
        self._head = torch.zeros((), dtype=torch.int64, device=dev)   # slot of the oldest entry (device state)
        self._size = 0                                                # entries queued (host: moves by host-known amounts)
        self._lane = torch.arange(self.capacity, device=dev)          # 0 .. capacity - 1, reused by every slot range

    def _slots(self, offset: int, n: int) -> torch.Tensor:
        """Ring slots of the entries offset, offset + 1, ... (n of them) counted from the oldest, wrapped."""
        return (self._lane[:n] + (self._head + offset)) % self.capacity

    def _check_rows(self, values: torch.Tensor) -> None:
        assert values.dim() == len(self.dims) + 1, (
            "The rank of values (ignoring the first dim which is the number of examples) and self.dims should match")
        assert list(values.shape[1:]) == self.dims, (
            "The shape of values (ignoring the first dim which is the number of examples) and self.dims should match")

    def _push(self, rows: torch.Tensor) -> None:
        n = int(rows.shape[0])
        if n > self.capacity:  # only the newest `capacity` rows can survive
            rows, n = rows[n - self.capacity:], self.capacity
        self.storage.index_copy_(0, self._slots(self._size, n), rows.to(self.storage.dtype))
        lost = max(0, self._size + n - self.capacity)  # overwritten entries were the oldest: the head moves past them
        if lost:
            self._head.add_(lost).remainder_(self.capacity)
        self._size = min(self.capacity, self._size + n)

    def _pop(self, n: int) -> torch.Tensor:
        rows = self.storage.index_select(0, self._slots(0, n))
        self._head.add_(n).remainder_(self.capacity)
        self._size -= n
        return rows

    def enqueue(self, val: torch.Tensor) -> None:
        assert val.dim() == len(self.dims), "The rank of val and self.dims should match"
        assert list(val.shape) == self.dims, "The shape of val and self.dims should match"
        self._push(val.unsqueeze(0))

    def enqueue_many(self, vals: torch.Tensor) -> None:
        self._check_rows(vals)
        self._push(vals)

    def dequeue(self) -> torch.Tensor:
        if self._size == 0:
            raise IndexError("The queue is empty")
        return self._pop(1)[0]

    def dequeue_many(self, n: int) -> torch.Tensor:
        if self._size == 0:
            raise IndexError("The queue is empty")
        if n <= 0:
            raise ValueError("The number of elements to dequeue must be greater than 0.")
        return self._pop(min(int(n), self._size))

    def list_all(self) -> torch.Tensor:
        """Every queued entry, oldest first."""
        return self.storage.index_select(0, self._slots(0, self._size))

    def count(self) -> int:
        return self._size

    @property
    def at_full_capacity(self) -> bool:
        return self._size == self.capacity

    def clear(self) -> None:
        self._head.zero_()
        self._size = 0

    def index_of(self, ids: torch.Tensor) -> torch.Tensor:
        """Slot in the STORAGE of every id (first match), -1 if absent; integer queues of scalars only."""
        assert not self.queue_dtype.is_floating_point, "The index_of method is only available for queues with an int dtype"
        assert self.dims == [], "The index_of method is only available for queues of scalars (dims=[])"
        hit = self.storage.reshape(1, -1) == ids.reshape(-1, 1).to(self.storage.dtype)
        slot = torch.where(hit, self._lane.reshape(1, -1), self.capacity).amin(dim=1)  # smallest matching slot
        return torch.where(slot < self.capacity, slot, -1)

    def get_values_by_indices(self, indices: torch.Tensor) -> torch.Tensor:
        return self.storage.index_select(0, indices.reshape(-1).long())

    def update_by_indices(self, indices: torch.Tensor, values: torch.Tensor) -> None:
        self._check_rows(values)
        self.storage.index_copy_(0, indices.reshape(-1).long(), values.to(self.storage.dtype))


class CachedCrossBatchSampler(CandidateSampler):
    """Cross-batch negatives: item embeddings (and ids) of the PREVIOUS batches, cached in FIFO queues, are appended to
    the negatives of the current batch -- the sampler blocks/sampling/queue.py:28-29 and in_batch.py:30 name.  Cached
    embeddings are constants (no gradient flows into them; they were produced by older weights).
    ``ignore_last_batch_on_sample``: the current batch enters the queue only AFTER it was scored, so that it is not
    sampled twice when combined with the in-batch sampler."""

    def __init__(self, capacity: int, ignore_last_batch_on_sample: bool = True, name: Optional[str] = None):
        super().__init__(capacity, name)
        self.capacity = int(capacity)
        self.ignore_last_batch_on_sample = ignore_last_batch_on_sample
        self._ids: Optional[FIFOQueue] = None
        self._emb: Optional[FIFOQueue] = None
        self._pending: Optional[Candidate] = None

    @property
    def graph_capturable(self) -> bool:
        """Once the queues are full (and a batch is pending) every step enqueues and lists the same number of rows: fixed
        shapes, the ring head advances on the device."""
        return (self._emb is not None and self._emb.at_full_capacity and self._ids.at_full_capacity
                and (self._pending is not None or not self.ignore_last_batch_on_sample))

    def _ensure(self, items: Candidate) -> None:
        if self._emb is None:
            dev = items.embedding.device
            self._ids = FIFOQueue(self.capacity, torch.int64, [], "item_ids", device=dev)
            self._emb = FIFOQueue(self.capacity, torch.float32, [items.embedding.shape[1]], "item_embeddings", device=dev)

    def add(self, items: Candidate) -> None:
        self._ensure(items)
        self._ids.enqueue_many(items.id.reshape(-1).to(torch.int64))
        self._emb.enqueue_many(items.embedding.detach())

    def forward(self, items: Candidate, features=None, targets=None, training: bool = False, testing: bool = False) -> Candidate:
        self._ensure(items)
        if self._pending is not None:  # the previous batch becomes available now
            self.add(self._pending)
            self._pending = None
        if training:
            if self.ignore_last_batch_on_sample:
                # a snapshot (the caller's buffers are reused by the next step, no gradient flows into the cache) written IN
                # PLACE into persistent buffers: the batch pending from one step is consumed by the next, and a step replayed
                # from a hipGraph can only hand a tensor to its next replay through an address that does not change
                hold = getattr(self, "_hold", None)
                if hold is None or hold.id.shape != items.id.shape or hold.embedding.shape != items.embedding.shape:
                    if hold is not None:  # a captured step may still address the old buffers
                        from . import ops

                        ops.park_replaced(hold.id)
                        ops.park_replaced(hold.embedding)
                    hold = Candidate(torch.empty_like(items.id), {EMBEDDING_KEY: torch.empty_like(items.embedding.detach())})
                if hold.id.is_cuda:
                    from . import ops as _ops

                    _ops.note_captured(hold.id)
                    _ops.note_captured(hold.embedding)
                hold.id.copy_(items.id.detach())
                hold.embedding.copy_(items.embedding.detach())
                self._hold = self._pending = hold
            else:
                self.add(items)
        return self.sample()

    def sample(self) -> Candidate:
        return Candidate(self._ids.list_all().reshape(-1, 1), {EMBEDDING_KEY: self._emb.list_all()})


def parse_negative_samplers(negative_samplers) -> List[CandidateSampler]:
    """str | sampler | sequence -> list of samplers (outputs/sampling/base.py: parse_negative_samplers)."""
    if negative_samplers is None:
        negative_samplers = ["in-batch"]
    if not isinstance(negative_samplers, (list, tuple)):
        negative_samplers = [negative_samplers]
    out = []
    for s in negative_samplers:
        if isinstance(s, str):
            if s not in ("in-batch", "in_batch"):
                raise ValueError(f"unknown negative sampler {s!r} (registered: 'in-batch')")
            s = InBatchSamplerV2()
        if not isinstance(s, CandidateSampler):
            raise TypeError(f"negative sampler must be a str or a CandidateSampler, got {type(s).__name__}")
        out.append(s)
    return out


class PopularityLogitsCorrection(Block):
    """``logits -= reg_factor * log(item_prob + 1e-16)`` for the positive and every negative candidate
    (transforms/bias.py:77-290), ``item_prob`` = normalised item frequencies.  Used as ``post=`` of ContrastiveOutput:
    the correction then acts on every column AFTER the false-negative rescoring."""

    def __init__(self, item_freq_probs: Union[torch.Tensor, Sequence, None] = None, is_prob_distribution: bool = False,
                 reg_factor: float = 1.0, schema: Optional[Schema] = None, candidate_tag_id=Tags.ITEM_ID,
                 name: Optional[str] = None):
        super().__init__(name)
        self.reg_factor = float(reg_factor)
        self.schema = schema
        self.candidate_id_name = None
        self.cardinality = None
        if schema is not None:
            col = schema.select_by_tag(candidate_tag_id).first
            self.candidate_id_name = col.name
            self.cardinality = int(col.int_domain.max) + 1
        self.candidate_probs: Optional[torch.Tensor] = None
        if item_freq_probs is not None:
            self.update(item_freq_probs, is_prob_distribution)

    def update(self, item_freq_probs, is_prob_distribution: bool = False) -> None:
        t = torch.as_tensor(item_freq_probs)
        if self.cardinality is not None and t.shape[0] != self.cardinality:
            raise ValueError("The item frequency table length does not match the item ids cardinality"
                             f"(expected {self.cardinality}, got {t.shape[0]})")
        if is_prob_distribution:
            if t.dtype != torch.float32:
                raise TypeError("The item_weights should have tf.float32 dtype")
            p = t
        else:  # utils/tf_utils.py:349-390: frequencies -> probabilities
            t = t.to(torch.float32)
            p = t / t.sum()
        self.candidate_probs = p.reshape(-1)
        self._log_probs = None

    def log_probs(self, device) -> torch.Tensor:
        """reg_factor * log(p + 1e-16) per item id, cached on the device."""
        if getattr(self, "_log_probs", None) is None or self._log_probs.device != device:
            self._log_probs = (self.reg_factor * torch.log(self.candidate_probs.to(torch.float32) + LOGQ_EPS)).to(device)
        return self._log_probs

    def logq(self, ids: torch.Tensor) -> torch.Tensor:
        return self.log_probs(ids.device)[ids.reshape(-1).long()]
