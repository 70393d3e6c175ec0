"""Input blocks of the kept ``mm`` surface (reference layer L2, merlin/models/tf/inputs/).

``EmbeddingTable`` (embedding.py:153-471), ``Embeddings`` (:585-683), ``ContinuousFeatures``
(continuous.py:73-138) and ``InputBlockV2`` (base.py:216-341).  The per-table Keras
``Embedding`` layers of the reference become ONE multi-table HIP gather launch per batch
(``mh_embedding_gather_fwd``) that writes straight into a stacked ``[B, F, D]`` buffer.
"""
from __future__ import annotations

from typing import Callable, Dict, List, NamedTuple, Optional, Sequence, Union

import numpy as np
import torch

from . import ops
from .core import Block, ParallelBlock, Parameter, TabularData, parse_aggregation
from .schema import ColumnSchema, Schema, Tags, infer_embedding_dim as _infer_dim_from_cardinality


class Ragged(NamedTuple):
    """CSR list feature: ``values[nnz]`` + ``offsets[B+1]`` -- what PrepareFeatures builds from
    ``name__values`` / ``name__offsets`` (tf/transforms/features.py:324-379)."""

    values: torch.Tensor
    offsets: torch.Tensor


def default_device() -> torch.device:
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def infer_embedding_dim(col_schema: ColumnSchema, multiplier: float = 2.0, ensure_multiple_of_8: bool = True) -> int:
    """utils/schema_utils.py:198-207."""
    return _infer_dim_from_cardinality(int(col_schema.int_domain.max) + 1, multiplier, ensure_multiple_of_8)


# Row sharding at construction (models_amd.distributed.sharded_tables): (rank, world, threshold) or None
_SHARD_CTX = None
_INIT_CHUNK = 1 << 20  # rows per init chunk of a large table


def _init_table(initializer, rows: int, dim: int, device, seed: Optional[int], shard=None) -> torch.Tensor:
    """keras Embedding default "uniform" = U(-0.05, 0.05) (embedding.py:205); V1 EmbeddingFeatures
    default TruncatedNormal(0, 0.05) (:1051) is available as "truncated_normal"; an array / tensor / callable gives the
    values themselves (pretrained tables, embedding.py:283).

    ``shard = (rank, W)`` returns the rows ``rank, rank + W, ...`` of EXACTLY the table an unsharded build on the same device
    would hold, without materialising that table on the device:
      * up to 2^24 elements the table is drawn from one CPU generator (as ever) and sliced on the host;
      * larger tables are drawn on the device in chunks of 2^20 rows, each from its own generator seeded by (seed, chunk) --
        the values of row r do not depend on the partition -- whether sharded or not;
      * given values are sliced on the host."""
    rank, W = shard if shard is not None else (0, 1)
    base = 0 if seed is None else seed
    kind = "uniform" if initializer is None else initializer
    if isinstance(kind, str) and kind in ("uniform", "truncated_normal"):
        def draw(shape, gen, dev):
            if kind == "uniform":
                return (torch.rand(shape, generator=gen, device=dev) - 0.5) * 0.1
            t = torch.empty(shape, device=dev)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=0.05, a=-0.1, b=0.1, generator=gen)
            return t

        if rows * dim <= (1 << 24):
            g = torch.Generator(device="cpu")
            g.manual_seed(base)
            return draw((rows, dim), g, "cpu")[rank::W].contiguous().to(device)
        n_local = (rows - rank + W - 1) // W if rows > rank else 0
        out = torch.empty((n_local, dim), dtype=torch.float32, device=device)
        gd = torch.Generator(device=device)
        o = 0
        for c0 in range(0, rows, _INIT_CHUNK):
            c1 = min(rows, c0 + _INIT_CHUNK)
            gd.manual_seed((base * 1_000_003 + c0 // _INIT_CHUNK) & 0x7FFFFFFFFFFFFFFF)
            chunk = draw((c1 - c0, dim), gd, device)
            first = (rank - c0) % W  # first row of this chunk owned by `rank`
            part = chunk[first::W]
            out[o:o + part.shape[0]] = part
            o += part.shape[0]
        return out
    if isinstance(initializer, str):
        raise ValueError(f"unknown embeddings_initializer {initializer!r} ('uniform', 'truncated_normal', values or a callable)")
    if callable(initializer):
        initializer = initializer((rows, dim))
    w = torch.as_tensor(np.asarray(initializer.cpu() if isinstance(initializer, torch.Tensor) else initializer),
                        dtype=torch.float32)
    if tuple(w.shape) != (rows, dim):
        raise ValueError(f"initializer gave shape {tuple(w.shape)}, expected {(rows, dim)}")
    return w[rank::W].contiguous().to(device)


class EmbeddingTable(Block):
    """One embedding table shared by one or more categorical columns (embedding.py:153).

    ``input_dim = int_domain.max + 1`` (:91-93, :244-249).  Accepted inputs per feature:
    ``[B]`` / ``[B,1]`` ids (one-hot), ``[B,L]`` dense list (reduced by ``sequence_combiner``),
    or :class:`Ragged` CSR (``safe_embedding_lookup_sparse`` semantics, :432-441).
    """

    def __init__(self, dim: int, *col_schemas: ColumnSchema, sequence_combiner: Optional[str] = None,
                 embeddings_initializer="uniform", trainable: bool = True, name: Optional[str] = None,
                 device=None, seed: Optional[int] = None, l2_batch_regularization_factor: float = 0.0):
        if not col_schemas:
            raise ValueError("EmbeddingTable needs at least one ColumnSchema")
        first = col_schemas[0]
        super().__init__(name or (first.int_domain.name if first.int_domain and first.int_domain.name else first.name))
        self.dim = int(dim)
        if self.dim % 4 != 0:
            raise ValueError(f"embedding dim must be a multiple of 4 on the HIP path, got {dim}")
        self.features: Dict[str, ColumnSchema] = {}
        self.input_dim = None
        for col in col_schemas:
            self.add_feature(col)
        if sequence_combiner is not None and sequence_combiner not in ("mean", "sum", "sqrtn", "max"):
            raise ValueError("sequence_combiner must be 'mean', 'sum', 'sqrtn' (ragged) or 'max' (dense lists)")
        self.sequence_combiner = sequence_combiner
        # embedding.py:463-464: every lookup adds factor * sum(out^2) over the batch to the loss
        self.l2_batch_regularization_factor = float(l2_batch_regularization_factor or 0.0)
        self.device = torch.device(device) if device is not None else default_device()
        # under distributed.sharded_tables(): a large table is ALLOCATED as this rank's row shard (rows rank, rank + W, ...)
        self.shard = None
        if _SHARD_CTX is not None and self.input_dim >= _SHARD_CTX[2]:
            self.shard = (_SHARD_CTX[0], _SHARD_CTX[1])
        w = _init_table(embeddings_initializer, self.input_dim, self.dim, self.device, seed, shard=self.shard)
        self.table = Parameter(w, name=f"{self.name}/embeddings", trainable=trainable, sparse=True)

    # embedding.py:111-128
    def add_feature(self, col: ColumnSchema) -> None:
        if col.int_domain is None or col.int_domain.max is None:
            raise ValueError(f"`col_schema` {col.name!r} needs to have an int-domain")
        card = int(col.int_domain.max) + 1
        if self.input_dim is not None and card != self.input_dim:
            raise ValueError(
                f"`col_schema` {col.name!r} does not share the table domain ({card} != {self.input_dim})"
            )
        self.input_dim = card
        self.features[col.name] = col

    @classmethod
    def from_pretrained(cls, data, col_schema: Optional[ColumnSchema] = None, trainable: bool = True,
                        name: Optional[str] = None, **kwargs) -> "EmbeddingTable":
        """embedding.py:253-311: table initialised from a [rows, dim] array."""
        arr = np.asarray(data.cpu() if isinstance(data, torch.Tensor) else data, dtype=np.float32)
        rows, dim = arr.shape
        if col_schema is None:
            from .schema import categorical

            col_schema = categorical(name or "pretrained", rows)
        return cls(dim, col_schema, embeddings_initializer=arr, trainable=trainable, name=name, **kwargs)

    @property
    def schema(self) -> Schema:
        return Schema(self.features.values())

    def own_parameters(self):
        return [self.table]

    def _lookup(self, x, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        w = self.table.data
        if isinstance(x, Ragged):
            if not self.sequence_combiner:
                raise ValueError("Ragged inputs need a str sequence_combiner on the HIP path")
            return ops.embedding_bag(w, x.values, x.offsets, self.sequence_combiner, out=out)
        if x.dim() == 3 and x.shape[-1] == 1:
            x = x.squeeze(-1)
        if x.dim() == 2 and x.shape[1] > 1:
            if not self.sequence_combiner:
                raise ValueError("dense list inputs need a sequence_combiner ('mean', 'sum' or 'max')")
            if self.sequence_combiner == "sqrtn":
                raise ValueError("Only 'mean', 'sum', and 'max' str combiners is implemented for dense list/multi-hot embedded features.")
            return ops.embedding_dense_list(w, x, self.sequence_combiner, out=out)
        res = ops.embedding_gather([w], [x], out=None if out is None else out.unsqueeze(1))
        return res[:, 0]

    def forward(self, inputs):
        if isinstance(inputs, dict):
            self._last = {n: inputs[n] for n in self.features if n in inputs}
            return {n: self._lookup(v) for n, v in self._last.items()}
        self._last = inputs
        return self._lookup(inputs)


class EmbeddingsBlock(ParallelBlock):
    """ParallelBlock{table_name: EmbeddingTable} whose one-hot features are fetched by ONE
    multi-table gather launch (what ``Embeddings(...)`` returns, embedding.py:585-683)."""

    def __init__(self, tables: Dict[str, EmbeddingTable], schema: Schema, aggregation=None, name="embeddings"):
        super().__init__(tables, aggregation=None, name=name, schema=schema)
        self.agg = parse_aggregation(aggregation)
        self.feature_table: Dict[str, EmbeddingTable] = {}
        for t in tables.values():
            for fname in t.features:
                self.feature_table[fname] = t

    @property
    def feature_names(self) -> List[str]:
        return [c.name for c in self.schema]

    def _is_onehot(self, v) -> bool:
        return isinstance(v, torch.Tensor) and (v.dim() == 1 or v.shape[-1] == 1 and v.dim() == 2)

    def gather_into(self, inputs: TabularData, out: torch.Tensor, slots: Dict[str, int]) -> None:
        """Write feature ``n`` into ``out[:, slots[n], :]`` for every categorical feature."""
        names = [n for n in self.feature_names if n in inputs]
        onehot = [n for n in names if self._is_onehot(inputs[n])]
        if onehot:
            ops.embedding_gather([self.feature_table[n].table.data for n in onehot],
                                 [inputs[n] for n in onehot], out=out, out_slot=[slots[n] for n in onehot])
        for n in names:
            if n not in onehot:
                self.feature_table[n]._lookup(inputs[n], out=out[:, slots[n]])
        self._last = {n: inputs[n] for n in names}
        self._record_fwd_out(names, lambda n: out[:, slots[n]])

    def _record_fwd_out(self, names, view) -> None:
        """Forward output views of the features whose table carries an l2 batch regulariser (the backward adds
        2 factor out to their gradient, embedding.py:463-464); ``view(n)`` = where feature n was written."""
        if not self.has_batch_regularization:
            return
        kept = getattr(self, "_fwd_out", {})
        for n in names:
            if self.feature_table[n].l2_batch_regularization_factor > 0:
                kept[n] = view(n)
        self._fwd_out = kept

    def gather_concat(self, inputs: TabularData, names: Sequence[str], buf: torch.Tensor, offsets: Dict[str, int]) -> None:
        """One-hot feature ``n`` -> ``buf[:, offsets[n] : offsets[n] + dim]`` of a [B, W] concat buffer (the
        ConcatFeatures layout, core/aggregation.py:54-66) for all ``names`` in ONE multi-table launch."""
        ops.embedding_gather([self.feature_table[n].table.data for n in names], [inputs[n] for n in names], out=buf,
                             out_offset=[offsets[n] for n in names])
        self._last = {n: inputs[n] for n in names}
        self._record_fwd_out(names, lambda n: buf[:, offsets[n]:offsets[n] + self.feature_table[n].dim])

    def prepare_sparse(self, inputs: TabularData, names: Sequence[str]) -> None:
        """Start the id-only half of the fused sparse update (segmented sort + piece list) NOW, on the "sort" side stream,
        so that it runs beside the forward pass instead of on the step's critical path (eager steps under ``blocks.tape``
        only: a replayed hipGraph is one hardware queue).  Only the plain case is prepared: every feature one-hot, one
        embedding dim, one launch; ``_apply_sparse_now`` falls back to the single call for anything else."""
        self._prepared = None
        if not names or len(names) > 63 or not ops.SIDE.active("sort"):
            return
        fts = [self.feature_table[n] for n in names]
        if len({ft.dim for ft in fts}) != 1 or not all(ft.table.trainable for ft in fts):
            return
        if not all(self._is_onehot(inputs[n]) for n in names) or self.has_batch_regularization:
            return
        with ops.SIDE.on("sort", keep=tuple(inputs[n] for n in names)):
            h = ops.embedding_gather_backward_prepare([ft.table.data for ft in fts], [inputs[n] for n in names])
        self._prepared = (tuple(names), h) if h is not None else None

    @property
    def has_batch_regularization(self) -> bool:
        return any(t.l2_batch_regularization_factor > 0 for t in self.parallel_layers.values())

    def regularization_loss(self) -> Optional[torch.Tensor]:
        """Sum of the tables' factor * sum(out^2) terms of the LAST train step (device scalar), or None."""
        return getattr(self, "_reg_loss", None) if self.has_batch_regularization else None

    def forward(self, inputs: TabularData):
        names = [n for n in self.feature_names if n in inputs]
        if not names:
            raise ValueError("no categorical feature of the schema found in the inputs")
        dims = {self.feature_table[n].dim for n in names}
        outputs: Dict[str, torch.Tensor] = {}
        for d in sorted(dims):
            group = [n for n in names if self.feature_table[n].dim == d]
            B = (inputs[group[0]].offsets.shape[0] - 1) if isinstance(inputs[group[0]], Ragged) else inputs[group[0]].shape[0]
            buf = torch.empty((B, len(group), d), dtype=torch.float32, device=self.feature_table[group[0]].table.data.device)
            self.gather_into({n: inputs[n] for n in group}, buf, {n: i for i, n in enumerate(group)})
            for i, n in enumerate(group):
                outputs[n] = buf[:, i]
        self._last = {n: inputs[n] for n in names}
        if self.agg is not None:
            return self.agg(outputs)
        return outputs

    def tables(self) -> Dict[str, EmbeddingTable]:
        return self.parallel_layers  # type: ignore[return-value]

    # --- training: fused backward + sparse optimizer step -------------------------------------
    def set_pending_grad(self, grad: torch.Tensor, offsets: Dict[str, int], ready: bool = False) -> None:
        """``grad`` is a contiguous [B, ...] buffer; feature n's gradient row starts
        ``offsets[n]`` floats into each row (same layout the forward wrote).  ``ready``: the buffer is final at
        this point of the launch stream (an event is recorded so that the sparse update can start from here on a
        side stream while the caller keeps enqueueing independent work)."""
        self._pending = (grad, offsets)
        self._pending_event = None
        if ready and grad.is_cuda and ops.SIDE.active("sparse"):
            self._pending_event = ops.SIDE.mark()

    def backward(self, grad):
        if isinstance(grad, dict):
            raise NotImplementedError("dict-shaped embedding gradients: use the fused input block")
        return None

    def apply_sparse(self, opt) -> None:
        pending = getattr(self, "_pending", None)
        if pending is None:
            return
        grad, offsets = pending
        self._pending = None
        ev = getattr(self, "_pending_event", None)
        self._pending_event = None
        if ev is not None and ops.SIDE.active("sparse"):
            with ops.SIDE.on("sparse", after=[ev, getattr(opt, "_wait_event", None)], keep=(grad,) + tuple(self._last.values())):
                self._apply_sparse_now(opt, grad, offsets)
            ops.SIDE.maybe_join()
            return
        self._apply_sparse_now(opt, grad, offsets)

    def _apply_batch_regularization(self, g2: torch.Tensor, offsets, reset: bool = True) -> None:
        """d (factor * sum out^2) / d out = 2 factor out, added to the incoming gradient (embedding.py:463-464).  ``reset=False``: the
        step's regularisation loss keeps accumulating (MultiOptimizer: one call per optimizer group over disjoint features)."""
        if not self.has_batch_regularization:
            return
        if getattr(self, "_reg_loss", None) is None:
            self._reg_loss = torch.zeros(1, dtype=torch.float32, device=g2.device)
            reset = False
        if reset:
            self._reg_loss.zero_()
        for n in offsets:
            ft = self.feature_table[n]
            if n in self._last and ft.l2_batch_regularization_factor > 0:
                ops.l2_batch_reg(self._fwd_out[n], g2[:, offsets[n]:offsets[n] + ft.dim],
                                 ft.l2_batch_regularization_factor, self._reg_loss)

    def _apply_sparse_now(self, opt, grad, offsets, reset_reg: bool = True) -> None:
        names = [n for n in offsets if n in self._last and self.feature_table[n].table.trainable]
        lists = [n for n in names if not self._is_onehot(self._last[n])]
        names = [n for n in names if n not in lists]

        def _states(t):
            if opt.name == "adagrad":
                if "accumulator" not in t.state:
                    t.state["accumulator"] = torch.full_like(t.data, opt.initial_accumulator_value)
                return t.state["accumulator"], None
            if opt.name == "adam":
                if "m" not in t.state:
                    t.state["m"], t.state["v"] = torch.zeros_like(t.data), torch.zeros_like(t.data)
                return t.state["m"], t.state["v"]
            return None, None

        g2 = grad.reshape(grad.shape[0], -1)
        self._apply_batch_regularization(g2, offsets, reset=reset_reg)
        # A table looked up through a list feature AND any other feature (item_id + item_id_history is the common case):
        # Keras sums the IndexedSlices of all lookups of a variable before ONE optimizer apply, so Adagrad / Adam must see
        # the summed row gradient once -- two fused launches would step the shared rows twice.  Those tables go through
        # ONE dedup + update over the concatenation of every lookup's (id, gradient row) pairs.
        by_table: Dict[int, list] = {}
        for n in lists + names:
            by_table.setdefault(id(self.feature_table[n].table), []).append(n)
        merged = [grp for grp in by_table.values() if len(grp) > 1 and any(n in lists for n in grp)]
        for grp in merged:
            ft0 = self.feature_table[grp[0]]
            st, st2 = _states(ft0.table)
            ids_all, rows_all = [], []
            for n in grp:
                ft, x = self.feature_table[n], self._last[n]
                gcol = g2[:, offsets[n]:offsets[n] + ft.dim]
                if n in lists:
                    vals, offs = (x.values, x.offsets) if isinstance(x, Ragged) else (x, None)
                    v, gexp = ops.embedding_bag_expand(ft.table.data, vals, offs, gcol, ft.sequence_combiner)
                else:
                    v, gexp = x.reshape(-1), gcol
                ids_all.append(v.to(torch.int64) if v.dtype != torch.int64 else v)
                rows_all.append(gexp)
            ids_cat = torch.cat(ids_all)
            rows_cat = torch.cat(rows_all, dim=0)  # contiguous [sum nnz, D]
            ops.embedding_gather_backward([ft0.table.data], None if st is None else [st], [ids_cat], rows_cat, [0], opt.name,
                                          opt.learning_rate, opt.epsilon, None if st2 is None else [st2], opt.beta_1,
                                          opt.beta_2, opt.lr_device)
        done = {n for grp in merged for n in grp}
        lists = [n for n in lists if n not in done]
        names = [n for n in names if n not in done]
        # ragged / dense-list features over their own tables: the features of one width, combiner and id layout go through ONE
        # sort + segmented reduce + optimizer (mh_embedding_bag_bwd_multi reads a value's gradient row through its bag index)
        multi: Dict[tuple, list] = {}
        for n in lists:
            ft, x = self.feature_table[n], self._last[n]
            ragged = isinstance(x, Ragged)
            if ft.sequence_combiner in ("sum", "mean", "sqrtn") and g2.is_contiguous():
                key = (ft.dim, ft.sequence_combiner, ragged, x.values.dtype if ragged else x.dtype,
                       x.offsets.dtype if ragged else tuple(x.shape[1:]))
                multi.setdefault(key, []).append(n)
        for key, grp in multi.items():
            if len(grp) < 2:
                continue
            for start in range(0, len(grp), 63):
                part = grp[start:start + 63]
                tabs = [self.feature_table[n].table for n in part]
                sts = [_states(t) for t in tabs]
                xs = [self._last[n] for n in part]
                ops.embedding_bag_backward_multi(
                    [t.data for t in tabs], None if sts[0][0] is None else [a for a, _ in sts],
                    [x.values if key[2] else x for x in xs], [x.offsets for x in xs] if key[2] else None, g2,
                    [offsets[n] for n in part], key[1], opt.name, opt.learning_rate, opt.epsilon,
                    None if sts[0][1] is None else [b for _, b in sts], opt.beta_1, opt.beta_2, opt.lr_device)
            lists = [n for n in lists if n not in grp]
        # the rest: one fused launch chain each
        for n in lists:
            ft = self.feature_table[n]
            x = self._last[n]
            st, st2 = _states(ft.table)
            gcol = g2[:, offsets[n]:offsets[n] + ft.dim]
            vals, offs = (x.values, x.offsets) if isinstance(x, Ragged) else (x, None)
            ops.embedding_bag_backward(ft.table.data, st, vals, offs, gcol, ft.sequence_combiner, opt.name,
                                       opt.learning_rate, opt.epsilon, st2, opt.beta_1, opt.beta_2, opt.lr_device)
        for d in sorted({self.feature_table[n].dim for n in names}):
            grp = [n for n in names if self.feature_table[n].dim == d]
            tabs = [self.feature_table[n].table for n in grp]
            states = states2 = None
            if opt.name == "adagrad":
                for t in tabs:
                    if "accumulator" not in t.state:
                        t.state["accumulator"] = torch.full_like(t.data, opt.initial_accumulator_value)
                states = [t.state["accumulator"] for t in tabs]
            elif opt.name == "adam":  # LazyAdam rows
                for t in tabs:
                    if "m" not in t.state:
                        t.state["m"], t.state["v"] = torch.zeros_like(t.data), torch.zeros_like(t.data)
                states, states2 = [t.state["m"] for t in tabs], [t.state["v"] for t in tabs]
            prep = getattr(self, "_prepared", None)
            self._prepared = None
            for start in range(0, len(grp), 63):
                sl = slice(start, start + 63)
                handle = prep[1] if prep is not None and prep[0] == tuple(grp[sl]) else None
                ops.embedding_gather_backward([t.data for t in tabs[sl]], None if states is None else states[sl],
                                              [self._last[n] for n in grp[sl]], grad, [offsets[n] for n in grp[sl]],
                                              opt.name, opt.learning_rate, opt.epsilon,
                                              None if states2 is None else states2[sl], opt.beta_1, opt.beta_2, opt.lr_device,
                                              prepared=handle)


def Embeddings(schema: Schema, dim: Optional[Union[Dict[str, int], int]] = None,
               infer_dim_fn: Callable[[ColumnSchema], int] = infer_embedding_dim,
               sequence_combiner: Optional[Union[str, Dict[str, str]]] = "mean",
               embeddings_initializer=None, trainable: Optional[Union[bool, Dict[str, bool]]] = None,
               aggregation=None, block_name: str = "embeddings", device=None, seed: int = 0,
               l2_batch_regularization_factor: Optional[Union[float, Dict[str, float]]] = 0.0) -> EmbeddingsBlock:
    """embedding.py:585-683: one EmbeddingTable per categorical column; columns whose
    ``int_domain.name`` coincide share one table; ``dim=None`` -> ``infer_dim_fn(col)``."""
    schema = schema.select_by_tag(Tags.CATEGORICAL) if any(Tags.CATEGORICAL in c.tags for c in schema) else schema
    tables: Dict[str, EmbeddingTable] = {}

    def pick(opt, col):
        if isinstance(opt, dict):
            return opt.get(col.name)
        return opt

    for i, col in enumerate(schema):
        if col.int_domain is None:
            raise ValueError(f"categorical column {col.name!r} needs an int_domain")
        tname = col.int_domain.name or col.name
        if tname in tables:
            tables[tname].add_feature(col)
            continue
        d = pick(dim, col) or infer_dim_fn(col)
        kw = {}
        tr = pick(trainable, col)
        if tr is not None:
            kw["trainable"] = tr
        init = pick(embeddings_initializer, col)
        kw["l2_batch_regularization_factor"] = pick(l2_batch_regularization_factor, col) or 0.0
        tables[tname] = EmbeddingTable(d, col, sequence_combiner=pick(sequence_combiner, col),
                                       embeddings_initializer=init if init is not None else "uniform",
                                       name=tname, device=device, seed=seed + i, **kw)
    return EmbeddingsBlock(tables, schema, aggregation=aggregation, name=block_name)


class ContinuousFeatures(Block):
    """continuous.py:73-138: select the continuous columns, expand 1-D -> [B, 1]."""

    def __init__(self, features: Sequence[str], aggregation=None, name: Optional[str] = None):
        super().__init__(name)
        self.features = list(features)
        self.aggregation = parse_aggregation(aggregation)

    @classmethod
    def from_schema(cls, schema: Schema, aggregation=None, **kwargs) -> "ContinuousFeatures":
        return cls(schema.select_by_tag(Tags.CONTINUOUS).column_names, aggregation=aggregation, **kwargs)

    def forward(self, inputs: TabularData):
        out = {}
        for n in self.features:
            if n in inputs:
                v = inputs[n]
                out[n] = v.unsqueeze(-1) if v.dim() == 1 else v
        if self.aggregation is not None:
            return self.aggregation(out)
        return out

    def backward(self, grad):
        return None


Continuous = ContinuousFeatures


class InputBlockV2(Block):
    """base.py:216-341: {"categorical": Embeddings, "continuous": Continuous} merged and aggregated.

    With the default ``aggregation="concat"`` the block is fused: ONE multi-table gather writes
    every embedding straight into its sorted-name column range of the ``[B, W]`` concat buffer
    (ConcatFeatures order, core/aggregation.py:54-66) and the continuous columns are copied next
    to them -- no per-feature tensors, no torch.cat.  The backward hands the ``[B, W]`` gradient
    to the fused embedding backward with the same column offsets.
    """

    def __init__(self, schema: Schema, categorical: Optional[Block] = None, continuous: Optional[Block] = None,
                 aggregation="concat", dim=None, device=None, name: Optional[str] = None, **kwargs):
        super().__init__(name or "input_block")
        cat_schema = schema.select_by_tag(Tags.CATEGORICAL).excluding_by_tag(Tags.TARGET)
        con_schema = schema.select_by_tag(Tags.CONTINUOUS).excluding_by_tag(Tags.TARGET)
        if categorical is None and len(cat_schema):
            categorical = Embeddings(cat_schema, dim=dim, device=device, **kwargs)
        if continuous is None and len(con_schema):
            continuous = ContinuousFeatures.from_schema(con_schema)
        if categorical is None and continuous is None:
            raise ValueError("InputBlockV2: the schema has neither categorical nor continuous features")
        if aggregation not in ("concat", None):
            raise NotImplementedError("InputBlockV2 supports aggregation='concat' or None on the HIP path")
        self.schema = schema
        self.categorical: Optional[EmbeddingsBlock] = categorical
        self.continuous: Optional[ContinuousFeatures] = continuous
        self.aggregation = aggregation
        self.parallel_layers = {k: v for k, v in (("categorical", categorical), ("continuous", continuous)) if v is not None}

    def children(self):
        return list(self.parallel_layers.values())

    def forward(self, inputs: TabularData):
        cat_names = [n for n in (self.categorical.feature_names if self.categorical else []) if n in inputs]
        con = self.continuous(inputs) if self.continuous is not None else {}
        if self.aggregation is None:
            out = dict(self.categorical({n: inputs[n] for n in cat_names})) if cat_names else {}
            out.update(con)
            return out
        widths = {n: self.categorical.feature_table[n].dim for n in cat_names}
        widths.update({n: (v.shape[1] if v.dim() == 2 else v[0].numel()) for n, v in con.items()})
        order = sorted(widths)
        offsets, o = {}, 0
        for n in order:
            offsets[n] = o
            o += widths[n]
        W = o
        ld = (W + 3) // 4 * 4
        first = inputs[cat_names[0]] if cat_names else next(iter(con.values()))
        B = first.offsets.shape[0] - 1 if isinstance(first, Ragged) else first.shape[0]
        dev = (self.categorical.feature_table[cat_names[0]].table.data.device if cat_names else first.device)
        buf = torch.empty((B, ld), dtype=torch.float32, device=dev)
        if ld != W:
            ops.zero_pad_columns(buf, W)
        aligned = all(offsets[n] % 4 == 0 for n in cat_names)
        self._fused = aligned and len({widths[n] for n in cat_names}) <= 1 and all(
            self.categorical._is_onehot(inputs[n]) for n in cat_names)
        if cat_names:
            if self._fused:
                self.categorical.gather_concat(inputs, cat_names, buf, offsets)
            else:
                emb = self.categorical({n: inputs[n] for n in cat_names})
                for n in cat_names:
                    buf[:, offsets[n]:offsets[n] + widths[n]] = emb[n]
        for n, v in con.items():
            buf[:, offsets[n]:offsets[n] + widths[n]] = v.reshape(B, -1)
        self._offsets, self._W, self._ld, self._cat_names = offsets, W, ld, cat_names
        return buf[:, :W]

    def backward(self, grad):
        if grad is None or not self._cat_names:
            return None
        B = grad.shape[0]
        if not self._fused:
            # list / ragged features, mixed embedding widths or columns that do not start on a 16-byte boundary: the embedding
            # columns are compacted into one aligned buffer (every width is a multiple of 4 floats); the fused sparse update
            # takes the one-hot features from it, list features go through the bag backward (EmbeddingsBlock._apply_sparse_now)
            dims = [self.categorical.feature_table[n].dim for n in self._cat_names]
            g = torch.empty((B, sum(dims)), dtype=torch.float32, device=grad.device)
            offs, o = {}, 0
            for n, d in zip(self._cat_names, dims):
                g[:, o:o + d] = grad[:, self._offsets[n]:self._offsets[n] + d]
                offs[n] = o
                o += d
            self.categorical.set_pending_grad(g, offs)
            return None
        if grad.is_contiguous() and grad.shape[1] == self._W and self._W % 4 == 0 and grad.data_ptr() % 16 == 0:
            g = grad
        elif (grad.shape[1] == self._W and grad.stride(1) == 1 and grad.stride(0) == self._ld and grad.data_ptr() % 16 == 0
              and grad.storage_offset() + (B - 1) * self._ld + self._ld <= grad.untyped_storage().nbytes() // 4):
            # the first W columns of an ld-pitched [B, ld] buffer (what the cross / Dense backward hand out: 3341 of 3344): the
            # fused sparse update addresses rows by their pitch, so the buffer is used in place (the re-pitch below cost a
            # 0.9 GB fill + a 0.9 GB copy per DCN-v2 step)
            g = grad.as_strided((B, self._ld), (self._ld, 1))
        else:  # re-pitch to the 16-byte aligned row stride the fused backward needs
            g = torch.zeros((B, self._ld), dtype=torch.float32, device=grad.device)
            g[:, :self._W] = grad
        self.categorical.set_pending_grad(g, {n: self._offsets[n] for n in self._cat_names})
        return None


# --------------------------------------------------------------------------------------------
# V1 surface (deprecated in the reference, still the route mm.TwoTowerModel V1 takes)
# --------------------------------------------------------------------------------------------
class EmbeddingOptions:
    """inputs/embedding.py:931-942."""

    def __init__(self, embedding_dims: Optional[Dict[str, int]] = None, embedding_dim_default: Optional[int] = 64,
                 infer_embedding_sizes: bool = False, infer_embedding_sizes_multiplier: float = 2.0,
                 infer_embeddings_ensure_dim_multiple_of_8: bool = False, embeddings_initializers=None,
                 combiner: Optional[str] = "mean"):
        self.embedding_dims = embedding_dims
        self.embedding_dim_default = embedding_dim_default
        self.infer_embedding_sizes = infer_embedding_sizes
        self.infer_embedding_sizes_multiplier = infer_embedding_sizes_multiplier
        self.infer_embeddings_ensure_dim_multiple_of_8 = infer_embeddings_ensure_dim_multiple_of_8
        self.embeddings_initializers = embeddings_initializers
        self.combiner = combiner


class EmbeddingFeatures:
    """V1 EmbeddingFeatures (inputs/embedding.py:950-1156): same lookup math as EmbeddingTable
    (``lookup_feature`` :1126-1156 = tf.gather / safe_embedding_lookup_sparse), default initializer
    TruncatedNormal(0, 0.05) (:1051), default dim 64.  Built on the same fused EmbeddingsBlock."""

    @classmethod
    def from_schema(cls, schema: Schema, embedding_options: Optional[EmbeddingOptions] = None, aggregation=None,
                    device=None, **kwargs) -> EmbeddingsBlock:
        opt = embedding_options or EmbeddingOptions()
        cat = schema.select_by_tag(Tags.CATEGORICAL)
        dims: Dict[str, int] = dict(opt.embedding_dims or {})
        for col in cat:
            if col.name in dims:
                continue
            if opt.infer_embedding_sizes:
                dims[col.name] = infer_embedding_dim(col, opt.infer_embedding_sizes_multiplier,
                                                     opt.infer_embeddings_ensure_dim_multiple_of_8)
            else:
                dims[col.name] = opt.embedding_dim_default
        init = opt.embeddings_initializers if opt.embeddings_initializers is not None else "truncated_normal"
        return Embeddings(cat, dim=dims, sequence_combiner=opt.combiner, embeddings_initializer=init,
                          aggregation=aggregation, device=device)


def InputBlock(schema: Schema, aggregation="concat", embedding_options: Optional[EmbeddingOptions] = None,
               add_continuous_branch: bool = True, add_embedding_branch: bool = True, device=None, **kwargs) -> InputBlockV2:
    """V1 InputBlock (inputs/base.py:40-206): continuous + EmbeddingFeatures branches, same aggregation."""
    cat = EmbeddingFeatures.from_schema(schema, embedding_options, device=device) if (
        add_embedding_branch and len(schema.select_by_tag(Tags.CATEGORICAL))) else None
    con_schema = schema.select_by_tag(Tags.CONTINUOUS).excluding_by_tag(Tags.TARGET)
    con = ContinuousFeatures.from_schema(con_schema) if (add_continuous_branch and len(con_schema)) else None
    return InputBlockV2(schema, categorical=cat, continuous=con, aggregation=aggregation, device=device)
