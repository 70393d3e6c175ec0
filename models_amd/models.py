"""Model factories of the kept ``mm`` surface (reference layer L5, merlin/models/tf/models/).

``Model`` (base.py:311-1854) is reduced to what the hot path needs: ``model(batch)`` forward,
``compile(optimizer=...)`` + ``train_step`` (base.py:1121-1174 without Keras machinery),
``fit`` over an iterable of batches, ``evaluate``.  Factories: DLRMModel / DCNModel
(ranking.py:23-168), TwoTowerModel / TwoTowerModelV2 (retrieval.py:106-486).
"""
from __future__ import annotations

import os
import time
from typing import Dict, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from .blocks import tape as blocks_tape
from . import ops, optim
from .blocks import CrossBlock, DLRMBlock, MLPBlock, TwoTowerBlock
from .core import Block, SequentialBlock, TabularData, call_layer
from .inputs import EmbeddingsBlock, InputBlockV2, Ragged
from .outputs import BinaryOutput, ContrastiveOutput, Prediction, TopKOutput
from .schema import ColumnSchema, Schema, Tags


def prepare_features(inputs: TabularData) -> TabularData:
    """The part of PrepareFeatures the hot path relies on (tf/transforms/features.py:324-379):
    ``name__values`` + ``name__offsets`` -> one ragged feature ``name``; 1-D scalars -> [B, 1]."""
    out: TabularData = {}
    for k, v in inputs.items():
        if k.endswith("__values"):
            name = k[: -len("__values")]
            out[name] = Ragged(v, inputs[name + "__offsets"])
        elif k.endswith("__offsets"):
            continue
        elif isinstance(v, torch.Tensor) and v.dim() == 1:
            out[k] = v.unsqueeze(-1)
        else:
            out[k] = v
    return out


# checkpoint layout tag written by save_weights: "2" = DLRM top-MLP input [bottom output | interactions] (the reference's
# order), BatchNormalization with four tensors (gamma, beta, moving mean, moving variance)
CHECKPOINT_FORMAT = "models_amd/2"


class Model(Block):
    """Sequence of blocks ending in an output head (models/base.py:1805-1854 ``Model.call``)."""

    def __init__(self, *blocks: Block, schema: Optional[Schema] = None, name: Optional[str] = None):
        super().__init__(name)
        self.blocks: List[Block] = list(blocks)
        self.schema = schema
        self.optimizer: Optional[optim.Optimizer] = None

    def children(self):
        return self.blocks

    @property
    def output_block(self) -> Block:
        return self.blocks[-1]

    def forward(self, inputs: TabularData, targets=None, training: bool = False, testing: bool = False):
        x = prepare_features(inputs)
        feats = x
        for block in self.blocks:
            x = call_layer(block, x, features=feats, targets=targets, training=training, testing=testing)
        return x

    def __call__(self, inputs, targets=None, training: bool = False, testing: bool = False):
        return self.forward(inputs, targets=targets, training=training, testing=testing)

    # --- training (models/base.py:311-508, 1121-1174) ---
    def compile(self, optimizer: Union[str, "optim.Optimizer"] = "adagrad", **kwargs) -> "Model":
        self.optimizer = optim.get(optimizer, **kwargs)
        return self

    def train_step(self, inputs: TabularData, targets: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    # --- pipelined steps: work of step t that nothing in step t waits for runs at the start of step t + 1 ----------------------------
    def _pipeline_blocks(self):
        return [b for b in optim._walk(self) if hasattr(b, "launch_deferred") and not isinstance(b, Model)]

    def pipelined_updates(self):
        """``with model.pipelined_updates():`` -- eager train steps inside run software-pipelined: the dW GEMM of the DLRM block's
        first top-MLP layer and that layer's dense update are launched at the START of the next step, on the side stream beside the
        HBM-bound gather -> interaction kernel, instead of beside the interaction backward (blocks.DLRMBlock.backward).  Exact: the
        same kernels on the same operands and the same update, applied before the layer's next forward; at exit (and before
        ``evaluate`` / ``predict`` / ``save_weights``) everything outstanding is flushed, the weights are then bit-identical to
        un-pipelined steps.  Only with SGD / Adagrad at a constant learning rate on one rank (a schedule, Adam's step counter or a
        dense all-reduce would have to travel with the deferred update); otherwise the context changes nothing.
        OPT-IN by entering the context (``fit`` does not): measured SLOWER on MI355X -- beside
        the deferred GEMM (84 us alone, 219 us there) the gather -> interaction kernel takes 197 us instead of 98: the two do not
        overlap, they share the CUs (1.028 vs 0.963 ms per step, profiles/r5_step_timeline_pipelined.txt)."""
        model = self

        class _Ctx:
            def __enter__(self_):
                opt = model.optimizer
                ok = (opt is not None and opt.name in ("sgd", "adagrad") and getattr(opt, "lr_device", None) is None
                      and getattr(model, "loss_grad_divisor", 1) == 1)
                self_.blocks = model._pipeline_blocks() if ok else []
                for b in self_.blocks:
                    b.pipeline_dw = opt
                model._pipelined = bool(self_.blocks)
                return model

            def __exit__(self_, *exc):
                model.flush_deferred()
                for b in self_.blocks:
                    b.pipeline_dw = None
                model._pipelined = False
                return False

        return _Ctx()

    def launch_deferred(self) -> None:
        if getattr(self, "_pipelined", False):
            for b in self._pipeline_blocks():
                b.launch_deferred()

    def flush_deferred(self) -> None:
        """Everything a pipelined step left for the next one, now (no-op otherwise)."""
        for b in self._pipeline_blocks():
            b.flush_deferred()

    @staticmethod
    def _split(batch):
        """A batch is ``inputs`` or ``(inputs, targets)`` (what ``models_amd.Loader`` yields)."""
        if isinstance(batch, tuple):
            return batch[0], (batch[1] if len(batch) > 1 else None)
        return batch, None

    def predict(self, batches) -> np.ndarray:
        """Minimal ``Model.predict`` (models/base.py): forward over the batches, outputs concatenated on the host."""
        outs = []
        self.flush_deferred()
        for batch in batches:
            x, _ = self._split(batch)
            y = self(x)
            y = y.outputs if isinstance(y, Prediction) else y
            outs.append(y.cpu().numpy())
        return np.concatenate(outs) if outs else np.zeros((0, 1), np.float32)

    def evaluate(self, batches, **kwargs) -> Dict[str, float]:
        raise NotImplementedError

    # --- checkpoint / resume (the reference relies on Keras `model.save_weights` / `load_weights`) ---
    def save_weights(self, path, skip=()) -> None:
        """Parameters (by position and name), their optimizer state (Adagrad accumulators, Adam moments) and the
        optimizer's step counter into one ``.npz``.  The model must have been built (called once).  ``skip``: ids of
        Parameters stored elsewhere (the row shards of a distributed model, written per rank)."""
        arrays: Dict[str, np.ndarray] = {}
        names = []
        self.flush_deferred()
        for i, p in enumerate(self.parameters()):
            names.append(p.name)
            if id(p) in skip:
                continue
            arrays[f"p{i}"] = p.data.detach().cpu().numpy()
            for k, v in p.state.items():
                arrays[f"s{i}:{k}"] = v.detach().cpu().numpy()
        arrays["__names__"] = np.array(names)
        arrays["__format__"] = np.array(CHECKPOINT_FORMAT)
        opt = self.optimizer
        if opt is not None and getattr(opt, "_step_dev", None) is not None:
            arrays["__adam_step__"] = opt._step_dev.detach().cpu().numpy()
        np.savez(path, **arrays)

    def load_weights(self, path, skip=(), legacy_layout: Optional[str] = None) -> None:
        """Inverse of ``save_weights`` for an identically constructed (and built) model.  Values are copied INTO the
        existing tensors (parameters, optimizer state, the on-device Adam step) wherever they exist, so a captured
        hipGraph keeps updating the restored buffers.

        ``legacy_layout``: only for checkpoints WITHOUT the ``__format__`` tag of a model holding a DLRMBlock.  Such files were
        written in one of two layouts of the first top-MLP kernel -- ``"interactions_first"`` (the oldest ones: rows
        [interactions | bottom output]; rotated here to today's order, optimizer state with them) or ``"bottom_first"`` (already the
        reference's [bottom output | interactions]; loaded as they are) -- and the file cannot say which (round-5 advisor finding:
        guessing corrupted the second kind).  Without the argument an untagged DLRM checkpoint is refused."""
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        self.flush_deferred()
        params = self.parameters()
        names = [str(n) for n in z["__names__"]]
        if len(names) != len(params):
            raise ValueError(f"checkpoint has {len(names)} parameters, the model has {len(params)}")
        if legacy_layout not in (None, "interactions_first", "bottom_first"):
            raise ValueError(f"legacy_layout={legacy_layout!r}: expected 'interactions_first' or 'bottom_first'")
        rotate = {}
        if "__format__" not in z.files:
            from .blocks import DLRMBlock, _dense_layers

            ambiguous = {}
            for blk in self.blocks_of_type(DLRMBlock):
                tl = _dense_layers(blk.top_block) if blk.top_block is not None else None
                if tl and blk.bottom_block is not None:
                    ambiguous[id(tl[0].kernel)] = blk.dim
            if ambiguous and legacy_layout is None:
                raise ValueError("untagged checkpoint of a model with a DLRMBlock: the first top-MLP kernel is stored either as "
                                 "[interactions | bottom] (oldest files) or as [bottom | interactions] (the reference's order) and the file "
                                 "does not say which; pass legacy_layout='interactions_first' or 'bottom_first'")
            if legacy_layout == "interactions_first":
                rotate = ambiguous
        elif str(z["__format__"]) != CHECKPOINT_FORMAT:
            raise ValueError(f"checkpoint format {str(z['__format__'])!r} is not {CHECKPOINT_FORMAT!r}")
        elif legacy_layout is not None:
            raise ValueError("legacy_layout is for untagged checkpoints only; this one is tagged " + CHECKPOINT_FORMAT)
        fix = lambda a, D: np.concatenate([a[-D:], a[:-D]], axis=0)
        for i, p in enumerate(params):
            if id(p) in skip:
                continue
            w = z[f"p{i}"]
            if id(p) in rotate and tuple(w.shape) == tuple(p.data.shape):
                w = fix(w, rotate[id(p)])
            if tuple(w.shape) != tuple(p.data.shape):
                raise ValueError(f"parameter {i} ({p.name}): checkpoint shape {w.shape} != model shape {tuple(p.data.shape)}")
            p.data.copy_(torch.from_numpy(w))
            prefix = f"s{i}:"
            keys = {key[len(prefix):] for key in z.files if key.startswith(prefix)}
            for k in list(p.state):
                if k not in keys:
                    del p.state[k]
            for k in keys:
                arr = z[prefix + k]
                if id(p) in rotate and tuple(arr.shape) == tuple(p.data.shape):
                    arr = fix(arr, rotate[id(p)])
                val = torch.from_numpy(arr).to(p.data.device)
                if k in p.state and p.state[k].shape == val.shape:
                    p.state[k].copy_(val)
                else:
                    p.state[k] = val
        if "__adam_step__" in z.files and self.optimizer is not None and getattr(self.optimizer, "name", "") == "adam":
            dev = params[0].data.device
            step = torch.from_numpy(z["__adam_step__"]).to(dev)  # the next tick recomputes lr_device from it
            if getattr(self.optimizer, "_step_dev", None) is not None:
                self.optimizer._step_dev.copy_(step)
            else:
                self.optimizer._step_dev = step
            if self.optimizer.lr_device is None:
                self.optimizer.lr_device = torch.zeros(1, dtype=torch.float32, device=dev)

    @property
    def graph_capturable(self) -> bool:
        """Whether a train step is a fixed launch sequence (no host-side state that changes per step)."""
        return True

    @staticmethod
    def _batch_size(x: TabularData) -> int:
        v = next(iter(x.values()))
        return (v.offsets.shape[0] - 1) if isinstance(v, Ragged) else v.shape[0]

    def fit(self, batches: Iterable, epochs: int = 1, steps_per_epoch: Optional[int] = None, graph: Optional[bool] = None):
        """Minimal fit loop; samples/sec follows ExamplesPerSecondCallback
        (tf/logging/callbacks.py:174-189): batch_size * steps / elapsed, first step discarded.
        Batches are ``inputs`` or ``(inputs, targets)`` (what ``models_amd.Loader`` yields; retrieval models take no
        targets).  ``graph=None``: when the step is a fixed launch sequence (dense inputs of one static shape, stateless
        samplers) it is captured ONCE (``graph.SegmentedStep``: per-stream hipGraph segments) and replayed with each batch copied
        into the static inputs --
        eager Python dispatch costs ~3x the kernel time of a DLRM step; batches of another shape (the last partial
        one) run eagerly."""
        from .graph import GraphedStep, PackedBatch, SegmentedStep

        # the fastest replay mode: per-stream graph segments keep the side-stream overlap of the eager step (one graph of the
        # whole step replays on one hardware queue); without side streams the single graph is the same thing with fewer launches
        Step = SegmentedStep if ops.SIDE.enabled else GraphedStep
        if self.optimizer is None:
            self.compile()
        history = {"loss": [], "examples_per_sec": []}
        graphed, sig = None, None
        # graph=None: the captured step and the eager step are BOTH timed over PROBE steps each (one device synchronisation per
        # window) and the faster one runs the rest -- on a fast host eager launches with side streams can beat the replay by a
        # few percent, on a slow or busy host the replay wins by a lot.  graph=True forces the replay, graph=False eager.
        PROBE = 12
        probe = {"replay": None, "eager": None, "t": None, "n": 0, "mode": "replay" if graph is None else None}
        prefer_eager = False

        def probe_tick(done_mode):
            """Called after every static-shape step while the two modes are being compared."""
            nonlocal prefer_eager
            if probe["mode"] is None:
                return
            if probe["t"] is None:
                torch.cuda.synchronize()
                probe["t"], probe["n"] = time.perf_counter(), 0
                return
            probe["n"] += 1
            if probe["n"] < PROBE:
                return
            torch.cuda.synchronize()
            probe[probe["mode"]] = (time.perf_counter() - probe["t"]) / probe["n"]
            probe["t"] = None
            if probe["mode"] == "replay":
                probe["mode"] = "eager"
            else:
                prefer_eager = probe["eager"] < probe["replay"] * 0.98
                probe["mode"] = None
                history["launch_probe"] = {"replay_ms": probe["replay"] * 1e3, "eager_ms": probe["eager"] * 1e3,
                                           "chosen": "eager" if prefer_eager else "replay"}

        def pack(x, y):
            d = dict(x)
            if y is not None:
                d["__targets__"] = y
            return d

        def eager(d):
            d = dict(d)
            y = d.pop("__targets__", None)
            return self.train_step(d, y)

        def can_graph(x, y):
            ok = all(isinstance(v, torch.Tensor) and v.is_cuda for v in x.values())
            return ok and (y is None or isinstance(y, torch.Tensor)) and self.graph_capturable

        try:
            return self._fit_loop(batches, epochs, steps_per_epoch, graph, Step, history, probe, probe_tick, pack, eager, can_graph,
                                  lambda: prefer_eager)
        finally:
            self.flush_deferred()  # a caller's own `with model.pipelined_updates():` around fit leaves nothing pending

    def _fit_loop(self, batches, epochs, steps_per_epoch, graph, Step, history, probe, probe_tick, pack, eager, can_graph, prefer_eager_fn):
        graphed, sig = None, None
        for _ in range(epochs):
            t0, n, steps, last = None, 0, 0, None
            for step, batch in enumerate(batches):
                if steps_per_epoch is not None and step >= steps_per_epoch:
                    break
                x, y = self._split(batch)
                use_graph = (graph is not False) and can_graph(x, y)
                if use_graph:
                    d = pack(x, y)
                    this_sig = tuple((k, tuple(v.shape), v.dtype) for k, v in d.items())
                    if graphed is None and (graph or step >= 1):  # step 0 runs eagerly: builds lazily-shaped layers
                        # warmup=0: step 0 already ran eagerly (layers built, kernels' LDS attributes set); a warm-up
                        # replay here would TRAIN on this batch several times
                        self.flush_deferred()
                        graphed, sig = Step(eager, d, warmup=0), this_sig
                    if graphed is not None and this_sig == sig:
                        if prefer_eager_fn() or probe["mode"] == "eager":
                            last = eager(d)
                        else:
                            self.flush_deferred()  # what a pipelined eager step left behind
                            last = graphed.replay(d)  # the batch's columns go into the static inputs in ONE launch
                        probe_tick(probe["mode"])
                    else:
                        last = eager(d)
                else:
                    last = self.train_step(x, y)
                if step == 0:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                else:
                    n += self._batch_size(x)
                    steps += 1
            torch.cuda.synchronize()
            for smp in getattr(getattr(self, "output", None), "negative_samplers", None) or ():  # device-side status words
                if hasattr(smp, "check_status"):
                    smp.check_status()
            if last is not None:
                history["loss"].append(float(last))
            if t0 is not None and steps:
                history["examples_per_sec"].append(n / (time.perf_counter() - t0))
        return history


class RankingModel(Model):
    """input/body block -> BinaryOutput; train_step = fwd, BCE, explicit bwd, fused updates."""

    def __init__(self, body: Block, output: BinaryOutput, schema: Schema, name: Optional[str] = None):
        super().__init__(body, output, schema=schema, name=name)
        self.body, self.output = body, output

    def _predict(self, x: TabularData) -> torch.Tensor:
        if getattr(self.body, "accepts_head", False):  # DLRM: the Dense(1, sigmoid) head rides on the top-MLP chain
            return self.body(x, head=self.output.to_call)
        return self.output(self.body(x))

    def forward(self, inputs: TabularData, targets=None, training: bool = False, testing: bool = False):
        return self._predict(prepare_features(inputs))

    def train_step(self, inputs: TabularData, targets: torch.Tensor) -> torch.Tensor:
        if self.optimizer is None:
            self.compile()
        x = prepare_features(inputs)
        fused_head = bool(getattr(self.body, "accepts_head", False))
        self.launch_deferred()  # pipelined steps: the previous step's deferred dW + update start beside this forward
        with blocks_tape():
            p = self._predict(x)
        self.optimizer.ensure_begun(p.device)
        with ops.tail_work():  # second halves nothing on the critical path waits for (BCE partial sum, chain slab reductions) go to the tail
            return self._train_tail(p, x, targets, fused_head)

    def _train_tail(self, p, x, targets, fused_head):
        loss, dlogit = self.output.loss_and_grad(p, targets)
        div = getattr(self, "loss_grad_divisor", 1)
        if div != 1:  # data parallel: gradients are partial sums of the GLOBAL-mean loss, every reduction a plain SUM
            dlogit = dlogit / div
        xa = getattr(self.body, "output_activation", None)
        with ops.SIDE.deferred():  # dW GEMMs and the sparse update run on side streams, joined after the update
            if fused_head:  # the head went forward at the end of the body's top-MLP chain: it goes back the same way
                self.body.backward(dlogit)
            else:
                dh = self.output.backward(dlogit, x_activation=xa)
                if xa is not None:
                    self.body.backward(dh, pre_masked=True)
                else:
                    self.body.backward(dh)
            self.optimizer.apply(self)
        return _with_regularization(self, loss)

    def evaluate(self, batches, **kwargs) -> Dict[str, float]:
        return _ranking_evaluate(self, batches)


def _with_regularization(model, loss: torch.Tensor) -> torch.Tensor:
    """Keras adds the layers' ``add_loss`` terms to the training loss (models/base.py:1137-1151): here the
    l2_batch_regularization terms of the embedding tables (inputs/embedding.py:463-464)."""
    for blk in model.blocks_of_type(EmbeddingsBlock):
        r = blk.regularization_loss()
        if r is not None:
            loss = loss + r[0]
    return loss


def _ranking_evaluate(model, batches) -> Dict[str, float]:
    """loss = mean BCE over all samples, binary_accuracy at threshold 0.5 and AUC (the default metrics of
    BinaryOutput, outputs/classification.py:72-123, minus precision / recall): losses come from the HIP BCE kernel,
    the rank statistic of the AUC is host bookkeeping over the collected predictions."""
    ps, ys, loss_sum, n = [], [], 0.0, 0
    for batch in batches:
        x, y = model._split(batch)
        if y is None:
            raise ValueError("evaluate needs (inputs, targets) batches")
        p = model(x)
        loss, _ = model.output.loss_and_grad(p, y, need_grad=False)
        b = p.shape[0]
        loss_sum += float(loss) * b
        n += b
        ps.append(p.reshape(-1).cpu().numpy())
        ys.append(y.reshape(-1).cpu().numpy())
    if n == 0:
        return {"loss": float("nan"), "binary_accuracy": float("nan"), "auc": float("nan")}
    p, y = np.concatenate(ps), np.concatenate(ys)
    pos, neg = int((y > 0.5).sum()), int((y <= 0.5).sum())
    auc = float("nan")
    if pos and neg:
        order = np.argsort(p, kind="stable")
        ranks = np.empty(len(p), dtype=np.float64)
        sp = p[order]
        i = 0
        while i < len(sp):  # average ranks over ties
            j = i
            while j + 1 < len(sp) and sp[j + 1] == sp[i]:
                j += 1
            ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
            i = j + 1
        auc = float((ranks[y > 0.5].sum() - pos * (pos + 1) / 2.0) / (pos * neg))
    return {"loss": loss_sum / n, "binary_accuracy": float(((p > 0.5) == (y > 0.5)).mean()), "auc": auc}


def _target_column(schema: Schema) -> Optional[ColumnSchema]:
    t = schema.select_by_tag(Tags.TARGET)
    return t.first if len(t) else None


def _binary_head(prediction_tasks, schema: Schema, device) -> BinaryOutput:
    """``prediction_tasks=``: a ``BinaryOutput`` (V2 vocabulary), a ``BinaryClassificationTask`` (V1: prediction_tasks/
    classification.py:37-130), or None (the schema's target column)."""
    if prediction_tasks is None:
        return BinaryOutput(_target_column(schema), device=device)
    if hasattr(prediction_tasks, "to_output"):
        return prediction_tasks.to_output(device=device)
    return prediction_tasks


def DLRMModel(schema: Schema, *, embedding_dim: Optional[int] = None, embeddings=None,
              bottom_block: Optional[Block] = None, top_block: Optional[Block] = None,
              prediction_tasks: Optional[BinaryOutput] = None, device=None) -> RankingModel:
    """ranking.py:23-92."""
    body = DLRMBlock(schema, embedding_dim=embedding_dim, embeddings=embeddings, bottom_block=bottom_block,
                     top_block=top_block, device=device)
    head = _binary_head(prediction_tasks, schema, device)
    return RankingModel(body, head, schema, name="dlrm_model")


class DCNBody(Block):
    """ranking.py:159-168: InputBlockV2 (concat) -> CrossBlock -> deep MLP (stacked) or
    concat(cross, deep) (parallel).

    Column order of the PARALLEL form.  The reference builds ``input_block.connect_branch(CrossBlock(depth), deep_block,
    aggregation="concat")`` (ranking.py:161-164): ``ParallelBlock(*branches)`` keys each branch by its Keras layer name
    (tf/core/combinators.py:366-368, ``use_layer_name``), both branches are ``SequentialBlock``s -- auto-named
    ``sequential_block``, ``sequential_block_1``, ... in CREATION order over the whole process -- and ``ConcatFeatures`` sorts
    those names as STRINGS (tf/core/aggregation.py:54-66).  The order is therefore not a property of the model: it depends on
    how many SequentialBlocks the process created before (``sequential_block_12`` sorts before ``sequential_block_3``; the
    import-time default ``deep_block`` of ranking.py:98 is named without a suffix and sorts first).  ``parallel_concat`` states
    the order explicitly: ``("cross", "deep")`` (default) or ``("deep", "cross")`` -- to load weights exported from a reference
    model, pass the order its two layer names sort in (``[l.name for l in model.blocks[0].layers[-1].parallel_layers]``)."""

    def __init__(self, schema: Schema, depth: int, deep_block: Block, stacked: bool = True, input_block=None,
                 embedding_dim: Optional[int] = None, device=None, low_rank_dim: Optional[int] = None,
                 parallel_concat: Sequence[str] = ("cross", "deep")):
        super().__init__("dcn_body")
        self.input_block = input_block or InputBlockV2(schema, dim=embedding_dim, device=device)
        self.cross = CrossBlock(depth, low_rank_dim=low_rank_dim, device=device)
        self.deep = deep_block
        self.stacked = stacked
        if tuple(parallel_concat) not in (("cross", "deep"), ("deep", "cross")):
            raise ValueError("parallel_concat must be ('cross', 'deep') or ('deep', 'cross')")
        self.parallel_concat = tuple(parallel_concat)

    def children(self):
        return [self.input_block, self.cross, self.deep]

    def forward(self, inputs: TabularData):
        x = self.input_block(inputs)
        if self.stacked:
            return self.deep(self.cross(x))
        c, dp = self.cross(x), self.deep(x)
        self._split = c.shape[1] if self.parallel_concat[0] == "cross" else dp.shape[1]
        return torch.cat([c, dp] if self.parallel_concat[0] == "cross" else [dp, c], dim=-1)

    def backward(self, grad):
        if self.stacked:
            g = self.deep.backward(grad)
            g = self.cross.backward(g)
            return self.input_block.backward(g)
        # parallel form: the head saw concat([cross(x), deep(x)]); both branches read the same x
        k = self._split
        first, second = grad[:, :k].contiguous(), grad[:, k:].contiguous()
        gcross, gdeep = (first, second) if self.parallel_concat[0] == "cross" else (second, first)
        gc = self.cross.backward(gcross)
        gd = self.deep.backward(gdeep)
        return self.input_block.backward(ops.eltwise("add", gc.contiguous(), gd.contiguous()))


def DCNModel(schema: Schema, depth: int, deep_block: Optional[Block] = None, stacked: bool = True,
             input_block=None, embedding_dim: Optional[int] = None, prediction_tasks=None, device=None,
             low_rank_dim: Optional[int] = None, parallel_concat: Sequence[str] = ("cross", "deep")) -> RankingModel:
    """ranking.py:95-168 (deep_block default MLPBlock([512, 256]), :98; ``**kwargs`` of the reference reach
    ``CrossBlock``, of which ``low_rank_dim`` is the one on the hot path)."""
    deep_block = deep_block or MLPBlock([512, 256], device=device)
    body = DCNBody(schema, depth, deep_block, stacked, input_block, embedding_dim, device, low_rank_dim, parallel_concat)
    head = _binary_head(prediction_tasks, schema, device)
    return RankingModel(body, head, schema, name="dcn_model")


class RetrievalModel(Model):
    """TwoTower body -> ContrastiveOutput (retrieval.py:409-486; base.py:2491-2663)."""

    def __init__(self, body: TwoTowerBlock, output: ContrastiveOutput, schema: Schema, name=None):
        super().__init__(body, output, schema=schema, name=name)
        self.body, self.output = body, output

    def forward(self, inputs: TabularData, targets=None, training: bool = False, testing: bool = False):
        x = prepare_features(inputs)
        emb = self.body(x)
        return self.output.forward({"query": emb["query"], "candidate": emb["item"]}, features=x,
                                   training=training, testing=testing)

    @property
    def graph_capturable(self) -> bool:
        return all(s.graph_capturable for s in self.output.negative_samplers)

    def train_step(self, inputs: TabularData, targets=None) -> torch.Tensor:
        """fwd (fused scorer: no [B, B] logits in HBM) -> bwd -> fused updates."""
        if self.optimizer is None:
            self.compile()
        x = prepare_features(inputs)
        emb = self.body(x)
        q, it = emb["query"], emb["item"]
        out = self.output
        ids = None
        if out.downscore_false_negatives or out.post is not None or out.logq_sampling_correction:
            ids = x[out.col_schema.name].reshape(-1)
        neg = out.negatives(it, ids, x, training=True)
        B = q.shape[0]
        if neg.n_inbatch not in (0, B):
            raise ValueError("in-batch negatives must cover the whole batch")
        pid, nid = (ids, neg.ids) if out.downscore_false_negatives else (None, None)
        kw = dict(pos_logq=neg.pos_logq, neg_logq=neg.neg_logq, logq_after_mask=neg.logq_after_mask,
                  grad_scale=1.0 / (B * getattr(self, "loss_grad_divisor", 1)))
        # one pass over the score tiles gives loss, lse AND dq (flash-style); the column pass then gives dneg
        fused = ops.inbatch_softmax_train(q, it, neg.embedding, pid, nid, out.logits_temperature, out.false_negative_score, **kw)
        if fused is not None:
            res, dq, ditem = fused
            _, _, dneg = ops.inbatch_softmax_backward(q, it, neg.embedding, res.lse, pid, nid, out.logits_temperature,
                                                      out.false_negative_score, need_dq=False, **kw)
        else:  # E > 128: tiled kernels
            res = ops.inbatch_softmax(q, it, neg.embedding, pid, nid, out.logits_temperature, out.false_negative_score,
                                      materialize=False, **{k: v for k, v in kw.items() if k != "grad_scale"})
            dq, ditem, dneg = ops.inbatch_softmax_backward(q, it, neg.embedding, res.lse, pid, nid, out.logits_temperature,
                                                           out.false_negative_score, **kw)
        if neg.n_inbatch:  # rows 0..B of the negatives ARE this batch's items; cached / sampled rows are constants here
            ditem = ops.eltwise("add", ditem, dneg if dneg.shape[0] == B else dneg[:B].contiguous())
        if self.body.l2_normalization:  # L2Norm sits between the towers and the scorer (retrieval/base.py:98-121)
            dq = ops.l2norm_backward(self.body._raw["query"], dq)
            ditem = ops.l2norm_backward(self.body._raw["item"], ditem)
        with ops.SIDE.deferred():
            self.body.parallel_layers["query"].backward(dq)
            self.body.parallel_layers["item"].backward(ditem)
            self.optimizer.apply(self)
        return _with_regularization(self, ops.mean(res.loss) if res.loss.is_cuda else res.loss.mean())

    def evaluate(self, batches, k: int = 10, **kwargs) -> Dict[str, float]:
        """In-batch evaluation as the reference runs it under ``testing=True`` (outputs/contrastive.py:223-344): every
        sample is ranked against the other items of its batch (duplicates of its own item masked), the positive sits
        in column 0 of the logits, and the top-k metrics of tf/metrics/topk.py are averaged over the samples."""
        tot, n, loss_sum = None, 0, 0.0
        for batch in batches:
            x, _ = self._split(batch)
            pred = self.forward(x, testing=True)
            logits = pred.outputs  # [B, 1 + B]
            B = logits.shape[0]
            kk = min(k, logits.shape[1])
            rank = (logits[:, 1:] > logits[:, :1]).sum(dim=1)  # ties resolve to the lower index = the positive
            labels = (torch.arange(kk, device=logits.device).unsqueeze(0) == rank.unsqueeze(1)).float()
            m = ops.topk_metrics(labels, kk, torch.ones(B, device=logits.device)).sum(dim=0)
            tot = m if tot is None else tot + m
            loss_sum += float(self.output.last_loss.mean()) * B
            n += B
        if n == 0:
            return {}
        vals = (tot / n).cpu().tolist()
        out = {f"{name}_at_{k}": v for name, v in zip(ops.TOPK_METRIC_NAMES, vals)}
        out["loss"] = loss_sum / n
        return out

    def query_embeddings(self, inputs: TabularData) -> torch.Tensor:
        return self.body.parallel_layers["query"](prepare_features(inputs))

    def candidate_embeddings(self, inputs: TabularData) -> torch.Tensor:
        return self.body.parallel_layers["item"](prepare_features(inputs))

    def to_top_k_encoder(self, candidates: torch.Tensor, identifiers: Optional[torch.Tensor] = None, k: int = 10):
        """base.py:2632-2663: query tower + TopKOutput(BruteForce.index(candidates, ids))."""
        return TopKEncoder(self.body.parallel_layers["query"], TopKOutput(candidates=candidates, identifiers=identifiers, k=k))


class TopKEncoder(Block):
    """core/encoder.py:427-482."""

    def __init__(self, query_encoder: Block, topk_layer: TopKOutput, name=None):
        super().__init__(name)
        self.query_encoder, self.topk_layer = query_encoder, topk_layer

    def forward(self, inputs: TabularData, **kwargs):
        return self.topk_layer(self.query_encoder(prepare_features(inputs)), **kwargs)

    @property
    def k(self) -> int:
        return self.topk_layer.to_call._k

    def evaluate(self, batches, item_id: str, k: Optional[int] = None) -> Dict[str, float]:
        """Retrieval evaluation against the indexed catalogue (core/encoder.py:427-482 + outputs/topk.py:224-236):
        the true item id of every query (feature ``item_id`` of the batch, or the batch's targets) is compared with
        the top-k identifiers and the ranking metrics of tf/metrics/topk.py are averaged over the queries."""
        k = self.k if k is None else k
        tot, n = None, 0
        for batch in batches:
            x, y = Model._split(batch)
            truth = y if y is not None else x[item_id]
            pred = self.forward({kk_: v for kk_, v in x.items()}, targets=truth, testing=True, k=k)
            B = pred.targets.shape[0]
            m = ops.topk_metrics(pred.targets.contiguous(), k, torch.ones(B, device=pred.targets.device)).sum(dim=0)
            tot = m if tot is None else tot + m
            n += B
        if n == 0:
            return {}
        return {f"{name}_at_{k}": v for name, v in zip(ops.TOPK_METRIC_NAMES, (tot / n).cpu().tolist())}

    def batch_predict(self, dataset, batch_size: Optional[int] = None, output_schema: Optional[Schema] = None,
                      schema: Optional[Schema] = None) -> Dict[str, np.ndarray]:
        """core/encoder.py:602-656: top-k prediction over a whole dataset in batches.  ``dataset`` is a
        ``models_amd.loader.Loader`` (read in order) or anything it accepts plus ``schema``.  Returns the columns of a
        prediction frame: ``score_0..score_{k-1}``, ``id_0..id_{k-1}`` (``TopKPrediction.output_names``), preceded by
        the input columns named in ``output_schema``."""
        from .loader import Loader
        from .outputs import TopKPrediction

        if not isinstance(dataset, Loader):
            if schema is None or batch_size is None:
                raise ValueError("batch_predict needs a Loader, or a dataset together with `schema` and `batch_size`")
            dataset = Loader(dataset, schema, batch_size, shuffle=False)
        if dataset.shuffle:
            raise ValueError("batch_predict reads the dataset in order: build the Loader with shuffle=False")
        keep = [c.name for c in output_schema] if output_schema is not None else []
        scores, ids, kept = [], [], {n: [] for n in keep}
        for inputs, _ in dataset:
            pred = self.forward(inputs)
            scores.append(pred.scores.cpu().numpy())
            ids.append(pred.identifiers.cpu().numpy())
            for n in keep:
                kept[n].append(inputs[n].reshape(inputs[n].shape[0], -1)[:, 0].cpu().numpy())
        S, I = np.concatenate(scores), np.concatenate(ids)
        out = {n: np.concatenate(v) for n, v in kept.items()}
        k = S.shape[1]
        for name, col in zip(TopKPrediction.output_names(k), list(S.T) + list(I.T)):
            out[name] = col
        return out


def TwoTowerModel(schema: Schema, query_tower: Block, item_tower: Optional[Block] = None,
                  query_tower_tag=Tags.USER, item_tower_tag=Tags.ITEM, embedding_dim: Optional[int] = None,
                  samplers=(), logits_temperature: float = 1.0, l2_normalization: bool = False,
                  downscore_false_negatives: bool = True, post_logits=None, device=None, prediction_tasks=None) -> RetrievalModel:
    """retrieval.py:106-203 (V1 route; same scorer math, item branch key "item").  ``samplers``: "in-batch" (default)
    and / or sampler objects of ``models_amd.sampling`` (e.g. ``CachedCrossBatchSampler``); ``post_logits``: a
    ``PopularityLogitsCorrection`` (the reference's logQ correction for popularity-biased in-batch negatives)."""
    body = TwoTowerBlock(schema, query_tower, item_tower, query_tower_tag, item_tower_tag, embedding_dim,
                         l2_normalization, device)
    item_id = schema.select_by_tag(Tags.ITEM_ID)
    if prediction_tasks is not None:  # V1 vocabulary: ItemRetrievalTask(schema, samplers=..., logits_temperature=...) (retrieval.py:163-177)
        out = prediction_tasks.to_output(downscore_false_negatives) if hasattr(prediction_tasks, "to_output") else prediction_tasks
        return RetrievalModel(body, out, schema, name="two_tower_model")
    out = ContrastiveOutput(item_id.first if len(item_id) else None, list(samplers) if samplers else "in-batch",
                            downscore_false_negatives=downscore_false_negatives and len(item_id) > 0,
                            logits_temperature=logits_temperature, post=post_logits)
    return RetrievalModel(body, out, schema, name="two_tower_model")


class Encoder(SequentialBlock):
    """core/encoder.py:41-239: ``InputBlockV2(schema)`` followed by the given blocks -- one tower of the V2 API.
    ``encode`` is the batched embedding export of ``Encoder.encode / batch_predict`` (:155-208)."""

    def __init__(self, inputs: Union[Schema, Block], *blocks: Block, device=None, name: Optional[str] = None):
        if isinstance(inputs, Schema):
            self._schema = inputs
            input_block: Block = InputBlockV2(inputs, device=device)
        else:
            input_block = inputs
            self._schema = getattr(inputs, "schema", None)
        layers: List[Block] = [input_block]
        for b in blocks:
            layers.extend(b.layers if isinstance(b, SequentialBlock) else [b])
        super().__init__(layers, name=name or "encoder")

    @property
    def schema(self) -> Optional[Schema]:
        return self._schema

    def forward(self, inputs, **kwargs):
        return super().forward(prepare_features(inputs) if isinstance(inputs, dict) else inputs, **kwargs)

    def encode(self, dataset, batch_size: Optional[int] = None) -> np.ndarray:
        """Embeddings of a whole dataset (a ``Loader`` read in order, or anything it accepts + ``batch_size``)."""
        from .loader import Loader

        if not isinstance(dataset, Loader):
            if batch_size is None or self._schema is None:
                raise ValueError("encode needs a Loader, or a dataset together with `batch_size` (and a schema)")
            dataset = Loader(dataset, self._schema, batch_size, shuffle=False)
        if dataset.shuffle:
            raise ValueError("encode reads the dataset in order: build the Loader with shuffle=False")
        return np.concatenate([self.forward(x).cpu().numpy() for x, _ in dataset])


def TwoTowerModelV2(query_tower: Encoder, candidate_tower: Encoder, candidate_id_tag=Tags.ITEM_ID, outputs=None,
                    logits_temperature: float = 1.0, negative_samplers=None, schema: Optional[Schema] = None,
                    downscore_false_negatives: bool = True) -> RetrievalModel:
    """retrieval.py:409-486: two ``Encoder``s -> ContrastiveOutput(DotProduct, in-batch negatives)."""
    if not isinstance(query_tower, Encoder) or not isinstance(candidate_tower, Encoder):
        raise ValueError("The query and candidate towers should be instances of the `Encoder` class")
    if outputs is None:
        item_id = candidate_tower.schema.select_by_tag(candidate_id_tag) if candidate_tower.schema is not None else []
        outputs = ContrastiveOutput(item_id.first if len(item_id) else None, negative_samplers or "in-batch",
                                    downscore_false_negatives=downscore_false_negatives and len(item_id) > 0,
                                    logits_temperature=logits_temperature)
    body = TwoTowerBlock.from_towers(query_tower, candidate_tower, schema)
    return RetrievalModel(body, outputs, schema, name="two_tower_model")
