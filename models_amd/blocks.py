"""Hot-path blocks of the kept ``mm`` surface (reference layer L3, merlin/models/tf/blocks/).

MLPBlock / _Dense (mlp.py:35-139, 210-300), DotProductInteraction (interaction.py:35-124),
DLRMBlock (dlrm.py:32-170), CrossBlock / Cross (cross.py:29-202), TwoTowerBlock
(retrieval/two_tower.py:32-118).
"""
from __future__ import annotations

import os

import math
from typing import List, Optional, Sequence, Union

import numpy as np
import torch
from contextlib import nullcontext as _nullcontext

from . import ops
from .core import Block, ConcatFeatures, ParallelBlock, Parameter, SequentialBlock, TabularData
from .inputs import ContinuousFeatures, Embeddings, EmbeddingsBlock, default_device
from .schema import Schema, Tags

_SUPPORTED_ACT = ("relu", "sigmoid", "linear", None)


def _glorot_uniform(fan_in: int, fan_out: int, seed: int) -> torch.Tensor:
    """keras glorot_uniform, the Dense default (mlp.py:39)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand((fan_in, fan_out), generator=g) * 2.0 - 1.0) * lim


def _truncated_normal(shape, std: float, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
    return t


_seed_counter = [1000]


def _next_seed() -> int:
    _seed_counter[0] += 1
    return _seed_counter[0]


def set_seed(seed: int) -> None:
    """Reproducible construction (the role of ``tf.keras.utils.set_random_seed`` in the reference's tests): layers built
    without an explicit ``seed=`` draw theirs from a counter; resetting it makes two models built by the same code
    identical -- in one process or in the ranks of a job."""
    _seed_counter[0] = 1000 + 7919 * int(seed)
    torch.manual_seed(int(seed))


class _Dense(Block):
    """A Dense layer that concat-aggregates dict inputs before projecting (mlp.py:210-300).
    ``kernel`` is [in, out] (Keras layout); built lazily on the first call."""

    def __init__(self, units: int, activation: Optional[str] = None, use_bias: bool = True,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", pre_aggregation="concat",
                 name: Optional[str] = None, device=None, seed: Optional[int] = None):
        super().__init__(name)
        if activation not in _SUPPORTED_ACT:
            raise NotImplementedError(f"activation {activation!r} is not fused into the Dense kernels (relu / sigmoid / linear are); "
                                      f"MLPBlock(activation=...) places {sorted(ops.ACTX)} behind a linear Dense as an Activation layer")
        self.units = int(units)
        self.activation = None if activation == "linear" else activation
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.pre_aggregation = ConcatFeatures() if pre_aggregation == "concat" else None
        self.device = torch.device(device) if device is not None else default_device()
        self.seed = _next_seed() if seed is None else seed
        self.kernel: Optional[Parameter] = None
        self.bias: Optional[Parameter] = None

    def build(self, in_dim: int) -> None:
        init = self.kernel_initializer
        if init == "glorot_uniform":
            w = _glorot_uniform(in_dim, self.units, self.seed)
        elif init == "truncated_normal":
            w = _truncated_normal((in_dim, self.units), 0.05, self.seed)
        elif callable(init):
            w = torch.as_tensor(np.asarray(init((in_dim, self.units))), dtype=torch.float32)
        else:
            w = torch.as_tensor(np.asarray(init), dtype=torch.float32)
        if tuple(w.shape) != (in_dim, self.units):
            raise ValueError(f"kernel initializer gave {tuple(w.shape)}, expected {(in_dim, self.units)}")
        self.kernel = Parameter(w.contiguous().to(self.device), name=f"{self.name}/kernel")
        if self.use_bias:
            if self.bias_initializer == "zeros":
                b = torch.zeros(self.units)
            else:
                b = torch.as_tensor(np.asarray(self.bias_initializer), dtype=torch.float32).reshape(self.units)
            self.bias = Parameter(b.to(self.device), name=f"{self.name}/bias")

    def own_parameters(self):
        return [p for p in (self.kernel, self.bias) if p is not None]

    def forward(self, inputs, out: Optional[torch.Tensor] = None):
        if isinstance(inputs, dict):
            inputs = self.pre_aggregation(inputs)
        if self.kernel is None:
            self.build(inputs.shape[-1])
        self._x = inputs
        self._y = ops.linear(inputs, self.kernel.data, None if self.bias is None else self.bias.data,
                             self.activation, out=out)
        return self._y

    def backward(self, grad, need_dx: bool = True, pre_masked: bool = False, x_activation=None, zero_pad: bool = True,
                 late_dw: Optional[list] = None):
        """``pre_masked``: ``grad`` is already dz (the consumer folded this layer's activation
        derivative into its dX epilogue).  ``x_activation``: activation that produced this layer's
        input; its derivative is folded into the returned dx.  ``zero_pad=False``: the consumer of dx never reads the
        alignment columns beyond K (saves a fill launch)."""
        dx, dW, db = ops.linear_backward(self._x, self.kernel.data, self._y, grad,
                                         None if pre_masked else self.activation, need_dx=need_dx,
                                         need_db=self.bias is not None, x_activation=x_activation, zero_pad=zero_pad,
                                         late_dw=late_dw)
        self.kernel.grad = dW
        if self.bias is not None:
            self.bias.grad = db
        return dx


_CHAIN_OK: dict = {}
_TAPE = [0]


class tape:
    """``with tape():`` -- the forward inside runs for a backward (the role of ``tf.GradientTape`` in
    BaseModel.train_step, models/base.py:1121-1174): layers keep what their backward would otherwise recompute
    (the cross layer's p = x W + b: one extra [B, d] store instead of a second d x d product)."""

    def __enter__(self):
        _TAPE[0] += 1
        return self

    def __exit__(self, *exc):
        _TAPE[0] -= 1
        return False



def _chain_supported(dims) -> bool:
    import os

    if os.environ.get("MERLIN_HIP_MLP_CHAIN", "1") == "0":  # debugging / A-B switch: every layer goes through ops.linear
        return False
    key = tuple(int(d) for d in dims)
    ok = _CHAIN_OK.get(key)
    if ok is None:
        ok = _CHAIN_OK[key] = ops.mlp_chain_supported(key)
    return ok


def mlp_forward(layers, x, out_last: Optional[torch.Tensor] = None):
    """Forward through consecutive _Dense layers.  Runs of 2-3 SMALL layers (every width <= 128: the DLRM bottom MLP,
    the tail of the top MLP with the head) go through ONE fused launch (``ops.mlp_chain``); the partition is remembered
    on the first layer so that ``mlp_backward`` mirrors it.  ``out_last``: destination of the last layer's output."""
    if isinstance(x, dict):
        x = layers[0].pre_aggregation(x)
    d = x.shape[-1]
    for l in layers:
        if l.kernel is None:
            l.build(d)
        d = l.units
    dims = [x.shape[-1]] + [l.units for l in layers]
    n, i, plan = len(layers), 0, []
    while i < n:
        run = 1
        if x.is_cuda:
            for r in (3, 2):
                if i + r <= n and _chain_supported(dims[i:i + r + 1]):
                    run = r
                    break
        plan.append((i, run))
        i += run
    layers[0]._chain_plan = plan
    for i, r in plan:
        seg = layers[i:i + r]
        last = out_last if i + r == n else None
        if r == 1:
            x = seg[0].forward(x, out=last)
            continue
        ys = ops.mlp_chain(x, [l.kernel.data for l in seg], [None if l.bias is None else l.bias.data for l in seg],
                           [l.activation for l in seg], [None] * (r - 1) + [last])
        for l, xin, y in zip(seg, [x] + ys[:-1], ys):
            l._x, l._y = xin, y
        x = ys[-1]
    return x


def mlp_backward(layers, grad, need_dx: bool = True, pre_masked: bool = False, zero_pad: bool = True,
                 late_dw_first: Optional[list] = None):
    """Backward through consecutive _Dense layers, chaining the activation derivative of layer i-1
    into the dX epilogue of layer i (no separate elementwise pass between layers); the fused runs of
    ``mlp_forward`` go back through one ``ops.mlp_chain_backward`` launch each."""
    plan = getattr(layers[0], "_chain_plan", None) if layers else None
    if not plan or sum(r for _, r in plan) != len(layers):
        plan = [(i, 1) for i in range(len(layers))]
    with ops.SIDE.deferred():  # the dW GEMMs of all layers overlap the dX chain; joined at the outermost exit
        for i, r in reversed(plan):
            prev_act = layers[i - 1].activation if i > 0 else None
            want_dx = (i > 0) or need_dx
            if r == 1:
                grad = layers[i].backward(grad, need_dx=want_dx, pre_masked=pre_masked, x_activation=prev_act,
                                          zero_pad=zero_pad or i > 0, late_dw=late_dw_first if i == 0 else None)
            else:
                seg = layers[i:i + r]
                grad, dWs, dbs = ops.mlp_chain_backward(seg[0]._x, [l.kernel.data for l in seg], [l._y for l in seg],
                                                        [l.activation for l in seg], grad, pre_masked=pre_masked,
                                                        need_dx=want_dx, need_db=[l.bias is not None for l in seg],
                                                        x_activation=prev_act)
                for l, dW, db in zip(seg, dWs, dbs):
                    l.kernel.grad = dW
                    if l.bias is not None:
                        l.bias.grad = db
            pre_masked = prev_act is not None
    return grad


class Activation(Block):
    """tf.keras.layers.Activation(name) for the Keras activations an ``MLPBlock(activation=...)`` may name beyond relu / sigmoid /
    linear (those ride in the GEMM epilogues): tanh, elu, selu, softplus, swish / silu, gelu, leaky_relu, relu6.  The Dense layer in
    front of it runs linear; this layer keeps its input for the backward (``mh_activation``)."""

    def __init__(self, activation: str, name: Optional[str] = None):
        super().__init__(name)
        if activation not in ops.ACTX:
            raise ValueError(f"Unknown activation {activation!r}; element-wise activation layers: {sorted(ops.ACTX)}")
        self.activation = activation

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._x = x
        return ops.activation(x, self.activation)

    def backward(self, grad: torch.Tensor) -> torch.Tensor:
        return ops.activation(self._x, self.activation, dy=grad)


class Dropout(Block):
    """tf.keras.layers.Dropout(rate) behind a Dense layer of an MLPBlock (mlp.py:108-114, 126-127): active under
    ``blocks.tape()`` (the training forward), identity otherwise.  The mask is counter-based (``mh_dropout``): nothing is kept
    for the backward, a replayed graph draws a new mask every step."""

    def __init__(self, rate: float, seed: Optional[int] = None, name: Optional[str] = None, device=None):
        super().__init__(name)
        if not 0.0 <= float(rate) < 1.0:
            raise ValueError(f"dropout rate must be in [0, 1), got {rate}")
        self.rate = float(rate)
        self.seed = _next_seed() if seed is None else int(seed)
        self._state: Optional[torch.Tensor] = None
        self._active = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._active = _TAPE[0] > 0 and self.rate > 0.0
        if not self._active:
            return x
        if self._state is None or self._state.device != x.device:
            self._state = torch.tensor([self.seed, 0, 0], dtype=torch.int64, device=x.device)
        return ops.dropout(x.contiguous(), self.rate, self._state)

    def backward(self, grad: torch.Tensor) -> torch.Tensor:
        if not self._active:
            return grad
        return ops.dropout(grad.contiguous(), self.rate, self._state, backward=True)


class BatchNormalization(Block):
    """tf.keras.layers.BatchNormalization() over the last axis (mlp.py:129-131; Keras defaults momentum 0.99, epsilon 1e-3,
    gamma 1, beta 0, moving mean 0 / variance 1): batch statistics under ``blocks.tape()`` (moving statistics updated), the
    moving statistics otherwise."""

    def __init__(self, momentum: float = 0.99, epsilon: float = 1e-3, center: bool = True, scale: bool = True,
                 name: Optional[str] = None, device=None):
        super().__init__(name)
        self.momentum, self.epsilon, self.center, self.scale = float(momentum), float(epsilon), center, scale
        self.device = torch.device(device) if device is not None else default_device()
        self.gamma: Optional[Parameter] = None
        self.beta: Optional[Parameter] = None
        # the moving statistics are WEIGHTS of the layer (Keras: non-trainable weights, part of save_weights / get_weights):
        # non-trainable, non-sparse Parameters, so that checkpoints / state_dict / the rank-0 broadcast carry them while every
        # optimizer skips them (no gradient, trainable=False)
        self._moving_mean: Optional[Parameter] = None
        self._moving_variance: Optional[Parameter] = None

    @property
    def moving_mean(self) -> Optional[torch.Tensor]:
        return None if self._moving_mean is None else self._moving_mean.data

    @property
    def moving_variance(self) -> Optional[torch.Tensor]:
        return None if self._moving_variance is None else self._moving_variance.data

    def build(self, n: int) -> None:
        if self.scale:
            self.gamma = Parameter(torch.ones(n, device=self.device), name=f"{self.name}/gamma")
        if self.center:
            self.beta = Parameter(torch.zeros(n, device=self.device), name=f"{self.name}/beta")
        self._moving_mean = Parameter(torch.zeros(n, device=self.device), name=f"{self.name}/moving_mean", trainable=False)
        self._moving_variance = Parameter(torch.ones(n, device=self.device), name=f"{self.name}/moving_variance", trainable=False)

    def own_parameters(self):
        return [p for p in (self.gamma, self.beta, self._moving_mean, self._moving_variance) if p is not None]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.moving_mean is None:
            self.build(x.shape[-1])
        self._training = _TAPE[0] > 0
        self._x = x
        y, self._mean, self._invstd = ops.batchnorm(x, None if self.gamma is None else self.gamma.data,
                                                    None if self.beta is None else self.beta.data, self.moving_mean,
                                                    self.moving_variance, self.epsilon, self.momentum, self._training)
        return y

    def backward(self, grad: torch.Tensor) -> torch.Tensor:
        dx, dgamma, dbeta = ops.batchnorm_backward(self._x, grad, None if self.gamma is None else self.gamma.data,
                                                   self._mean, self._invstd, self._training)
        if self.gamma is not None:
            self.gamma.grad = dgamma
        if self.beta is not None:
            self.beta.grad = dbeta
        return dx


def _dense_layers(block):
    layers = block.layers if isinstance(block, SequentialBlock) else [block]
    return layers if all(isinstance(l, _Dense) for l in layers) else None


def MLPBlock(dimensions: Sequence[int], activation: Union[str, List[str]] = "relu", use_bias: bool = True,
             kernel_initializer="glorot_uniform", bias_initializer="zeros", dropout: Optional[float] = None,
             normalization=None, filter=None, no_activation_last_layer: bool = False,
             block_name: str = "MLPBlock", device=None, seed: Optional[int] = None, **kwargs) -> SequentialBlock:
    """mlp.py:35-139: Dense layers, each optionally followed by Dropout(dropout) (not behind a last layer without activation,
    :104-114) and by BatchNormalization (``normalization="batch_norm"`` or a layer instance, :129-135)."""
    if isinstance(activation, list) and len(activation) != len(dimensions):
        raise ValueError(
            f"Activation and Dimensions length mismatch. "
            f"Activation length: {len(activation)}, Dimensions length: {len(dimensions)}"
        )
    if normalization is not None and normalization != "batch_norm" and not isinstance(normalization, Block):
        raise ValueError("Normalization needs to be an instance `Layer` or " "`batch_norm`")
    layers = []
    for idx, dim in enumerate(dimensions):
        act = (activation or "linear") if not isinstance(activation, list) else activation[idx]
        drop = None
        if no_activation_last_layer and idx == len(dimensions) - 1:
            act = "linear"
        elif dropout:
            drop = Dropout(dropout, seed=None if seed is None else seed + 100 + idx, device=device)
        post = None
        if act not in _SUPPORTED_ACT and act not in ops.ACTX:
            raise ValueError(f"Unknown activation {act!r}: fused into the Dense kernels: relu, sigmoid, linear; as an Activation "
                             f"layer: {sorted(ops.ACTX)}")
        if act in ops.ACTX:  # not an epilogue activation: Dense(linear) + an element-wise Activation layer (same function)
            post, act = Activation(act), "linear"
        layers.append(_Dense(dim, activation=act, use_bias=use_bias, kernel_initializer=kernel_initializer,
                             bias_initializer=bias_initializer, device=device,
                             seed=None if seed is None else seed + idx))
        if post is not None:
            layers.append(post)
        if drop is not None:
            layers.append(drop)
        if normalization == "batch_norm":
            layers.append(BatchNormalization(device=device))
        elif normalization is not None:
            layers.append(normalization)
    return SequentialBlock(layers, name=block_name, filter=filter)


class DotProductInteraction(Block):
    """interaction.py:35-124 with interaction_type=None, self_interaction=False."""

    def __init__(self, interaction_type=None, self_interaction: bool = False, name: Optional[str] = None):
        super().__init__(name)
        if interaction_type is not None or self_interaction:
            raise NotImplementedError("only the DLRM default (plain dot, no self interaction) is on the hot path")

    def forward(self, inputs: torch.Tensor, tail: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        self._x, self._has_tail = inputs, tail is not None
        return ops.dot_interaction(inputs, tail, out=out)

    def backward(self, grad, tail_slot: int = -1, tail_width: int = 0):
        return ops.dot_interaction_backward(self._x, grad, tail_slot, tail_width)

    def compute_output_shape(self, input_shape):
        return input_shape[0], input_shape[1] * (input_shape[1] - 1) // 2


def DotProductInteractionBlock() -> SequentialBlock:
    """dlrm.py:169-170."""
    from .core import StackFeatures

    return SequentialBlock([StackFeatures(axis=1), DotProductInteraction()])


def _last_dense_units(block: Block) -> int:
    dense = [l for l in getattr(block, "layers", [block]) if isinstance(l, _Dense)]
    if not dense:
        raise ValueError("bottom_block must contain a Dense layer")
    return dense[-1].units


class DLRMBlock(Block):
    """dlrm.py:32-133, executed as a fused pipeline:

        continuous -> bottom MLP --(last layer writes slot "bottom_block")--+
        categorical ids -> ONE multi-table gather -> slots (sorted names) --+-> stacked [B, F, D]
        stacked -> dot interaction (behind the bottom output)  -> [B, D + F(F-1)/2] -> top MLP

    The stack order is ``sorted(feature names + ["bottom_block"])`` (core/aggregation.py:101-108)
    and the concat order is **[bottom output | interactions]**: the shortcut branch of ``connect_with_shortcut``
    (dlrm.py:126-130) is ``Filter("bottom_block")``, which returns the DICT ``{"bottom_block": t}``
    (core/tabular.py:552-576); ``ParallelBlock.call`` merges dict-valued branch outputs by ``update``
    (core/combinators.py:564-569) -- the branch key "shortcut" never appears -- while the interaction branch returns a
    tensor and is keyed by its Keras name ``sequential_block_<n>``; ``ConcatFeatures`` then concatenates in sorted-key order
    (core/aggregation.py:54-66) and "bottom_block" < "sequential_block...".  The reference's torch twin states the same
    order directly: ``torch.cat((inputs["continuous"], outputs), dim=1)`` (torch/blocks/dlrm.py:102-104).  Pinned by
    ``tests/golden/make_golden.py`` (both statements executed from the reference source).  Weights of the first top-MLP
    layer exported from the reference therefore load unchanged: rows 0..D-1 multiply the bottom output.
    """

    def __init__(self, schema: Schema, *, embedding_dim: Optional[int] = None, embeddings: Optional[EmbeddingsBlock] = None,
                 bottom_block: Optional[Block] = None, top_block: Optional[Block] = None, device=None,
                 name: Optional[str] = None):
        super().__init__(name)
        if schema is None:
            raise ValueError("The schema is required by DLRM")
        con_schema = schema.select_by_tag(Tags.CONTINUOUS).excluding_by_tag(Tags.TARGET)
        cat_schema = schema.select_by_tag(Tags.CATEGORICAL).excluding_by_tag(Tags.TARGET)
        if not len(cat_schema) > 0:
            raise ValueError("DLRM requires categorical features")
        if embeddings is None:
            if embedding_dim is None:
                raise ValueError("The embedding_dim is required")
            if bottom_block is not None and embedding_dim != _last_dense_units(bottom_block):
                raise ValueError(
                    f"The embedding_dim ({embedding_dim}) needs to match the "
                    f"last layer of bottom MLP ({_last_dense_units(bottom_block)}) "
                )
            embeddings = Embeddings(cat_schema, dim=embedding_dim, sequence_combiner="mean", device=device)
        if len(con_schema) > 0:
            if bottom_block is None:
                raise ValueError(
                    "The bottom_block is required by DLRM when "
                    "continuous features are available in the schema"
                )
            self.continuous = ContinuousFeatures.from_schema(con_schema, aggregation="concat")
        else:
            self.continuous = None
            bottom_block = None
        self.schema = schema
        self.embeddings = embeddings
        self.bottom_block = bottom_block
        self.top_block = top_block
        self.interaction = DotProductInteraction()
        self.cat_names = [c.name for c in cat_schema]
        dims = {embeddings.feature_table[n].dim for n in self.cat_names}
        if len(dims) != 1:
            raise ValueError("DLRM needs one embedding dim for all categorical features")
        self.dim = dims.pop()
        keys = self.cat_names + (["bottom_block"] if self.bottom_block is not None else [])
        self.stack_order = sorted(keys)
        self.slots = {k: i for i, k in enumerate(self.stack_order)}

    def children(self):
        return [b for b in (self.embeddings, self.bottom_block, self.top_block) if b is not None]

    @property
    def num_features(self) -> int:
        return len(self.stack_order)

    def _fusable(self, inputs) -> bool:
        # gather -> interaction in one kernel (forward 123 us instead of 283 us for gather + interaction at C2; the backward
        # re-gathers the rows instead of reading a saved stack: 230 us vs 216 us) -- on whenever the geometry fits, every
        # categorical input is one-hot and the rows are local (a sharded lookup delivers rows through an exchange, the
        # l2 batch regulariser needs the gathered rows); MERLIN_HIP_FUSED_DLRM=0 forces the unfused pair.
        import os

        if os.environ.get("MERLIN_HIP_FUSED_DLRM", "1") == "0":
            return False
        D, F = self.dim, self.num_features
        if F < 2 or F * D > 2048 or F > 32 or D not in (16, 32, 64, 128):
            return False
        emb = self.embeddings
        if "gather_into" in emb.__dict__ or "gather_concat" in emb.__dict__ or emb.has_batch_regularization:
            return False
        if any(emb.feature_table[n].dim != D for n in self.cat_names):
            return False
        if not all(emb._is_onehot(inputs[n]) for n in self.cat_names):
            return False
        return len({inputs[n].dtype for n in self.cat_names}) == 1

    @property
    def accepts_head(self) -> bool:
        """The model's Dense head can ride at the end of the top MLP's fused chain (``forward(inputs, head=...)``)."""
        return self.top_block is not None and _dense_layers(self.top_block) is not None

    def _top(self, top_in: torch.Tensor, head: Optional[_Dense]):
        self._head = head
        tl = _dense_layers(self.top_block)
        if tl is None:
            out = self.top_block(top_in)
            return out if head is None else head(out)
        return mlp_forward(tl + ([head] if head is not None else []), top_in)

    def forward(self, inputs: TabularData, head: Optional[_Dense] = None):
        first = inputs[self.cat_names[0]]
        B = first.shape[0] if isinstance(first, torch.Tensor) else first.offsets.shape[0] - 1
        dev = self.embeddings.feature_table[self.cat_names[0]].table.data.device
        F, D = self.num_features, self.dim
        P = F * (F - 1) // 2
        self._fused = self._fusable(inputs)
        if self._fused:
            # gather -> stack -> interaction in ONE kernel: the stacked [B, F, D] tensor never exists in HBM
            dense = None
            if self.bottom_block is not None:
                dense = self.bottom_block(self.continuous(inputs))
            slot_tables = [None if k == "bottom_block" else self.embeddings.feature_table[k].table.data for k in self.stack_order]
            slot_ids = [None if k == "bottom_block" else inputs[k] for k in self.stack_order]
            self._slots_ctx = (slot_tables, slot_ids, dense)
            self.embeddings._last = {n: inputs[n] for n in self.cat_names}
            if self.top_block is None:
                return ops.dlrm_interaction_fused(slot_tables, slot_ids, dense, append_dense=False)
            width = P + (D if dense is not None else 0)
            ld = (width + 3) // 4 * 4
            # persistent per block: the alignment column behind `width` is zeroed ONCE (no kernel writes it), not by a fill launch
            # in every step
            slot_name = "_top_buf"
            if getattr(self, "pipeline_dw", None) is not None and _TAPE[0] > 0:
                # pipelined steps: the first top layer's dW GEMM of step t reads step t's top_in while step t + 1 writes its own
                self._top_flip = not getattr(self, "_top_flip", False)
                slot_name = "_top_buf_b" if self._top_flip else "_top_buf"
            buf = getattr(self, slot_name, None)
            if buf is None or buf.shape != (B, ld) or buf.device != dev:
                ops.park_replaced(buf)  # a captured step may still address the old one
                buf = torch.zeros((B, ld), dtype=torch.float32, device=dev)
                setattr(self, slot_name, buf)
            ops.note_captured(buf)
            top_in = buf[:, :width]
            ops.dlrm_interaction_fused(slot_tables, slot_ids, dense, append_dense=True, out=top_in)
            if _TAPE[0] > 0:
                # training, eager step: the id-only sort of the sparse update is forked HERE, behind the HBM-bound gather ->
                # interaction kernel, so that it runs beside the MFMA-bound top MLP (a light, latency-bound partner for it)
                # rather than competing with the gather for memory
                self.embeddings.prepare_sparse(inputs, self.cat_names)
            self._top_in = top_in
            self.finish_deferred()  # the first top layer's weights of the previous step's update (pipelined steps)
            return self._top(top_in, head)
        self.flush_deferred()
        stacked = torch.empty((B, F, D), dtype=torch.float32, device=dev)
        tail = None
        if self.bottom_block is not None:
            x = self.continuous(inputs)
            layers = self.bottom_block.layers if isinstance(self.bottom_block, SequentialBlock) else [self.bottom_block]
            tail = stacked[:, self.slots["bottom_block"]]
            if all(isinstance(l, _Dense) for l in layers):
                mlp_forward(layers, x, out_last=tail)  # last bottom layer writes its slot of the stack
            else:  # Dropout / BatchNormalization layers in the bottom MLP: generic path, the result is copied into its slot
                tail.copy_(self.bottom_block(x))
        self.embeddings.gather_into(inputs, stacked, self.slots)
        self._stacked = stacked
        if self.top_block is None:
            return self.interaction.forward(stacked)  # dlrm.py:120-121: interactions only
        width = P + (D if tail is not None else 0)
        ld = (width + 3) // 4 * 4  # keep rows 16-byte aligned for the next layer's vector loads
        buf = torch.empty((B, ld), dtype=torch.float32, device=dev)  # pad column is never read (guarded k-tail)
        top_in = buf[:, :width]
        self.interaction.forward(stacked, tail, out=top_in)
        self._top_in = top_in
        return self._top(top_in, head)

    @property
    def output_activation(self):
        blk = self.top_block
        tl = _dense_layers(blk) if blk is not None else None
        return tl[-1].activation if tl else None

    # ---- pipelined steps: the first top layer's dW GEMM + dense update one step later (see backward) -------------------------
    def _can_defer(self, tl, head) -> bool:
        opt = getattr(self, "pipeline_dw", None)
        if opt is None or head is None or not tl or not getattr(self, "_fused", False) or self.bottom_block is None:
            return False
        if getattr(self, "_deferred", None) is not None:  # the previous one was never launched (no forward in between): not twice
            return False
        lay = tl[0]
        if getattr(lay, "kernel_regularizer", None) is not None or getattr(lay, "bias_regularizer", None) is not None:
            return False
        return ops.SIDE.active("dw") and ops.SIDE.recorder is None and not ops.SIDE.crec and not torch.cuda.is_current_stream_capturing()

    def launch_deferred(self) -> None:
        """Start of a pipelined step: last step's dW / db of the first top layer and that layer's dense update, on the side stream."""
        rec = getattr(self, "_deferred", None)
        if rec is None or rec.get("launched"):
            return
        opt, lay = rec["opt"], rec["layer"]
        side = ops.SIDE.active("dw") and ops.SIDE.recorder is None and not torch.cuda.is_current_stream_capturing()
        ctx = ops.SIDE.on("dw", keep=rec["keep"] + (rec["dW"],) + (() if rec["db"] is None else (rec["db"],))) if side else _nullcontext()
        with ctx:
            for fn in rec["fns"]:
                fn()
            lay.kernel.grad = rec["dW"]
            params = [lay.kernel]
            if lay.bias is not None:
                lay.bias.grad = rec["db"]
                params.append(lay.bias)
            ops.dense_optimizer_step_multi(opt, params)
            for p_ in params:
                p_.grad = None
        rec["launched"], rec["side"] = True, side

    def finish_deferred(self) -> None:
        """The launch stream waits for the pipelined update (before anything reads the first top layer's weights)."""
        rec = getattr(self, "_deferred", None)
        if rec is None:
            return
        if not rec.get("launched"):
            self.launch_deferred()
        if rec.get("side"):
            ops.SIDE.join_stream("dw")
        self._deferred = None

    def flush_deferred(self) -> None:
        self.finish_deferred()

    def backward(self, grad, pre_masked: bool = False):
        D = self.dim
        head = getattr(self, "_head", None)
        late_dw = None
        if self.top_block is not None:
            tl = _dense_layers(self.top_block)
            # zero_pad=False: the interaction backward reads the P + D gradient columns only, never the alignment column of dx
            # (Launching the first top-MLP layer's dW GEMM late -- behind the bottom-MLP backward, beside the sparse apply -- was
            # measured neutral in the eager step and slower under the segmented replay, and is gone: profiles/r5_notes.md.  The list
            # below only exists for the pipelined steps of Model.pipelined_updates.)
            defer = self._can_defer(tl, head)
            late = [] if (getattr(self, "_fused", False) and self.bottom_block is not None and defer) else None
            if head is not None and tl:  # grad is the head's dz (the loss gradient w.r.t. its pre-activation)
                grad = mlp_backward(tl + [head], grad, True, pre_masked=True, zero_pad=False, late_dw_first=late)
                late_dw = late
                if defer and late:
                    # PIPELINED STEPS (Model.pipelined_updates): dW / db of the first top layer (K = P + D = 415 at C2: the one large dW
                    # GEMM of the step, 84 us alone, 260-280 us beside the interaction backward, which it slows from 190 to 230 us and
                    # behind which the sparse apply then queues) are computed at the START of the next step, on the side stream beside
                    # the HBM-bound gather -> interaction kernel, followed by this layer's dense update; the launch stream waits for
                    # both right before the layer's forward.  Same kernels, same operands, same update, one step later in wall time:
                    # weights after a flush are bit-identical to un-pipelined steps (tests/test_gpu_models.py).  Opt-in (entering the
                    # context): the interaction backward does drop to 183 us and the apply starts the moment its input
                    # exists, but the forward pays more than that (Model.pipelined_updates).
                    lay = tl[0]
                    self._deferred = {"fns": late, "layer": lay, "dW": lay.kernel.grad, "db": None if lay.bias is None else lay.bias.grad,
                                      "keep": (lay._x, grad), "opt": self.pipeline_dw}
                    lay.kernel.grad = None
                    if lay.bias is not None:
                        lay.bias.grad = None
                    late_dw = None
            else:
                if head is not None:
                    grad = head.backward(grad, pre_masked=True)
                grad = mlp_backward(tl, grad, True, pre_masked, zero_pad=False) if tl else self.top_block.backward(grad)
        has_tail = self.bottom_block is not None and self.top_block is not None
        slot = self.slots["bottom_block"] if self.bottom_block is not None else -1
        if getattr(self, "_fused", False):
            st, si, dense = self._slots_ctx
            dstack = ops.dlrm_interaction_fused_backward(st, si, dense, grad, tail_to_dense=has_tail)
        else:
            dstack = ops.dot_interaction_backward(self._stacked, grad, slot if has_tail else -1, D if has_tail else 0)
        # the embedding gradients are final here: the fused sparse update may start now, beside the bottom MLP
        self.embeddings.set_pending_grad(dstack, {n: self.slots[n] * D for n in self.cat_names}, ready=True)
        if self.bottom_block is not None:
            layers = self.bottom_block.layers if isinstance(self.bottom_block, SequentialBlock) else [self.bottom_block]
            g = dstack[:, slot]  # strided [B, D] view; overwritten in place by the activation gradient
            if all(isinstance(l, _Dense) for l in layers):
                mlp_backward(layers, g, need_dx=False)
            else:
                self.bottom_block.backward(g.contiguous())
        for fn in (late_dw or ()):
            fn()
        return None


def _widen(x: torch.Tensor, d4: int) -> torch.Tensor:
    """[B, d] -> [B, d4] with zero pad columns (d4 = d rounded up to 4): free when ``x`` is already a view
    of a zero-padded [B, d4] buffer (what InputBlockV2 / linear_backward hand out), else one padded copy."""
    B, d = x.shape
    if d == d4 and x.is_contiguous():
        return x
    if x.stride(1) == 1 and x.stride(0) == d4 and x.storage_offset() % 4 == 0:
        return x.as_strided((B, d4), (d4, 1))
    out = torch.zeros((B, d4), dtype=x.dtype, device=x.device)
    out[:, :d] = x
    return out


class Cross(Block):
    """One DCN-v2 cross layer x0 * (x W + b) + x (cross.py:113-202): full-rank kernel W [d, d], or, with
    ``low_rank_dim = r``, W = U V with U [d, r] (no bias) and V [r, d] (DenseMaybeLowRank, mlp.py:365-396).

    The feature width d (3341 on the Criteo DCN config) is rarely a multiple of 4; kernels and bias are stored
    zero-padded to d4 = ceil(d/4)*4 (r4 likewise) so that every row is 16-byte aligned for the vector loads of the
    MFMA GEMM.  Pad rows / columns stay exactly zero under SGD and Adagrad (their gradients are zero);
    ``kernel.data[:d, :d]`` (``kernel_u.data[:d, :r]``, ``kernel.data[:r, :d]``) is the reference-shaped weight."""

    def __init__(self, low_rank_dim: Optional[int] = None, name: Optional[str] = None, device=None,
                 seed: Optional[int] = None):
        super().__init__(name)
        if low_rank_dim is not None and low_rank_dim < 1:
            raise ValueError("low_rank_dim must be positive")
        self.low_rank_dim = low_rank_dim
        self.device = torch.device(device) if device is not None else default_device()
        self.seed = _next_seed() if seed is None else seed
        self.kernel: Optional[Parameter] = None
        self.kernel_u: Optional[Parameter] = None
        self.bias: Optional[Parameter] = None
        self.d = self.d4 = self.r4 = None

    def build(self, d: int):
        # CrossBlock default kernel_initializer="truncated_normal", bias "zeros" (cross.py:34-35)
        self.d, self.d4 = d, (d + 3) // 4 * 4
        if self.low_rank_dim is None:
            w = torch.zeros((self.d4, self.d4))
            w[:d, :d] = _truncated_normal((d, d), 0.05, self.seed)
        else:
            r = self.low_rank_dim
            self.r4 = (r + 3) // 4 * 4
            u = torch.zeros((self.d4, self.r4))
            u[:d, :r] = _truncated_normal((d, r), 0.05, self.seed + 1)
            self.kernel_u = Parameter(u.to(self.device), name=f"{self.name}/kernel_u")
            w = torch.zeros((self.r4, self.d4))
            w[:r, :d] = _truncated_normal((r, d), 0.05, self.seed)
        self.kernel = Parameter(w.to(self.device), name=f"{self.name}/kernel")
        self.bias = Parameter(torch.zeros(self.d4, device=self.device), name=f"{self.name}/bias")

    def own_parameters(self):
        return [p for p in (self.kernel_u, self.kernel, self.bias) if p is not None]

    def forward(self, inputs):
        x0, x = inputs if isinstance(inputs, tuple) else (inputs, inputs)
        if x0.shape != x.shape:
            raise ValueError(f"`x0` ({tuple(x0.shape)}) and `x` ({tuple(x.shape)}) shapes mismatch!")
        if self.kernel is None:
            self.build(x.shape[-1])
        d, d4 = self.d, self.d4
        x0w, xw = _widen(x0, d4), _widen(x, d4)
        self._x0, self._x = x0w, xw
        self._p = None
        if self.kernel_u is None:
            if _TAPE[0] > 0:
                out, self._p = ops.cross_layer(x0w, xw, self.kernel.data, self.bias.data, save_p=True)
                return out[:, :d]
            return ops.cross_layer(x0w, xw, self.kernel.data, self.bias.data)[:, :d]
        self._h = ops.linear(xw, self.kernel_u.data, None, None)  # [B, r4], pad columns exactly zero
        return ops.cross_layer_lowrank(x0w, xw, self._h, self.kernel.data, self.bias.data)[:, :d]

    def backward(self, dout, dx0_acc: Optional[torch.Tensor] = None):
        """out = x0 * p + x with p = x W + b (W = U V when low-rank): returns (dx0_acc, dx); sets the grads.  ``dx0_acc``: the
        running sum of d loss / d x0 over the layers of the CrossBlock -- this layer's share ``dout * p`` is ADDED to it in the
        same pass that forms g = dout * x0 (``mh_cross_layer_bwd``), and dx = g W^T + dout leaves the GEMM with the residual
        already added: no element-wise launch of its own per layer, where the first version needed four."""
        x0, x = self._x0, self._x
        d, d4 = self.d, self.d4
        dout = _widen(dout, d4)
        if not dout.is_contiguous():
            dout = dout.contiguous()
        src = x if self.kernel_u is None else self._h
        p = getattr(self, "_p", None)  # stored by the forward under blocks.tape() ...
        if p is None:
            p = ops.linear(src, self.kernel.data, self.bias.data, None)  # ... recomputed otherwise
        self._p = None
        if self.kernel_u is None:
            dx0_acc, dx, dW, db = ops.cross_layer_backward(x0, x, p, dout, self.kernel.data, dx0_acc)
            self.kernel.grad, self.bias.grad = dW, db
            return dx0_acc, dx
        # low-rank: p = (x U) V + b.  g and this layer's dx0 share by the fused element-wise ops; dh = g V^T, dV = h^T g, db through
        # the Dense backward; dx = dh U^T + dout with the residual add in the GEMM epilogue; dU = x^T dh
        dx0 = ops.eltwise("mul", dout, p) if dx0_acc is None else ops.eltwise("fma", dout, p, dx0_acc)
        g = ops.eltwise("mul", dout, x0)                             # d loss / d p
        dh, dV, db = ops.linear_backward(src, self.kernel.data, None, g, None, need_dx=True, need_db=True)
        self.kernel.grad, self.bias.grad = dV, db
        dh = _widen(dh, self.r4)
        if not dh.is_contiguous():
            dh = dh.contiguous()
        _, dU, _ = ops.linear_backward(x, self.kernel_u.data, None, dh, None, need_dx=False, need_db=False)
        self.kernel_u.grad = dU
        return dx0, ops.cross_lowrank_dx(dh, self.kernel_u.data, dout)  # both [B, d4], pad columns zero


class CrossBlock(Block):
    """cross.py:29-109: ``depth`` stacked Cross layers, x_{l+1} = x0 * (W_l x_l + b_l) + x_l."""

    def __init__(self, depth: int = 1, low_rank_dim: Optional[int] = None, name: Optional[str] = None, device=None):
        super().__init__(name)
        if depth <= 0:
            raise ValueError(f"Number of cross layers (depth) should be positive but is {depth}.")
        self.layers = [Cross(low_rank_dim=low_rank_dim, device=device) for _ in range(depth)]
        self.pre_aggregation = ConcatFeatures()

    def children(self):
        return self.layers

    def forward(self, inputs):
        if isinstance(inputs, dict):
            inputs = self.pre_aggregation(inputs)
        x0 = x = inputs
        for layer in self.layers:
            x = layer((x0, x))
        return x

    def backward(self, grad):
        dx0_total = None
        dx = grad
        for layer in reversed(self.layers):
            dx0_total, dx = layer.backward(dx, dx0_total)  # [B, d4] zero-padded; every layer adds its share of d loss / d x0
        return ops.eltwise("add", dx0_total, dx)[:, :self.layers[0].d]  # layer 0 has x = x0


class TwoTowerBlock(ParallelBlock):
    """retrieval/two_tower.py:32-118: two independent towers -> {"query": [B,E], "item": [B,E]};
    the item tower defaults to a structural copy of the query tower (:98)."""

    def __init__(self, schema: Schema, query_tower: Block, item_tower: Optional[Block] = None,
                 query_tower_tag=Tags.USER, item_tower_tag=Tags.ITEM, embedding_dim: Optional[int] = None,
                 l2_normalization: bool = False, device=None, name: Optional[str] = None):
        from .inputs import InputBlockV2

        q_schema = schema.select_by_tag(query_tower_tag)
        i_schema = schema.select_by_tag(item_tower_tag)
        if not len(q_schema) or not len(i_schema):
            raise ValueError("The schema should contain features with the tags `user` and `item`")
        if item_tower is None:
            item_tower = _copy_mlp(query_tower, device)
        q_in = InputBlockV2(q_schema, dim=embedding_dim, device=device)
        i_in = InputBlockV2(i_schema, dim=embedding_dim, device=device)
        towers = {"query": q_in.connect(query_tower, block_name="query_tower"),
                  "item": i_in.connect(item_tower, block_name="item_tower")}
        super().__init__(towers, name=name or "two_tower")
        self.l2_normalization = l2_normalization
        self.schema = schema

    @classmethod
    def from_towers(cls, query: Block, item: Block, schema: Optional[Schema] = None, l2_normalization: bool = False,
                    name: Optional[str] = None) -> "TwoTowerBlock":
        """Two ready-made towers (``mm.Encoder``s of the V2 API) instead of a schema + MLP pair."""
        self = cls.__new__(cls)
        ParallelBlock.__init__(self, {"query": query, "item": item}, name=name or "two_tower")
        self.l2_normalization = l2_normalization
        self.schema = schema
        return self

    def forward(self, inputs: TabularData):
        out = super().forward(inputs)
        if self.l2_normalization:
            self._raw = out  # tower outputs before the normalisation: its backward needs them
            out = {k: ops.l2norm(v) for k, v in out.items()}
        return out


def _copy_mlp(block: Block, device=None) -> Block:
    """Structural copy (fresh weights) of an MLP tower -- ``query_tower.copy()`` in the reference."""
    if isinstance(block, SequentialBlock):
        return SequentialBlock([_copy_mlp(l, device) for l in block.layers], name=None)
    if isinstance(block, _Dense):
        return _Dense(block.units, activation=block.activation or "linear", use_bias=block.use_bias,
                      kernel_initializer=block.kernel_initializer, bias_initializer=block.bias_initializer,
                      device=device or block.device)
    if isinstance(block, Activation):
        return Activation(block.activation)
    if isinstance(block, Dropout):
        return Dropout(block.rate, device=device)
    if isinstance(block, BatchNormalization):
        return BatchNormalization(block.momentum, block.epsilon, block.center, block.scale, device=device or block.device)
    raise NotImplementedError(f"cannot copy tower layer {type(block).__name__}")
