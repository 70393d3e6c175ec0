"""Block algebra of the kept ``mm`` surface (reference layer L1, merlin/models/tf/core/).

The reference composes Keras layers (``Block.connect`` -> ``SequentialBlock``
core/base.py:306-337, ``ParallelBlock`` core/combinators.py:318-571, tabular aggregations
core/aggregation.py).  Here a Block is a plain Python object that owns device tensors
(``Parameter``) and implements an explicit ``forward`` and ``backward`` -- there is no autograd
tape and no tracing compiler: the step is a fixed sequence of HIP kernel launches on one stream.

Ordering contracts reproduced exactly (they change column order silently otherwise):
  * ConcatFeatures / StackFeatures iterate ``sorted(inputs.keys())`` (aggregation.py:54-66,
    101-108);
  * ParallelBlock merges dict-valued branches by ``update`` and keys tensor-valued branches by
    branch name (combinators.py:546-571).
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Union

import torch

TabularData = Dict[str, torch.Tensor]


class Parameter:
    """A trainable device tensor + its gradient slot.  ``sparse`` marks embedding tables, whose
    gradient is applied row-wise by the fused HIP backward and never materialised densely."""

    def __init__(self, data: torch.Tensor, name: str = "", trainable: bool = True, sparse: bool = False):
        self.data = data
        self.name = name
        self.trainable = trainable
        self.sparse = sparse
        self.grad: Optional[torch.Tensor] = None
        self.state: Dict[str, torch.Tensor] = {}

    @property
    def shape(self):
        return self.data.shape

    @property
    def embeddings(self) -> torch.Tensor:
        """``EmbeddingTable.table.embeddings`` of the reference (the Keras Embedding layer's weight variable)."""
        return self.data

    def numpy(self):
        return self.data.detach().cpu().numpy()

    def __repr__(self):
        return f"Parameter({self.name!r}, shape={tuple(self.data.shape)}, sparse={self.sparse})"


def call_layer(layer: Callable, inputs, **kwargs):
    """Filter kwargs against the callee's signature (tf/utils/tf_utils.py:433-451)."""
    fn = layer.forward if isinstance(layer, Block) else layer
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return layer(inputs)
    params = sig.parameters
    if any(p.kind == inspect.Parameter.VAR_KEYWORD for p in params.values()):
        return layer(inputs, **kwargs)
    return layer(inputs, **{k: v for k, v in kwargs.items() if k in params})


class Block:
    """Base of every layer-like object (reference: core/base.py:160)."""

    _counter: Dict[str, int] = {}

    def __init__(self, name: Optional[str] = None):
        base = name or _snake(type(self).__name__)
        if name is None:
            n = Block._counter.get(base, 0)
            Block._counter[base] = n + 1
            base = base if n == 0 else f"{base}_{n}"
        self.name = base
        self.training = False

    # --- call protocol ---
    def __call__(self, inputs, **kwargs):
        return call_layer(self.forward, inputs, **kwargs) if kwargs else self.forward(inputs)

    def forward(self, inputs, **kwargs):  # pragma: no cover - abstract
        raise NotImplementedError

    def backward(self, grad):
        raise NotImplementedError(f"{type(self).__name__} has no backward")

    # --- composition (core/base.py:306-406) ---
    def connect(self, *blocks: "Block", block_name: Optional[str] = None) -> "SequentialBlock":
        layers: List[Block] = list(self.layers) if isinstance(self, SequentialBlock) else [self]
        for b in blocks:
            layers.extend(b.layers if isinstance(b, SequentialBlock) else [b])
        return SequentialBlock(layers, name=block_name)

    # --- parameters ---
    def children(self) -> Iterable["Block"]:
        return ()

    def own_parameters(self) -> List[Parameter]:
        return []

    def parameters(self) -> List[Parameter]:
        """Distinct parameters in pre-order (a block's own, then its children's, in order).  ONE iterative walk: the optimizer
        calls this every step, and the recursive form (a list and a set rebuilt at every level of the tree) was the largest single
        item of the host time of an eagerly launched step (tools/dbg/host_profile.py: 0.7 of 2.2 ms on the authoring host)."""
        seen, out, stack = set(), [], [self]
        while stack:
            b = stack.pop()
            for p in b.own_parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
            ch = b.children()
            if ch:
                stack.extend(reversed(ch if isinstance(ch, (list, tuple)) else list(ch)))
        return out

    def blocks_of_type(self, cls) -> List["Block"]:
        """Every distinct descendant (self included) that is an instance of ``cls``."""
        seen, out, stack = set(), [], [self]
        while stack:
            b = stack.pop()
            if id(b) in seen:
                continue
            seen.add(id(b))
            if isinstance(b, cls):
                out.append(b)
            stack.extend(b.children())
        return out

    def train(self, mode: bool = True) -> "Block":
        self.training = mode
        for c in self.children():
            c.train(mode)
        return self

    def eval(self) -> "Block":
        return self.train(False)

    def state_dict(self) -> Dict[str, Any]:
        return {p.name: p.data.detach().cpu().numpy() for p in self.parameters()}


def _snake(name: str) -> str:
    out = []
    for i, ch in enumerate(name):
        if ch.isupper() and i and not name[i - 1].isupper():
            out.append("_")
        out.append(ch.lower())
    return "".join(out).lstrip("_")


class SequentialBlock(Block):
    """core/combinators.py:32: apply layers in order; kwargs are filtered per layer."""

    def __init__(self, layers: Sequence[Block], name: Optional[str] = None, filter: Optional["Filter"] = None):
        super().__init__(name or None)
        self.layers = list(layers)
        self.filter = filter

    def forward(self, inputs, **kwargs):
        x = inputs
        if self.filter is not None and isinstance(x, dict):
            x = self.filter(x)
        from .blocks import _Dense, mlp_forward  # local import: blocks imports core

        layers, i, n = self.layers, 0, len(self.layers)
        while i < n:
            j = i
            while j < n and isinstance(layers[j], _Dense):
                j += 1
            if j - i >= 2:  # a run of Dense layers: small ones are fused into one launch (blocks.mlp_forward)
                x = mlp_forward(layers[i:j], x)
                i = j
                continue
            x = call_layer(layers[i], x, **kwargs)
            i += 1
        return x

    def backward(self, grad):
        from .blocks import _Dense, mlp_backward  # local import: blocks imports core

        layers = list(self.layers)
        while layers:
            # peel the longest trailing run of Dense layers and back-propagate it as one fused chain
            j = len(layers)
            while j > 0 and isinstance(layers[j - 1], _Dense):
                j -= 1
            if j < len(layers):
                grad = mlp_backward(layers[j:], grad, need_dx=True)
                layers = layers[:j]
            else:
                grad = layers.pop().backward(grad)
        return grad

    def children(self):
        return self.layers

    def __getitem__(self, i):
        return self.layers[i]

    def __len__(self):
        return len(self.layers)


class Filter(Block):
    """Select named features from a dict (core/tabular.py Filter)."""

    def __init__(self, names: Union[str, Sequence[str]], name: Optional[str] = None):
        super().__init__(name)
        self.names = [names] if isinstance(names, str) else list(names)

    def forward(self, inputs: TabularData):
        return {k: v for k, v in inputs.items() if k in self.names}

    def backward(self, grad):
        return grad


# --- aggregations (core/aggregation.py) ---------------------------------------------------------
def _as_2d(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(t.shape[0], -1) if t.dim() != 2 else t


class ConcatFeatures(Block):
    """aggregation.py:38-66: concat over the last axis in sorted-key order (fp32).
    Wide parts never come here: the fused model paths let the producer kernels write straight into the concatenated
    buffer.  Narrow fp32 device columns (the continuous features) take ``mh_concat_columns``; anything else (CPU tensors,
    other dtypes, very wide inputs) is a plain ``torch.cat`` (buffer plumbing)."""

    def forward(self, inputs: TabularData):
        self._keys = sorted(inputs)
        cols = [_as_2d(inputs[k]) for k in self._keys]
        self._widths = [c.shape[1] for c in cols]
        if (cols and len(cols) <= 64 and sum(self._widths) <= 144
                and all(c.is_cuda and c.dtype == torch.float32 for c in cols)):
            # narrow device columns (the continuous features of a batch): ONE launch, rows padded to 16 bytes with zeros
            from . import ops

            return ops.concat_columns(cols, pad_to=4)
        return torch.cat([c.float() for c in cols], dim=-1)

    def backward(self, grad):
        out, o = {}, 0
        for k, w in zip(self._keys, self._widths):
            out[k] = grad[:, o:o + w]
            o += w
        return out


class StackFeatures(Block):
    """aggregation.py:85-108: stack on ``axis`` in sorted-key order."""

    def __init__(self, axis: int = 1, name: Optional[str] = None):
        super().__init__(name)
        self.axis = axis

    def forward(self, inputs: TabularData):
        self._keys = sorted(inputs)
        return torch.stack([inputs[k].float() for k in self._keys], dim=self.axis)

    def backward(self, grad):
        return {k: grad.select(self.axis, i) for i, k in enumerate(self._keys)}


AGGREGATIONS = {"concat": ConcatFeatures, "stack": StackFeatures}


def parse_aggregation(agg) -> Optional[Block]:
    if agg is None or isinstance(agg, Block):
        return agg
    if agg not in AGGREGATIONS:
        raise ValueError(f"unknown aggregation {agg!r}; registered: {sorted(AGGREGATIONS)}")
    return AGGREGATIONS[agg]()


class ParallelBlock(Block):
    """core/combinators.py:318-571: run named branches on the same input, merge the outputs."""

    def __init__(self, branches: Union[Dict[str, Block], Sequence[Block]], aggregation=None,
                 name: Optional[str] = None, schema=None, pre: Optional[Block] = None, post: Optional[Block] = None):
        super().__init__(name)
        if not isinstance(branches, dict):
            branches = {b.name: b for b in branches}
        self.parallel_layers: Dict[str, Block] = dict(branches)
        self.aggregation = parse_aggregation(aggregation)
        self.schema = schema
        self.pre, self.post = pre, post

    def select_by_name(self, name: str) -> Block:
        return self.parallel_layers[name]

    def __getitem__(self, name: str) -> Block:
        return self.parallel_layers[name]

    def forward(self, inputs, **kwargs):
        if self.pre is not None:
            inputs = call_layer(self.pre, inputs, **kwargs)
        outputs: Dict[str, torch.Tensor] = {}
        self._kinds = {}
        for name, layer in self.parallel_layers.items():
            out = call_layer(layer, inputs, **kwargs)
            if isinstance(out, dict):
                outputs.update(out)
                self._kinds[name] = list(out)
            else:
                outputs[name] = out
                self._kinds[name] = None
        if self.post is not None:
            outputs = call_layer(self.post, outputs, **kwargs)
        if self.aggregation is not None:
            return self.aggregation(outputs)
        return outputs

    def backward(self, grad):
        if self.aggregation is not None:
            grad = self.aggregation.backward(grad)
        for name, layer in self.parallel_layers.items():
            keys = self._kinds[name]
            g = grad[name] if keys is None else {k: grad[k] for k in keys if k in grad}
            layer.backward(g)
        return None

    def children(self):
        extra = [b for b in (self.pre, self.post, self.aggregation) if b is not None]
        return list(self.parallel_layers.values()) + extra


class TabularBlock(Block):
    """core/tabular.py:100: ``pre -> call -> post -> aggregation`` wrapper for dict outputs."""

    def __init__(self, aggregation=None, name: Optional[str] = None):
        super().__init__(name)
        self.aggregation = parse_aggregation(aggregation)

    def __call__(self, inputs, **kwargs):
        out = super().__call__(inputs, **kwargs)
        if self.aggregation is not None and isinstance(out, dict):
            out = self.aggregation(out)
        return out
