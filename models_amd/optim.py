"""Optimizers of the explicit train step (models/base.py:476-508, 1121-1174).

Dense parameters (MLP / cross / head weights) are updated by a HIP elementwise kernel; sparse
parameters (embedding tables) are updated row-wise inside the fused embedding backward
(``mh_embedding_gather_bwd``) and never see a dense gradient.
"""
from __future__ import annotations

from typing import Union

import torch


class Optimizer:
    name = "sgd"

    def __init__(self, learning_rate: float = 0.01, epsilon: float = 1e-7, initial_accumulator_value: float = 0.1,
                 beta_1: float = 0.9, beta_2: float = 0.999):
        self.learning_rate = float(learning_rate)
        self.epsilon = float(epsilon)
        self.initial_accumulator_value = float(initial_accumulator_value)
        self.beta_1, self.beta_2 = float(beta_1), float(beta_2)
        self.lr_device = None  # Adam: bias-corrected lr kept on the device (graph-replayable)
        self._step_dev = None

    def begin_step(self, device) -> None:
        """Per-step prologue (Adam advances its on-device step counter / bias-corrected lr)."""

    def ensure_begun(self, device) -> None:
        """Run the per-step prologue once per step.  train_step calls this BEFORE the backward so that the
        sparse update, which starts on a side stream as soon as the embedding gradients are final, is ordered
        after it."""
        if not getattr(self, "_begun", False):
            self.begin_step(device)
            self._begun = True

    def apply(self, model) -> None:
        from . import ops

        params = model.parameters()
        late = not getattr(self, "_begun", False)
        if params:
            self.ensure_begun(params[0].data.device)
        # prologue ran only now: a sparse update forked from an earlier event must also wait for it
        self._wait_event = None
        if late and self.lr_device is not None and ops.SIDE.active("sparse"):
            self._wait_event = ops.SIDE.mark()
        with ops.SIDE.deferred():
            # sparse first: it leaves for its side stream (if the gradients were announced ready) and runs beside
            # the tail of the backward and the dense update below
            for blk in _walk(model):
                if hasattr(blk, "apply_sparse"):
                    blk.apply_sparse(self)
            ops.run_tail()  # parked slab reductions (chain dW / db) and the BCE sum: here the launch stream idles behind the sparse apply
            ops.SIDE.join_stream("dw")  # the dW / db GEMMs of the MLP backward ran on their own stream
            dense = [p for p in params if not p.sparse and p.trainable and p.grad is not None]  # (one walk of the model per step)
            ops.dense_optimizer_step_multi(self, dense)  # one launch for all MLP / cross / head tensors
        self._begun = False


def _walk(block):
    """Pre-order walk of the block tree (iterative: called every step)."""
    stack = [block]
    while stack:
        b = stack.pop()
        yield b
        ch = b.children()
        if ch:
            stack.extend(reversed(ch if isinstance(ch, (list, tuple)) else list(ch)))


class SGD(Optimizer):
    name = "sgd"


class Adagrad(Optimizer):
    """keras Adagrad: acc += g^2; w -= lr * g / (sqrt(acc) + eps); acc0 = 0.1, eps = 1e-7."""

    name = "adagrad"

    def __init__(self, learning_rate: float = 0.001, **kw):
        super().__init__(learning_rate, **kw)


class Adam(Optimizer):
    """keras Adam for dense tensors; embedding tables get the LazyAdam row-wise variant the reference ships
    for large tables (blocks/optimizer.py:342-437): only rows present in the batch move their moments."""

    name = "adam"

    def __init__(self, learning_rate: float = 0.001, **kw):
        super().__init__(learning_rate, **kw)

    def begin_step(self, device) -> None:
        from . import ops

        if self._step_dev is None:
            self._step_dev = torch.zeros(1, dtype=torch.float32, device=device)
            self.lr_device = torch.zeros(1, dtype=torch.float32, device=device)
        ops.adam_tick(self)


LazyAdam = Adam


def get(opt: Union[str, Optimizer], **kwargs) -> Optimizer:
    if isinstance(opt, Optimizer):
        return opt
    table = {"sgd": SGD, "adagrad": Adagrad, "adam": Adam, "lazy_adam": Adam}
    if opt not in table:
        raise ValueError(f"unknown optimizer {opt!r}; on the HIP path: {sorted(table)}")
    return table[opt](**kwargs)
