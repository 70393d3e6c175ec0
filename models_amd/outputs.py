"""Model heads of the kept ``mm`` surface (reference layer L4, merlin/models/tf/outputs/).

BinaryOutput (classification.py:72-123), DotProduct (base.py:291-322), ContrastiveOutput
(contrastive.py:47-405), BruteForce / TopKOutput (topk.py:129-303).
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Sequence, Union

import numpy as np
import torch

from . import ops
from .blocks import _Dense
from .core import Block
from .schema import ColumnSchema, Schema, Tags

# merlin/models/utils/constants.py:19
MIN_FLOAT = float(np.finfo(np.float16).min) / 100.0


class Prediction(NamedTuple):
    """outputs/base.py Prediction(outputs, targets, ...)."""

    outputs: torch.Tensor
    targets: Optional[torch.Tensor] = None
    negative_candidate_ids: Optional[torch.Tensor] = None


class TopKPrediction(NamedTuple):
    """outputs/topk.py TopKPrediction(scores, identifiers)."""

    scores: torch.Tensor
    identifiers: torch.Tensor

    @staticmethod
    def output_names(k: int):
        """Column names of a batch-prediction frame (core/prediction.py:107-115)."""
        return [f"score_{i}" for i in range(k)] + [f"id_{i}" for i in range(k)]


class BinaryOutput(Block):
    """classification.py:72-123: Dense(1, activation="sigmoid") head, binary cross-entropy loss."""

    def __init__(self, target: Optional[Union[str, ColumnSchema]] = None, name: Optional[str] = None, device=None):
        super().__init__(name)
        self.target = target.name if isinstance(target, ColumnSchema) else target
        self.to_call = _Dense(1, activation="sigmoid", device=device)

    def children(self):
        return [self.to_call]

    def forward(self, inputs):
        return self.to_call(inputs)

    def loss_and_grad(self, predictions: torch.Tensor, targets: torch.Tensor, need_grad: bool = True):
        """Mean BCE over the batch (Keras SUM_OVER_BATCH_SIZE) and d(mean loss)/d(logit)."""
        return ops.bce(predictions, targets, need_grad=need_grad)

    def backward(self, dlogit: torch.Tensor, x_activation=None):
        # dlogit is the gradient w.r.t. the PRE-sigmoid activation -> bypass the sigmoid derivative;
        # x_activation = activation of the body's last layer (its derivative is folded into dx)
        d = self.to_call
        dx, dW, db = ops.linear_backward(d._x, d.kernel.data, d._y, dlogit, None, need_dx=True,
                                         need_db=d.bias is not None, x_activation=x_activation)
        d.kernel.grad = dW
        if d.bias is not None:
            d.bias.grad = db
        return dx


class DotProduct(Block):
    """outputs/base.py:291-322: sum(query * item, -1, keepdims=True)."""

    def __init__(self, query_name: str = "query", item_name: str = "candidate", name: Optional[str] = None):
        super().__init__(name)
        self.query_name, self.item_name = query_name, item_name

    def forward(self, inputs):
        return ops.rowwise_dot(inputs[self.query_name], inputs[self.item_name])


class ContrastiveOutput(Block):
    """contrastive.py:47-405 with the in-batch sampler (outputs/sampling/in_batch.py:25-114):
    training/testing -> logits [B, 1+B] (positives in column 0), false negatives (incl. the
    diagonal) rescored to ``false_negative_score``; inference -> positive scores [B, 1]."""

    def __init__(self, to_call: Optional[Union[Schema, ColumnSchema, Block]] = None,
                 negative_samplers: Union[str, Sequence[str]] = "in-batch", downscore_false_negatives: bool = True,
                 false_negative_score: float = MIN_FLOAT, logits_temperature: float = 1.0,
                 store_negative_ids: bool = False, query_name: str = "query", candidate_name: str = "candidate",
                 logq_sampling_correction: bool = False, name: Optional[str] = None):
        super().__init__(name)
        samplers = [negative_samplers] if isinstance(negative_samplers, str) else list(negative_samplers)
        if samplers != ["in-batch"]:
            raise NotImplementedError("only the 'in-batch' negative sampler is on the HIP hot path")
        if logq_sampling_correction:
            # contrastive.py:309-319 needs sampling probabilities, which only the popularity sampler provides
            # (outputs/sampling/popularity.py); the in-batch sampler has none (the reference warns and fails there)
            raise NotImplementedError("logq_sampling_correction needs a sampler with sampling probabilities "
                                      "(popularity-based); only the in-batch sampler is on the HIP hot path")
        if isinstance(to_call, Schema):
            to_call = to_call.select_by_tag(Tags.ITEM_ID).first
        self.col_schema = to_call if isinstance(to_call, ColumnSchema) else None
        self.downscore_false_negatives = downscore_false_negatives
        self.false_negative_score = float(false_negative_score)
        self.logits_temperature = float(logits_temperature)
        self.store_negative_ids = store_negative_ids
        self.query_name, self.candidate_name = query_name, candidate_name

    def forward(self, inputs, features=None, targets=None, training: bool = False, testing: bool = False,
                materialize: bool = True):
        q, c = inputs[self.query_name], inputs[self.candidate_name]
        if not (training or testing):
            return ops.rowwise_dot(q, c)  # contrastive.py:221 -> DotProduct
        ids = None
        if self.downscore_false_negatives:
            if features is None or self.col_schema is None or self.col_schema.name not in features:
                raise ValueError("downscore_false_negatives needs the item-id feature in `features`")
            ids = features[self.col_schema.name].reshape(-1)
        self._q, self._c, self._ids = q, c, ids
        res = ops.inbatch_softmax(q, c, c, ids, ids, self.logits_temperature, self.false_negative_score,
                                  materialize=materialize)
        self._lse = res.lse
        self.last_loss = res.loss
        if not materialize:
            return res
        tg = torch.zeros_like(res.logits)
        tg[:, 0] = 1.0
        return Prediction(res.logits, tg, ids if self.store_negative_ids else None)


class BruteForce(Block):
    """topk.py:129-237: brute-force top-k over an indexed candidate matrix."""

    def __init__(self, k: int = 10, name: Optional[str] = None):
        super().__init__(name)
        self._k = int(k)
        self._candidates: Optional[torch.Tensor] = None
        self._ids: Optional[torch.Tensor] = None

    def index(self, candidates: torch.Tensor, identifiers: Optional[torch.Tensor] = None) -> "BruteForce":
        if candidates.dim() != 2:
            raise ValueError(f"candidates must be 2-D tensor (got {tuple(candidates.shape)})")
        if identifiers is None:
            identifiers = torch.arange(candidates.shape[0], device=candidates.device)
        identifiers = identifiers.reshape(-1)
        if identifiers.shape[0] != candidates.shape[0]:
            raise ValueError(
                "The candidates and identifiers tensors must have the same number of rows "
                f"(got {candidates.shape[0]} candidates rows and {identifiers.shape[0]} identifier rows)."
            )
        self._ids = identifiers.to(torch.int32).contiguous()  # topk.py:162-179: ids stored as int32
        self._candidates = candidates.to(torch.float32).contiguous()
        return self

    def forward(self, inputs: torch.Tensor, targets: Optional[torch.Tensor] = None, testing: bool = False,
                k: Optional[int] = None):
        k = self._k if k is None else k
        if self._candidates is None:
            raise ValueError("You should call the `index` method first to set the _candidates index.")
        if inputs.shape[1] != self._candidates.shape[1]:
            raise ValueError(
                "Query and candidates vectors must have the same embedding size "
                f"(got query dimension of {inputs.shape[1]} and candidates dimension of {self._candidates.shape[1]} "
            )
        scores, ids, _ = ops.topk_dot(inputs, self._candidates, self._ids, k)
        if testing:
            if targets is None:
                raise ValueError("Targets should be provided during the evaluation mode")
            t = targets.reshape(-1, 1).to(torch.int32)
            return Prediction(scores, (t == ids).to(torch.float32))
        return TopKPrediction(scores, ids)


class TopKOutput(Block):
    """topk.py:247-303: wraps a top-k layer ("brute-force-topk") built from candidates."""

    def __init__(self, to_call: Union[str, BruteForce] = "brute-force-topk", candidates=None, identifiers=None,
                 k: int = 10, name: Optional[str] = None):
        super().__init__(name)
        if isinstance(to_call, str):
            if to_call != "brute-force-topk":
                raise ValueError(f"unknown top-k layer {to_call!r}")
            to_call = BruteForce(k=k)
        if candidates is not None:
            to_call.index(candidates, identifiers)
        self.to_call = to_call

    def forward(self, inputs, **kwargs):
        return self.to_call.forward(inputs, **kwargs)
