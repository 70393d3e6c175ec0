"""Model heads of the kept ``mm`` surface (reference layer L4, merlin/models/tf/outputs/).

BinaryOutput (classification.py:72-123), DotProduct (base.py:291-322), ContrastiveOutput
(contrastive.py:47-405), BruteForce / TopKOutput (topk.py:129-303).
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Union

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


class NegativeSet(NamedTuple):
    """What the fused scorer needs about the negatives of one batch."""

    embedding: torch.Tensor              # [Nn, E]
    ids: Optional[torch.Tensor]          # [Nn] (None when false negatives are not rescored and nothing stores them)
    n_inbatch: int                       # leading rows that ARE the batch's own candidate embeddings (gradient flows)
    pos_logq: Optional[torch.Tensor]     # [B]  (scaled) log sampling probability of the positives, or None
    neg_logq: Optional[torch.Tensor]     # [Nn]
    logq_after_mask: bool


class ContrastiveOutput(Block):
    """contrastive.py:47-405.  training / testing -> logits [B, 1 + Nn] (positives in column 0) against the
    negatives the samplers provide (``"in-batch"``: the batch's own items, outputs/sampling/in_batch.py:25-114;
    ``PopularityBasedSamplerV2``; ``CachedCrossBatchSampler``), false negatives (incl. the diagonal) rescored to
    ``false_negative_score``, optional logQ sampling correction (``logq_sampling_correction=True`` with ONE sampler
    that knows its sampling probabilities, contrastive.py:309-319; or ``post=PopularityLogitsCorrection(...)``,
    transforms/bias.py:77-290); inference -> positive scores [B, 1].

    ``to_call`` is the item-id column / schema (two-tower: candidate embeddings come from ``inputs``) or an
    ``EmbeddingTable`` whose rows are the candidate weights (``has_candidate_weights``: positives = lookup(targets),
    sampled negatives = lookup(sampled ids))."""

    def __init__(self, to_call=None, negative_samplers="in-batch", downscore_false_negatives: bool = True,
                 false_negative_score: float = MIN_FLOAT, logits_temperature: float = 1.0,
                 store_negative_ids: bool = False, query_name: str = "query", candidate_name: str = "candidate",
                 logq_sampling_correction: bool = False, post=None, name: Optional[str] = None):
        from .inputs import EmbeddingTable
        from .sampling import PopularityLogitsCorrection, parse_negative_samplers

        super().__init__(name)
        self.negative_samplers = parse_negative_samplers(negative_samplers)
        self.logq_sampling_correction = bool(logq_sampling_correction)
        if post is not None and not isinstance(post, PopularityLogitsCorrection):
            raise NotImplementedError("only PopularityLogitsCorrection is supported as `post` of ContrastiveOutput")
        if post is not None and logq_sampling_correction:
            raise NotImplementedError("logq_sampling_correction and a PopularityLogitsCorrection post block cannot be combined")
        self.post = post
        self.candidate_table = None
        if isinstance(to_call, Schema):
            if len(to_call.select_by_tag(Tags.ITEM_ID)) == 1:
                to_call = to_call.select_by_tag(Tags.ITEM_ID).first
            elif len(to_call) == 1:
                to_call = to_call.first
            else:
                raise ValueError("to_call must be a single column schema")
        if isinstance(to_call, EmbeddingTable):
            self.candidate_table = to_call
            to_call = next(iter(to_call.features.values()))
        self.col_schema = to_call if isinstance(to_call, ColumnSchema) else None
        self.downscore_false_negatives = downscore_false_negatives
        self.false_negative_score = float(false_negative_score)
        self.logits_temperature = float(logits_temperature)
        self.store_negative_ids = store_negative_ids
        self.query_name, self.candidate_name = query_name, candidate_name

    @property
    def has_candidate_weights(self) -> bool:
        return self.candidate_table is not None

    def children(self):
        return [self.candidate_table] if self.candidate_table is not None else []

    def embedding_lookup(self, ids: torch.Tensor) -> torch.Tensor:
        return ops.embedding_gather([self.candidate_table.table.data], [ids.reshape(-1)])[:, 0]

    # ---- contrastive.py:345-401 ------------------------------------------------------------------------------------
    def sample_negatives(self, positive, features, training: bool = False, testing: bool = False):
        if self.logq_sampling_correction and len(self.negative_samplers) > 1:
            raise ValueError("It is only possible to apply logQ sampling correction "
                             "(logq_sampling_correction=True) when only one negative sampler is provided.")
        candidates = []  # (sampler, its candidates): a sampler that returned nothing (an empty cache) is skipped WITH its slot
        for sampler in self.negative_samplers:
            neg = sampler(positive, features=features, training=training, testing=testing)
            positive = sampler.with_sampling_probs(positive)
            if neg is not None and neg.id is not None and neg.id.numel() > 0:
                candidates.append((sampler, sampler.with_sampling_probs(neg)))
        if not candidates:
            raise Exception(f"No negative items where sampled from samplers {self.negative_samplers}")
        return candidates, positive

    def negatives(self, positive_embedding: torch.Tensor, positive_id: Optional[torch.Tensor], features,
                  training: bool = False, testing: bool = False) -> NegativeSet:
        """Runs the samplers and assembles one negative matrix: the in-batch candidates first (they alias the
        positives: no copy when they are the only negatives), then the sampled / cached ones."""
        from .sampling import LOGQ_EPS, Candidate, InBatchSamplerV2

        B = positive_embedding.shape[0]
        if positive_id is None:
            positive_id = torch.arange(B, device=positive_embedding.device)  # unique ids: nothing is masked
        positive = Candidate(positive_id.reshape(-1), dict(features or {})).with_embedding(positive_embedding)
        cands, positive = self.sample_negatives(positive, features, training=training, testing=testing)
        embs, ids, probs, n_in = [], [], [], 0
        for sampler, c in cands:
            if isinstance(sampler, InBatchSamplerV2):
                if embs:
                    raise ValueError("the in-batch sampler must come first in negative_samplers")
                n_in = B
            elif not c.has_embedding:
                if not self.has_candidate_weights:
                    raise ValueError("Negative candidate must have an embedding")
                c = c.with_embedding(self.embedding_lookup(c.id))
            embs.append(c.embedding)
            ids.append(c.id.reshape(-1).to(positive_id.dtype))
            probs.append(c.sampling_prob)
        emb = embs[0] if len(embs) == 1 else torch.cat(embs, 0)
        nid = ids[0] if len(ids) == 1 else torch.cat(ids, 0)
        pos_logq = neg_logq = None
        after = False
        if self.logq_sampling_correction:
            if positive.sampling_prob is None or probs[0] is None:
                raise ValueError("The logQ sampling correction is enabled, but sampling probs were not found "
                                 "for both positive and negative candidates")
            pos_logq = torch.log(positive.sampling_prob.reshape(-1).float() + LOGQ_EPS)
            neg_logq = torch.log(probs[0].reshape(-1).float() + LOGQ_EPS)
        elif self.post is not None and training:  # the post block only acts in training (bias.py:223-236)
            pos_logq, neg_logq, after = self.post.logq(positive_id), self.post.logq(nid), True
        need_ids = self.downscore_false_negatives or self.store_negative_ids
        return NegativeSet(emb, nid if need_ids else None, n_in, pos_logq, neg_logq, after)

    def forward(self, inputs, features=None, targets=None, training: bool = False, testing: bool = False,
                materialize: bool = True):
        q = inputs[self.query_name] if isinstance(inputs, dict) else inputs
        if not (training or testing):
            if self.has_candidate_weights:
                raise NotImplementedError("inference over all candidate weights: use to_top_k_encoder / BruteForce")
            return ops.rowwise_dot(q, inputs[self.candidate_name])  # contrastive.py:221 -> DotProduct
        if self.has_candidate_weights:
            if targets is None:
                raise ValueError("ContrastiveOutput over an EmbeddingTable needs the positive ids as `targets`")
            pos_id = (targets[self.col_schema.name] if isinstance(targets, dict) else targets).reshape(-1)
            c = self.embedding_lookup(pos_id)
        else:
            c = inputs[self.candidate_name]
            pos_id = None
            if self.downscore_false_negatives or self.store_negative_ids or self.post is not None or self.logq_sampling_correction:
                if features is None or self.col_schema is None or self.col_schema.name not in features:
                    raise ValueError("downscore_false_negatives needs the item-id feature in `features`")
                pos_id = features[self.col_schema.name].reshape(-1)
        neg = self.negatives(c, pos_id, features, training=training, testing=testing)
        mask_ids = (pos_id, neg.ids) if self.downscore_false_negatives else (None, None)
        self._q, self._c, self._ids, self._neg = q, c, pos_id, neg
        res = ops.inbatch_softmax(q, c, neg.embedding, mask_ids[0], mask_ids[1], self.logits_temperature,
                                  self.false_negative_score, materialize=materialize, pos_logq=neg.pos_logq,
                                  neg_logq=neg.neg_logq, logq_after_mask=neg.logq_after_mask)
        self._lse = res.lse
        self.last_loss = res.loss
        if not materialize:
            return res
        tg = torch.zeros_like(res.logits)
        tg[:, 0] = 1.0
        return Prediction(res.logits, tg, neg.ids if self.store_negative_ids else None)


class BruteForce(Block):
    """topk.py:129-237: brute-force top-k over an indexed candidate matrix."""

    def __init__(self, k: int = 10, name: Optional[str] = None):
        super().__init__(name)
        self._k = int(k)
        self._candidates: Optional[torch.Tensor] = None
        self._ids: Optional[torch.Tensor] = None
        self._split = None

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
        # the bf16 (hi, lo) split of the catalogue, built once here: the filter stages of the call then run on the bf16 matrix
        # pipe while scores and indices stay bit-identical to the fp32 pipeline (ops.TopKSplit; MERLIN_HIP_TOPK=f32 turns it off)
        self._split = None
        if self._candidates.is_cuda and ops.TopKSplit.supported(self._candidates.shape[1]) and ops.topk_mode() == "split":
            self._split = ops.TopKSplit(self._candidates)
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
        scores, ids, _ = ops.topk_dot(inputs, self._candidates, self._ids, k, split=getattr(self, "_split", None))
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
