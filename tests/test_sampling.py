"""Negative samplers, FIFO queue and the logQ correction: host logic + the oracle against the golden vectors
(tests/golden/reference_vectors.npz, produced by executing the reference's own torch functions)."""
from pathlib import Path

import numpy as np
import pytest
import torch

import models_amd as mm
from oracle import oracle as O

G = np.load(Path(__file__).parent / "golden" / "reference_vectors.npz")


def test_oracle_logq_before_mask_matches_reference_vectors():
    logits, _ = O.contrastive_outputs(G["sc_q"], G["sc_pos"], G["sc_neg"], G["sc_pos_id"], G["sc_neg_id"],
                                      positive_sampling_prob=G["lq_ppos"], negative_sampling_prob=G["lq_pneg"])
    np.testing.assert_allclose(logits, G["lq_logits"], atol=2e-5)


def test_oracle_popularity_post_correction_matches_reference_vectors():
    probs = G["pc_probs"]
    logits, _ = O.contrastive_outputs(G["sc_q"], G["sc_pos"], G["sc_neg"], G["sc_pos_id"], G["sc_neg_id"],
                                      post_positive_prob=probs[G["sc_pos_id"]], post_negative_prob=probs[G["sc_neg_id"]],
                                      post_reg_factor=float(G["pc_reg"]))
    np.testing.assert_allclose(logits, G["pc_logits"], atol=2e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_log_uniform_distribution_matches_reference(tag):
    mx, mn, n = (int(v) for v in G[f"pop_{tag}_args"])
    s = mm.PopularityBasedSamplerV2(max_id=mx, min_id=mn, max_num_samples=n, unique=False)
    np.testing.assert_allclose(s.sampling_dist.numpy(), G[f"pop_{tag}_dist"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(O.log_uniform_sampling_distribution(mx, mn, n, unique=False), G[f"pop_{tag}_dist"], rtol=1e-3, atol=1e-7)  # fp32 log of numpy vs torch: differences of logs cancel
    u = mm.PopularityBasedSamplerV2(max_id=mx, min_id=mn, max_num_samples=n, unique=True)
    np.testing.assert_allclose(u.sampling_dist.numpy(), O.log_uniform_sampling_distribution(mx, mn, n, unique=True), rtol=1e-3, atol=1e-6)
    p = G[f"pop_{tag}_dist"].astype(np.float64)
    np.testing.assert_allclose(u.sampling_dist.numpy(), 1.0 - (1.0 - p) ** n, rtol=1e-3, atol=1e-6)  # P(sampled at least once)


def test_popularity_sampler_like_the_reference_test():
    # tests/unit/tf/outputs/test_sampling.py:52-78
    num_classes, min_id, num_sampled = 1000, 2, 10
    s = mm.PopularityBasedSamplerV2(max_num_samples=num_sampled, max_id=num_classes - 1, min_id=min_id, seed=3)
    item_ids = torch.randint(1, num_classes, (10, 1))
    out = s(mm.Candidate(item_ids, {}))
    assert len(torch.unique(out.id)) == num_sampled and bool((out.id >= min_id).all()) and bool((out.id < num_classes - 1 + min_id).all())
    with pytest.raises(Exception) as e:
        mm.PopularityBasedSamplerV2(max_num_samples=100, max_id=49, min_id=2)
    assert "Number of items to sample `100`" in str(e.value)
    # Zipfian: low ids are drawn far more often
    s2 = mm.PopularityBasedSamplerV2(max_num_samples=2000, max_id=100_000, unique=False, seed=0)
    ids = s2.sample().id.reshape(-1)
    assert (ids < 100).float().mean() > 0.3 and (ids > 50_000).float().mean() < 0.1
    emp = torch.bincount(ids, minlength=8)[:8].float() / 2000
    np.testing.assert_allclose(emp.numpy(), s2.sampling_dist[:8].numpy(), atol=0.03)
    probs = s.with_sampling_probs(mm.Candidate(torch.tensor([[2], [5], [999]]), {})).sampling_prob
    np.testing.assert_allclose(probs.numpy(), s.sampling_dist[[2, 5, 999]].numpy())


def test_fifo_queue_semantics():
    q = mm.FIFOQueue(5, torch.int64)
    assert q.count() == 0 and q.list_all().numel() == 0
    with pytest.raises(IndexError):
        q.dequeue()
    q.enqueue_many(torch.arange(3))
    q.enqueue(torch.tensor(3))
    assert q.list_all().tolist() == [0, 1, 2, 3] and q.count() == 4
    q.enqueue_many(torch.arange(4, 8))          # wraps: the three oldest are overwritten
    assert q.list_all().tolist() == [3, 4, 5, 6, 7] and q.count() == 5 and q.at_full_capacity
    assert q.index_of(torch.tensor([6, 0])).tolist()[1] == -1 and q.storage[q.index_of(torch.tensor([6]))[0]] == 6
    assert q.dequeue().item() == 3 and q.dequeue_many(2).tolist() == [4, 5] and q.count() == 2
    q.enqueue_many(torch.arange(100, 120))      # more than the capacity: only the newest 5 stay
    assert q.list_all().tolist() == list(range(115, 120))
    e = mm.FIFOQueue(3, torch.float32, [2])
    e.enqueue_many(torch.ones(2, 2))
    e.update_by_indices(torch.tensor([0]), torch.zeros(1, 2))
    assert e.get_values_by_indices(torch.tensor([0, 1])).tolist() == [[0, 0], [1, 1]]
    with pytest.raises(AssertionError):
        e.enqueue_many(torch.ones(2, 3))
    q.clear()
    assert q.count() == 0


def test_cached_cross_batch_sampler_lags_one_batch():
    s = mm.CachedCrossBatchSampler(capacity=6)
    mk = lambda lo: mm.Candidate(torch.arange(lo, lo + 4), {}).with_embedding(torch.full((4, 3), float(lo)))
    assert s(mk(0), training=True).id.numel() == 0          # nothing cached yet; batch 0 is pending
    out = s(mk(10), training=True)                           # batch 0 available, batch 10 pending
    assert out.id.reshape(-1).tolist() == [0, 1, 2, 3] and out.embedding[:, 0].tolist() == [0.0] * 4
    out = s(mk(20), training=True)                           # capacity 6: oldest two of batch 0 dropped
    assert out.id.reshape(-1).tolist() == [2, 3, 10, 11, 12, 13]
    out = s(mk(30), training=False)                          # evaluation: the pending batch lands, nothing new queued
    assert out.id.reshape(-1).tolist() == [12, 13, 20, 21, 22, 23]


def test_contrastive_output_sampler_validation():
    with pytest.raises(ValueError):
        mm.ContrastiveOutput(None, negative_samplers="nope")
    out = mm.ContrastiveOutput(None, negative_samplers=["in-batch", mm.CachedCrossBatchSampler(8)], logq_sampling_correction=True)
    with pytest.raises(ValueError) as e:
        out.sample_negatives(mm.Candidate(torch.arange(4), {}).with_embedding(torch.zeros(4, 2)), {}, training=True)
    assert "only one negative sampler" in str(e.value)
