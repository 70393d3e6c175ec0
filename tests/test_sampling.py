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


def test_popularity_sampler_validation_and_probabilities():
    # tests/unit/tf/outputs/test_sampling.py:52-78 (construction errors, sampling probabilities of given ids)
    with pytest.raises(Exception) as e:
        mm.PopularityBasedSamplerV2(max_num_samples=100, max_id=49, min_id=2)
    assert "Number of items to sample `100`" in str(e.value)
    with pytest.raises(ValueError):  # unique draws come from [0, max_id - min_id): more than that can never be returned
        mm.PopularityBasedSamplerV2(max_num_samples=48, max_id=49, min_id=2, unique=True)
    s = mm.PopularityBasedSamplerV2(max_num_samples=10, max_id=999, min_id=2, seed=3)
    probs = s.with_sampling_probs(mm.Candidate(torch.tensor([[2], [5], [999]]), {})).sampling_prob
    np.testing.assert_allclose(probs.numpy(), s.sampling_dist[[2, 5, 999]].numpy())
    with pytest.raises(RuntimeError):  # the draw itself is a HIP kernel: no CPU path
        s.sample(device="cpu")


def test_oracle_philox_known_answers_and_log_uniform_sampler():
    """Philox4x32-10 against the published known-answer vectors (Random123 kat_vectors), then the sampler statement:
    the empirical distribution follows get_sampling_distribution, unique draws are distinct and in first-appearance order."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want
    d = O.log_uniform_draws(0, 200_000, 0, 7, 100_000)
    p = O.log_uniform_sampling_distribution(100_000, 0, 10, unique=False)
    np.testing.assert_allclose(np.bincount(d, minlength=8)[:8] / 200_000, p[:8], atol=2e-3)
    assert d.min() >= 0 and d.max() < 100_000
    u = O.log_uniform_sample(1000, 300, True, seed=3, call=0, min_id=2)
    assert len(set(u.tolist())) == 300 and u.min() >= 2 and u.max() < 1002
    draws = (O.log_uniform_draws(0, 4096, 0, 3, 1000) + 2).tolist()
    first = list(dict.fromkeys(draws))[:300]
    assert u.tolist() == first
    assert not np.array_equal(u, O.log_uniform_sample(1000, 300, True, seed=3, call=1, min_id=2))  # a new call, a new sample


@pytest.mark.gpu
@pytest.mark.parametrize("range_max,min_id,n,unique", [(1000, 2, 10, True), (1000, 0, 999, True), (1000, 0, 1000, True),
                                                        (100_000, 5, 5000, True), (100_000, 0, 20_000, False),
                                                        (7, 3, 7, True), (1, 0, 1, True), (10_000_000, 0, 3000, True)])
def test_hip_log_uniform_sampler_equals_the_oracle(device, range_max, min_id, n, unique):
    """mh_log_uniform_sample == the sequential statement, id for id (integer work: bit-exact), over three consecutive
    calls (the device-side call counter advances by itself)."""
    from models_amd import ops

    seed = 1234 + n
    st = torch.tensor([seed, 0], dtype=torch.int64, device=device)
    for call in range(3):
        got = ops.log_uniform_sample(range_max, n, unique, st, min_id).cpu().numpy()
        np.testing.assert_array_equal(got, O.log_uniform_sample(range_max, n, unique, seed, call, min_id))
    assert st.tolist() == [seed, 3]


@pytest.mark.gpu
def test_popularity_sampler_on_the_device(device):
    # tests/unit/tf/outputs/test_sampling.py:52-78
    num_classes, min_id, num_sampled = 1000, 2, 10
    s = mm.PopularityBasedSamplerV2(max_num_samples=num_sampled, max_id=num_classes - 1, min_id=min_id, seed=3)
    item_ids = torch.randint(1, num_classes, (10, 1), device=device)
    out = s(mm.Candidate(item_ids, {}))
    assert out.id.device.type == "cuda" and out.id.shape == (num_sampled, 1)
    assert len(torch.unique(out.id)) == num_sampled and bool((out.id >= min_id).all()) and bool((out.id < num_classes - 1 + min_id).all())
    out2 = s(mm.Candidate(item_ids, {}))
    assert not torch.equal(out.id, out2.id)  # the call counter is device state
    s.check_status()  # the kernel's status word (what Model.fit reads at epoch ends): the unique draw completed
    from models_amd import ops

    assert ops.log_uniform_sample_status(device) == 0
    # Zipfian: low ids are drawn far more often
    s2 = mm.PopularityBasedSamplerV2(max_num_samples=2000, max_id=100_000, unique=False, seed=0)
    ids = s2.sample(device=device).id.reshape(-1)
    assert (ids < 100).float().mean() > 0.3 and (ids > 50_000).float().mean() < 0.1
    emp = torch.bincount(ids, minlength=8)[:8].float().cpu() / 2000
    np.testing.assert_allclose(emp.numpy(), s2.sampling_dist[:8].numpy(), atol=0.03)
    # replayed from a hipGraph: every replay is a new sample
    s3 = mm.PopularityBasedSamplerV2(max_num_samples=64, max_id=5000, seed=11)
    s3.sample(device=device)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        c = s3.sample(device=device)
    g.replay()
    a = c.id.clone()
    g.replay()
    assert not torch.equal(a, c.id) and len(torch.unique(c.id)) == 64


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
    # ring arithmetic against a plain Python list over many random operations (wraps, overfills, partial drains)
    import random

    rnd = random.Random(5)
    q, model, nxt = mm.FIFOQueue(7, torch.int64), [], 0
    for _ in range(300):
        op = rnd.random()
        if op < 0.55:
            n = rnd.randint(1, 11)
            q.enqueue_many(torch.arange(nxt, nxt + n))
            model = (model + list(range(nxt, nxt + n)))[-7:]
            nxt += n
        elif op < 0.8 and model:
            n = rnd.randint(1, 9)
            assert q.dequeue_many(n).tolist() == model[:n]
            model = model[n:]
        elif model:
            assert q.dequeue().item() == model.pop(0)
        assert q.list_all().tolist() == model and q.count() == len(model) and q.at_full_capacity == (len(model) == 7)


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
