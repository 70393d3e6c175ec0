"""Randomised (hypothesis) properties of the CPU oracle itself: the restatement must agree with independent torch
formulations and with the invariants the reference's tests assert, on shapes the fixed fixtures do not cover."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import oracle as O

SET = settings(max_examples=25, deadline=None)


@SET
@given(st.integers(1, 9), st.integers(2, 12), st.integers(1, 5), st.integers(0, 2**31 - 1))
def test_dot_interaction_is_the_strict_upper_triangle_in_row_major_order(B, F, D4, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, F, 4 * D4)).astype(np.float32)
    got = O.dot_interaction(X)
    want = [[float(X[b, i] @ X[b, j]) for i in range(F) for j in range(i + 1, F)] for b in range(B)]
    np.testing.assert_allclose(got, np.array(want, np.float32).reshape(B, -1), rtol=1e-5, atol=1e-5)


@SET
@given(st.integers(2, 40), st.integers(1, 6), st.integers(1, 8), st.integers(0, 2**31 - 1))
def test_inbatch_logits_mask_exactly_the_duplicates_of_the_positive(B, E4, n_ids, seed):
    """tests/unit/tf/outputs/test_contrastive.py:173-206 generalised: column 1 + j of row b is the false-negative score
    iff item j carries row b's item id (always true on the diagonal); everything else is the plain dot product."""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((B, 4 * E4)).astype(np.float32)
    it = rng.standard_normal((B, 4 * E4)).astype(np.float32)
    ids = rng.integers(0, n_ids, size=B)
    logits, targets = O.contrastive_outputs(q, it, it, ids, ids)
    assert logits.shape == (B, B + 1) and targets[:, 0].all() and not targets[:, 1:].any()
    np.testing.assert_allclose(logits[:, 0], (q * it).sum(-1), rtol=1e-5, atol=1e-5)
    mask = ids[:, None] == ids[None, :]
    assert np.all(logits[:, 1:][mask] == np.float32(O.MIN_FLOAT))
    np.testing.assert_allclose(logits[:, 1:][~mask], (q @ it.T)[~mask], rtol=1e-4, atol=1e-5)


@SET
@given(st.integers(1, 12), st.integers(1, 60), st.integers(1, 20), st.integers(0, 2**31 - 1))
def test_top_k_is_the_stable_descending_sort(B, N, k, seed):
    rng = np.random.default_rng(seed)
    k = min(k, N)
    scores = rng.integers(-3, 4, size=(B, N)).astype(np.float32)  # many ties
    vals, idx = O.top_k(scores, k)
    want = torch.sort(torch.from_numpy(scores), dim=1, descending=True, stable=True)
    np.testing.assert_array_equal(idx, want.indices[:, :k].numpy())
    np.testing.assert_array_equal(vals, want.values[:, :k].numpy())


@SET
@given(st.integers(1, 10), st.integers(1, 30), st.sampled_from(["sum", "mean", "sqrtn"]), st.integers(0, 2**31 - 1))
def test_embedding_bag_equals_torch_embedding_bag_semantics(B, V, combiner, seed):
    rng = np.random.default_rng(seed)
    D = 4
    W = rng.standard_normal((V, D)).astype(np.float32)
    lens = rng.integers(0, 5, size=B)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.integers(0, V, size=int(offsets[-1])).astype(np.int64)
    got = O.embedding_bag(W, values, offsets, combiner)
    want = np.zeros((B, D), np.float32)
    for b in range(B):
        rows = W[values[offsets[b]:offsets[b + 1]]]
        if len(rows):
            s = rows.sum(0)
            want[b] = s if combiner == "sum" else (s / len(rows) if combiner == "mean" else s / np.sqrt(np.float32(len(rows))))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


@SET
@given(st.integers(1, 50), st.integers(0, 2**31 - 1))
def test_bce_and_softmax_ce_match_torch(B, seed):
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.0, 1.0, size=(B, 1)).astype(np.float32)
    y = rng.integers(0, 2, size=(B, 1)).astype(np.float32)
    want = torch.nn.functional.binary_cross_entropy(torch.from_numpy(p).clamp(1e-7, 1 - 1e-7), torch.from_numpy(y), reduction="none")
    np.testing.assert_allclose(O.binary_crossentropy(p, y).reshape(-1), want.numpy().reshape(-1), rtol=1e-5, atol=1e-6)
    logits = rng.standard_normal((B, 7)).astype(np.float32) * 3
    loss, lse = O.softmax_ce_first_column(logits)
    want = torch.nn.functional.cross_entropy(torch.from_numpy(logits), torch.zeros(B, dtype=torch.long), reduction="none")
    np.testing.assert_allclose(loss, want.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lse, torch.logsumexp(torch.from_numpy(logits), 1).numpy(), rtol=1e-5, atol=1e-5)


def test_dropout_and_batchnorm_statements():
    """Counter-based dropout: keep fraction 1 - rate, kept values scaled by 1 / (1 - rate), another call another mask;
    batch norm (Keras non-fused 2-D semantics) against torch's functional batch_norm incl. the moving statistics."""
    import torch

    x = np.random.default_rng(0).normal(size=(4000, 16)).astype(np.float32)
    y0, y1 = O.dropout(x, 0.3, seed=9, call=0), O.dropout(x, 0.3, seed=9, call=1)
    assert abs((y0 != 0).mean() - 0.7) < 0.01 and not np.array_equal(y0 != 0, y1 != 0)
    np.testing.assert_allclose(y0[y0 != 0], (x * O.dropout_scale(0.3))[y0 != 0], rtol=1e-7)
    g, b = np.linspace(0.5, 1.5, 16).astype(np.float32), np.linspace(-0.1, 0.1, 16).astype(np.float32)
    y, mm_, mv_ = O.batchnorm_train(x, g, b, np.zeros(16, np.float32), np.ones(16, np.float32))
    rm, rv = torch.zeros(16), torch.ones(16)
    yt = torch.nn.functional.batch_norm(torch.from_numpy(x), rm, rv, torch.from_numpy(g), torch.from_numpy(b), True, 0.01, 1e-3)
    np.testing.assert_allclose(y, yt.numpy(), atol=2e-5)
    np.testing.assert_allclose(mm_, rm.numpy(), atol=1e-6)
    # torch's running variance is the UNBIASED batch variance, Keras' non-fused layer keeps the biased one
    np.testing.assert_allclose(mv_, 0.99 + 0.01 * x.astype(np.float64).var(0), atol=1e-6)
    np.testing.assert_allclose(O.batchnorm_infer(x, g, b, mm_, mv_),
                               torch.nn.functional.batch_norm(torch.from_numpy(x), torch.from_numpy(mm_), torch.from_numpy(mv_),
                                                              torch.from_numpy(g), torch.from_numpy(b), False, 0.0, 1e-3).numpy(), atol=2e-5)
