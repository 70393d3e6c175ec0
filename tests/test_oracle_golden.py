"""Pin the oracle against the reference's own known-answer tests (no GPU)."""
import numpy as np
import pytest

from oracle import cbind, oracle as O


def test_min_float():
    # merlin/models/utils/constants.py:19
    assert O.MIN_FLOAT == pytest.approx(-655.04)


def test_contrastive_without_downscore():
    # tests/unit/torch/outputs/test_constrastive.py:31-47
    q = np.array([[0.1, 0.2], [0.3, 0.4]], np.float32)
    p = np.array([[0.5, 0.6], [0.7, 0.8]], np.float32)
    n = np.array([[0.9, 1.0], [1.1, 1.2], [1.3, 1.4]], np.float32)
    out, tgt = O.contrastive_outputs(q, p, n, downscore_false_negatives=False)
    exp = np.array([[0.17, 0.29, 0.35, 0.41], [0.53, 0.67, 0.81, 0.95]], np.float32)
    np.testing.assert_allclose(out, exp, atol=1e-4)
    np.testing.assert_array_equal(tgt, [[1, 0, 0, 0], [1, 0, 0, 0]])


def test_contrastive_rescore_false_negatives():
    # tests/unit/torch/outputs/test_constrastive.py:49-73
    q = np.array([[0.1, 0.2]], np.float32)
    p = np.array([[0.5, 0.6]], np.float32)
    n = np.array([[0.5, 0.6], [0.9, 1.0]], np.float32)
    out, tgt = O.contrastive_outputs(q, p, n, np.array([[0]]), np.array([[0, 1]]), false_negative_score=-100.0)
    np.testing.assert_allclose(out, [[0.17, -100.0, 0.29]], atol=1e-4)
    np.testing.assert_array_equal(tgt, [[1, 0, 0]])


def test_inbatch_diagonal_masked():
    # tests/unit/tf/outputs/test_contrastive.py:173-206: diag == false_negative_score, off-diag != it
    rng = np.random.default_rng(0)
    B, E = 16, 8
    q, it = rng.normal(size=(B, E)).astype("f"), rng.normal(size=(B, E)).astype("f")
    ids = np.arange(B)
    out, _ = O.contrastive_outputs(q, it, it, ids, ids)
    neg = out[:, 1:]
    assert np.all(np.diag(neg) == np.float32(O.MIN_FLOAT))
    off = neg[~np.eye(B, dtype=bool)]
    assert np.all(off != np.float32(O.MIN_FLOAT))
    assert out.shape == (B, B + 1)


def test_extract_topk_known_answers():
    # tests/unit/tf/utils/test_tf_utils.py:42-75
    labels = np.array([[0, 1, 0, 1, 0, 0, 1, 0, 0, 0], [1, 0, 0, 1, 0, 0, 0, 0, 0, 0], [0, 1, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    preds = np.array([[1, 2, 3, 4, 5, 6, 7, 8, 9, 10], [1, 3, 5, 7, 9, 2, 4, 6, 8, 10], [1] * 10], np.float32)
    v, l, c = O.extract_topk(5, preds, labels)
    np.testing.assert_array_equal(v, [[10, 9, 8, 7, 6], [10, 9, 8, 7, 6], [1, 1, 1, 1, 1]])
    np.testing.assert_array_equal(l, [[0, 0, 0, 1, 0], [0, 0, 0, 1, 0], [0, 1, 0, 0, 1]])
    np.testing.assert_array_equal(c, [3, 2, 3])


def test_extract_topk_ties_keep_index_order():
    # tests/unit/tf/utils/test_tf_utils.py:78-98 (shuffle_ties=False branch)
    labels = np.array([[0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0]], np.float32)
    preds = np.full((1, 20), 10.0, np.float32)
    _, l, c = O.extract_topk(10, preds, labels)
    np.testing.assert_array_equal(l, labels[:, :10])
    assert c[0] == 5


def test_c_topk_matches_numpy_tie_rule():
    rng = np.random.default_rng(1)
    S = rng.integers(0, 5, size=(7, 50)).astype(np.float32)  # many ties
    v0, i0 = O.top_k(S, 9)
    v1, i1 = cbind.topk_rows(S, 9)
    np.testing.assert_array_equal(v0, v1)
    np.testing.assert_array_equal(i0, i1)


def test_topk_recommender_equals_matmul_topk_gather():
    # tests/unit/tf/core/test_index.py:76-121 (ids = gather(identifiers, top_k(q C^T)))
    rng = np.random.default_rng(2)
    q, c = rng.normal(size=(5, 8)).astype("f"), rng.normal(size=(40, 8)).astype("f")
    ids = rng.permutation(1000)[:40].astype(np.int32)
    vals, out_ids, idx = O.brute_force_topk(q, c, ids, 7)
    s = q @ c.T
    np.testing.assert_array_equal(out_ids, ids[np.argsort(-s, axis=1, kind="stable")[:, :7]])
    assert out_ids.dtype == np.int32  # tests/unit/tf/core/test_encoder.py:125-132


def test_dot_interaction_shape_and_order():
    # width F(F-1)/2 : tests/unit/tf/blocks/test_dlrm.py:26-38; order = row-major i<j
    rng = np.random.default_rng(3)
    X = rng.normal(size=(4, 5, 8)).astype("f")
    out = O.dot_interaction(X)
    assert out.shape == (4, 10)
    pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)]
    for p, (i, j) in enumerate(pairs):
        np.testing.assert_allclose(out[:, p], (X[:, i] * X[:, j]).sum(-1), rtol=1e-5, atol=1e-6)
    top_in = O.dlrm_interaction_concat(X, X[:, -1])
    assert top_in.shape == (4, 10 + 8)  # F(F-1)/2 + D
    # [bottom_block | interactions]: tf/core/combinators.py:564-569 (dict-valued branches merged by update) +
    # tf/core/aggregation.py:54-66 (sorted keys); pinned by the dl_* fixtures (test_golden_vectors.py)
    np.testing.assert_array_equal(top_in[:, :8], X[:, -1])
    np.testing.assert_array_equal(top_in[:, 8:], out)


def test_l2norm_unit_rows():
    # tests/unit/tf/blocks/retrieval/test_two_tower.py:94-108
    rng = np.random.default_rng(4)
    x = rng.normal(size=(6, 12)).astype("f")
    np.testing.assert_allclose(np.linalg.norm(O.l2norm(x), axis=-1), 1.0, atol=1e-6)


def test_bag_combiners_and_edge_cases():
    W = np.arange(20, dtype=np.float32).reshape(5, 4)
    values = np.array([1, 2, -1, 4, 4], np.int64)
    offsets = np.array([0, 2, 2, 3, 5], np.int64)  # bag1 empty, bag2 only a pruned id
    s = O.embedding_bag(W, values, offsets, "sum")
    np.testing.assert_array_equal(s[0], W[1] + W[2])
    np.testing.assert_array_equal(s[1], 0)
    np.testing.assert_array_equal(s[2], 0)
    m = O.embedding_bag(W, values, offsets, "mean")
    np.testing.assert_allclose(m[3], W[4])
    q = O.embedding_bag(W, values, offsets, "sqrtn")
    np.testing.assert_allclose(q[0], (W[1] + W[2]) / np.sqrt(2), rtol=1e-6)


def test_fmaf_gemm_close_to_blas():
    rng = np.random.default_rng(5)
    a, b = rng.normal(size=(9, 130)).astype("f"), rng.normal(size=(11, 130)).astype("f")
    np.testing.assert_allclose(cbind.gemm_nt_fmaf(a, b), a @ b.T, atol=1e-4)
    w = rng.normal(size=(130, 6)).astype("f")
    np.testing.assert_allclose(cbind.gemm_nn_fmaf(a, w), a @ w, atol=1e-4)


def test_topk_metrics_known_answers():
    # tests/unit/tf/metrics/test_metrics_topk.py:49-140: fixture (labels sorted by the predictions) + literals
    labels = np.array([[0, 1, 0, 1, 0], [1, 0, 0, 1, 0], [0, 0, 0, 0, 1]], np.float32)
    preds = np.array([[10, 9, 8, 7, 6], [1, 4, 3, 2, 5], [10, 9, 8, 7, 6]], np.float32)
    _, y, cnt = O.extract_topk(5, preds, labels)
    np.testing.assert_array_equal(cnt, [2, 2, 1])
    dcg_probe = lambda pos: 1.0 / np.log2(pos + 1)
    np.testing.assert_allclose(O.recall_at(y, cnt, 4), [2 / 2, 1 / 2, 0 / 1])
    np.testing.assert_allclose(O.precision_at(y, 4), [2 / 4, 1 / 4, 0])
    np.testing.assert_allclose(O.average_precision_at(y, cnt, 4), [(1 / 2 + 2 / 4) / 2, (1 / 4) / 2, 0])
    np.testing.assert_allclose(O.dcg_at(y, 4), [dcg_probe(2) + dcg_probe(4), dcg_probe(4), 0])
    ideal = dcg_probe(1) + dcg_probe(2)
    np.testing.assert_allclose(O.ndcg_at(y, cnt, 4), [(dcg_probe(2) + dcg_probe(4)) / ideal, dcg_probe(4) / ideal, 0])
    np.testing.assert_allclose(O.mrr_at(y, 4), [1 / 2, 1 / 4, 0])
    from sklearn.metrics import ndcg_score

    ref = ndcg_score(labels, preds, k=4, ignore_ties=True)  # the reference cross-checks against sklearn too
    np.testing.assert_allclose(O.ndcg_at(y, cnt, 4).mean(), ref, atol=1e-6)


def test_bag_sum_mean_match_the_reference_torch_embedding_bag():
    """Ragged lookup with sum / mean combiners pinned to the outputs of the reference torch backend's own
    EmbeddingTable.forward_bag (F.embedding_bag, torch/inputs/embedding.py:264-293; vectors by make_golden.py)."""
    from pathlib import Path

    import numpy as np

    from oracle import oracle as O

    G = np.load(Path(__file__).parent / "golden" / "reference_vectors.npz")
    for mode in ("sum", "mean"):
        got = O.embedding_bag(G["bag_W"], G["bag_values"], G["bag_offsets"], mode)
        np.testing.assert_allclose(got, G[f"bag_{mode}"], atol=1e-6)
