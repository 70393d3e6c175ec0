"""The thin `mm` names of models_amd/compat.py on the device: top-k metric classes against the reference's literals
(tests/unit/tf/metrics/test_metrics_topk.py:49-140), L2Norm, and a MultiOptimizer train step (blocks/optimizer.py:73-340)."""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import schema as S

pytestmark = pytest.mark.gpu


def test_topk_metric_classes_match_reference_literals(device):
    labels = torch.tensor([[0, 1, 0, 1, 0], [1, 0, 0, 1, 0], [0, 0, 0, 0, 1]], dtype=torch.float32, device=device)
    preds = torch.tensor([[10, 9, 8, 7, 6], [1, 4, 3, 2, 5], [10, 9, 8, 7, 6]], dtype=torch.float32, device=device)
    agg = mm.TopKMetricsAggregator(mm.RecallAt(4, pre_sorted=False), mm.PrecisionAt(4, pre_sorted=False),
                                   mm.AvgPrecisionAt(4, pre_sorted=False), mm.MRRAt(4, pre_sorted=False), mm.NDCGAt(4, pre_sorted=False))
    agg.update_state(labels, preds)
    r = agg.result()
    np.testing.assert_allclose(r["recall_at_4"], np.mean([1.0, 0.5, 0.0]), atol=1e-6)
    np.testing.assert_allclose(r["precision_at_4"], np.mean([0.5, 0.25, 0.0]), atol=1e-6)
    np.testing.assert_allclose(r["map_at_4"], np.mean([(1 / 2 + 2 / 4) / 2, (1 / 4) / 2, 0]), atol=1e-6)
    np.testing.assert_allclose(r["mrr_at_4"], np.mean([0.5, 0.25, 0.0]), atol=1e-6)
    assert 0.0 < r["ndcg_at_4"] < 1.0
    agg.update_state(labels, preds)  # streaming mean over two identical batches
    np.testing.assert_allclose(agg.result()["recall_at_4"], 0.5, atol=1e-6)
    agg.reset_state()
    assert agg.result()["recall_at_4"] == 0.0
    # pre-sorted labels (what BruteForce hands over in testing mode)
    m = mm.RecallAt(2)
    m.update_state(torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], device=device), None, label_relevant_counts=torch.ones(2, device=device))
    assert m.result() == 0.5


def test_l2norm_block_forward_backward(device):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(33, 48, generator=g)
    xt = x.clone().requires_grad_(True)
    ref = torch.nn.functional.normalize(xt, dim=-1)
    dy = torch.randn(33, 48, generator=g)
    ref.backward(dy)
    blk = mm.L2Norm()
    y = blk({"a": x.to(device)})
    np.testing.assert_allclose(y["a"].cpu().numpy(), ref.detach().numpy(), atol=1e-6)
    dx = blk.backward({"a": dy.to(device)})
    np.testing.assert_allclose(dx["a"].cpu().numpy(), xt.grad.numpy(), atol=1e-5)


def test_multi_optimizer_updates_each_block_with_its_own_rule(device):
    schema = mm.Schema([S.categorical("big", 5000), S.categorical("small", 40), S.continuous("x"), S.binary_target("y")])
    B, D = 512, 16
    g = torch.Generator().manual_seed(1)
    x = {"big": torch.randint(0, 5000, (B, 1), generator=g).to(device), "small": torch.randint(0, 40, (B, 1), generator=g).to(device),
         "x": torch.rand(B, 1, generator=g).to(device)}
    y = torch.randint(0, 2, (B, 1), generator=g).float().to(device)

    def build():
        mm.set_seed(3)
        return mm.DLRMModel(schema, embedding_dim=D, bottom_block=mm.MLPBlock([D], device=device),
                            top_block=mm.MLPBlock([16, 8], device=device), device=device)

    model = build()
    large, small = mm.split_embeddings_on_size(model.body.embeddings, 1000)
    opt = mm.MultiOptimizer([mm.OptimizerBlocks(mm.optim.SGD(0.05), large), mm.OptimizerBlocks("adagrad", small)],
                            default_optimizer=mm.optim.Adagrad(0.05))
    model.compile(optimizer=opt)
    big0 = large[0].table.data.clone()
    small0 = small[0].table.data.clone()
    l0 = float(model.train_step(x, y))
    l1 = float(model.train_step(x, y))
    assert np.isfinite(l0) and l1 < l0
    assert not torch.equal(large[0].table.data, big0) and not torch.equal(small[0].table.data, small0)
    assert "accumulator" not in large[0].table.state        # plain SGD rows: no optimizer state
    assert "accumulator" in small[0].table.state            # Adagrad rows
    # the SGD table of the multi-optimizer model moves exactly like the table of an all-SGD model in the FIRST step
    ref = build()
    ref.compile(optimizer="sgd", learning_rate=0.05)
    ref.train_step(x, y)
    model2 = build()
    l2, s2 = mm.split_embeddings_on_size(model2.body.embeddings, 1000)
    model2.compile(optimizer=mm.MultiOptimizer([mm.OptimizerBlocks(mm.optim.SGD(0.05), l2)], default_optimizer=mm.optim.Adagrad(0.05)))
    model2.train_step(x, y)
    rb = [t for t in ref.body.embeddings.feature_table.values() if t.input_dim == 5000][0]
    torch.testing.assert_close(l2[0].table.data, rb.table.data, atol=1e-6, rtol=1e-5)


def test_v1_item_retrieval_scorer_sampled_softmax_mode(device):
    """blocks/retrieval/base.py:274,313-331,400: sampled_softmax_mode scores the query against rows of the item embedding table
    (positives = lookup(targets), negatives = lookup(sampled ids)).  Here the table is handed over as `item_table`; the logits equal
    the oracle's contrastive_outputs over those rows, positives in column 0, false negatives at MIN_FLOAT."""
    from oracle import oracle as O

    torch.manual_seed(1)
    col = S.categorical("item_id", 101, [S.Tags.ITEM, S.Tags.ITEM_ID])
    table = mm.EmbeddingTable(16, col, device=device)
    sampler = mm.PopularityBasedSamplerV2(max_id=100, max_num_samples=20, min_id=1, seed=5)
    out = mm.ItemRetrievalScorer(samplers=[sampler], sampled_softmax_mode=True, item_table=table, store_negative_ids=True)
    assert out.sampled_softmax_mode and out.has_candidate_weights
    g = torch.Generator().manual_seed(2)
    q = torch.randn(50, 16, generator=g).to(device)
    tgt = torch.randint(1, 101, (50, 1), generator=g).to(device)
    pred = out({"query": q}, features={}, targets=tgt, training=True)
    nid = pred.negative_candidate_ids.cpu().numpy()
    W, t = table.table.data.cpu().numpy(), tgt.reshape(-1).cpu().numpy()
    want, _ = O.contrastive_outputs(q.cpu().numpy(), W[t], W[nid], t, nid)
    got = pred.outputs.cpu().numpy()
    assert got.shape == (50, 21) and np.array_equal(pred.targets.cpu().numpy()[:, 0], np.ones(50))
    np.testing.assert_allclose(got, want, atol=1e-4)
    hit = t[:, None] == nid[None, :]
    assert hit.any() and np.allclose(got[:, 1:][hit], mm.outputs.MIN_FLOAT)   # sampled positives are downscored (utils/constants.py:19)
    with pytest.raises(ValueError, match="item_table"):
        mm.ItemRetrievalScorer(samplers=[sampler], sampled_softmax_mode=True)
    with pytest.raises(ValueError, match="sampler"):
        mm.ItemRetrievalScorer(sampled_softmax_mode=True, item_table=table)


def test_v1_item_retrieval_scorer_downscores_by_the_named_feature_by_default(device):
    """base.py:313-316, 379-383: the ids come from features[item_id_feature_name]; no item_id_column is needed for the rescoring."""
    g = torch.Generator().manual_seed(3)
    q, c = torch.randn(8, 4, generator=g).to(device), torch.randn(8, 4, generator=g).to(device)
    ids = torch.tensor([5, 6, 5, 7, 8, 6, 9, 10], device=device).reshape(-1, 1)
    out = mm.ItemRetrievalScorer(item_id_feature_name="sku")
    assert out.downscore_false_negatives
    pred = out({"query": q, "item": c}, features={"sku": ids}, training=True)
    lg = pred.outputs.cpu().numpy()
    mf = mm.outputs.MIN_FLOAT
    assert lg.shape == (8, 9) and np.isclose(lg[0, 1 + 2], mf) and np.isclose(lg[2, 1 + 0], mf) and np.isclose(lg[1, 1 + 5], mf) and lg[0, 1 + 1] > -100
    with pytest.raises(ValueError):
        out({"query": q, "item": c}, features={}, training=True)
