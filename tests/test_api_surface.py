"""Construction-time behaviour of the kept mm surface (no GPU): shapes, sharing, inferred dims, errors.
Mirrors the reference's tests/unit/tf/inputs/test_embedding.py and tests/unit/tf/blocks/test_dlrm.py."""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import schema as S

CPU = torch.device("cpu")


def test_embedding_table_shapes_and_domain():
    # input_dim = int_domain.max + 1 (inputs/embedding.py:91-93)
    col = S.categorical("item_id", 1000)
    t = mm.EmbeddingTable(16, col, device=CPU)
    assert t.table.shape == (1000, 16) and t.input_dim == 1000
    assert abs(float(t.table.data.mean())) < 5e-3 and float(t.table.data.abs().max()) <= 0.05  # uniform(-0.05, 0.05)
    with pytest.raises(ValueError):
        mm.EmbeddingTable(16, S.ColumnSchema("no_domain"), device=CPU)
    with pytest.raises(ValueError):
        mm.EmbeddingTable(15, col, device=CPU)  # HIP path needs dim % 4 == 0


def test_shared_table_by_domain_name():
    # tests/unit/tf/inputs/test_embedding.py:231-253, 746-781: columns with one int_domain.name share a table
    a = S.categorical("item_id", 100, domain_name="item")
    b = S.categorical("last_item", 100, domain_name="item")
    c = S.categorical("city", 7)
    emb = mm.Embeddings(mm.Schema([a, b, c]), dim=8, device=CPU)
    assert sorted(emb.parallel_layers) == ["city", "item"]
    assert emb.feature_table["item_id"] is emb.feature_table["last_item"]
    with pytest.raises(ValueError):
        emb.feature_table["item_id"].add_feature(S.categorical("other", 5))  # different domain size


def test_inferred_dims_and_per_feature_dims():
    # tests/unit/tf/inputs/test_embedding.py:485-588
    sch = mm.Schema([S.categorical("a", 10), S.categorical("b", 100_000), S.categorical("c", 1_000_000)])
    emb = mm.Embeddings(sch, device=CPU)
    assert [emb.feature_table[n].dim for n in "abc"] == [8, 40, 64]
    emb = mm.Embeddings(sch, dim={"a": 16}, device=CPU)
    assert emb.feature_table["a"].dim == 16 and emb.feature_table["b"].dim == 40
    v1 = mm.EmbeddingFeatures.from_schema(sch, device=CPU)
    assert {emb_.dim for emb_ in v1.parallel_layers.values()} == {64}  # V1 default dim 64
    w = v1.feature_table["a"].table.data
    assert float(w.abs().max()) <= 0.1 + 1e-6  # truncated normal, 2 sigma


def test_pretrained_table_and_trainable_flag():
    # tests/unit/tf/inputs/test_embedding.py:203-229
    w = np.random.default_rng(0).normal(size=(20, 8)).astype(np.float32)
    t = mm.EmbeddingTable.from_pretrained(w, trainable=False, name="pre", device=CPU)
    np.testing.assert_array_equal(t.table.numpy(), w)
    assert t.table.trainable is False and t.table.sparse


def test_mlp_block_construction_errors():
    with pytest.raises(ValueError, match="mismatch"):
        mm.MLPBlock([8, 4], activation=["relu"])
    # Keras activations beyond relu / sigmoid / linear: a linear Dense followed by an element-wise Activation layer
    t = mm.MLPBlock([8, 4], activation="tanh", device=CPU)
    assert [type(l).__name__ for l in t.layers] == ["_Dense", "Activation", "_Dense", "Activation"]
    assert [l.activation for l in t.layers] == [None, "tanh", None, "tanh"]
    with pytest.raises(ValueError, match="Unknown activation"):
        mm.MLPBlock([8], activation="not_an_activation", device=CPU)
    blk = mm.MLPBlock([8, 4], no_activation_last_layer=True, device=CPU)
    assert [l.activation for l in blk.layers] == ["relu", None]


def test_sorted_key_orders():
    sch = mm.Schema([S.categorical(f"C{i}", 10) for i in range(1, 13)] + [S.continuous("I1")])
    blk = mm.DLRMBlock(sch, embedding_dim=8, bottom_block=mm.MLPBlock([8], device=CPU), device=CPU)
    # ASCII sort: "C1" < "C10" < "C11" < "C12" < "C2" ... < "bottom_block" (upper-case before lower-case)
    assert blk.stack_order[:5] == ["C1", "C10", "C11", "C12", "C2"] and blk.stack_order[-1] == "bottom_block"


def test_two_tower_requires_user_and_item_tags():
    sch = mm.Schema([S.categorical("x", 10)])
    with pytest.raises(ValueError, match="user.*item"):
        mm.TwoTowerModel(sch, mm.MLPBlock([8], device=CPU), device=CPU)


def test_contrastive_output_samplers():
    col = S.categorical("item_id", 10, [S.Tags.ITEM_ID])
    with pytest.raises(ValueError):  # unregistered sampler name
        mm.ContrastiveOutput(col, negative_samplers="popularity")
    out = mm.ContrastiveOutput(col, negative_samplers=["in-batch", mm.PopularityBasedSamplerV2(max_id=9, max_num_samples=3)])
    assert [type(s).__name__ for s in out.negative_samplers] == ["InBatchSamplerV2", "PopularityBasedSamplerV2"]
    with pytest.raises(NotImplementedError):  # only the popularity logQ block is a supported `post`
        mm.ContrastiveOutput(col, post=mm.MLPBlock([4]))


def test_prepare_features_ragged_contract():
    from models_amd.models import prepare_features

    x = {"a": torch.arange(4), "l__values": torch.arange(5), "l__offsets": torch.tensor([0, 2, 5])}
    out = prepare_features(x)
    assert out["a"].shape == (4, 1)
    assert isinstance(out["l"], mm.Ragged) and "l__values" not in out


def test_encoder_and_two_tower_v2_signature():
    """mm.Encoder(schema, *blocks) + mm.TwoTowerModelV2(query, candidate, ...) (models/retrieval.py:409-486)."""
    from models_amd import schema as S

    schema = mm.Schema([S.categorical("user_id", 500, [S.Tags.USER, S.Tags.USER_ID]),
                        S.categorical("item_id", 300, [S.Tags.ITEM, S.Tags.ITEM_ID]),
                        S.categorical("item_cat", 12, [S.Tags.ITEM])])
    q = mm.Encoder(schema.select_by_tag(S.Tags.USER), mm.MLPBlock([16], device="cpu"), device="cpu")
    c = mm.Encoder(schema.select_by_tag(S.Tags.ITEM), mm.MLPBlock([32, 16], device="cpu"), device="cpu")
    assert [type(l).__name__ for l in c.layers] == ["InputBlockV2", "_Dense", "_Dense"]
    assert c.schema.column_names == ["item_id", "item_cat"]
    model = mm.TwoTowerModelV2(q, c, schema=schema, logits_temperature=0.5)
    assert isinstance(model, mm.RetrievalModel) and model.output.col_schema.name == "item_id"
    assert model.output.logits_temperature == 0.5 and model.body.parallel_layers["item"] is c
    with pytest.raises(ValueError):
        mm.TwoTowerModelV2(mm.MLPBlock([8], device="cpu"), c)
    with pytest.raises(ValueError):  # not a registered sampler name
        mm.TwoTowerModelV2(q, c, negative_samplers=["popularity"])


def test_save_and_load_weights_round_trip(tmp_path):
    """Checkpoint / resume: parameters + optimizer state survive save_weights -> load_weights (CPU tensors here;
    the same code moves device tensors)."""
    import torch
    from models_amd import schema as S

    schema = mm.Schema([S.categorical("a", 50), S.categorical("b", 20), S.continuous("x"), S.binary_target("y")])

    def build(seed):
        m = mm.DLRMModel(schema, embedding_dim=8, bottom_block=mm.MLPBlock([8], device="cpu", seed=seed),
                         top_block=mm.MLPBlock([8, 4], device="cpu", seed=seed + 1), device="cpu")
        m.compile(optimizer="adagrad", learning_rate=0.1)
        m.body.bottom_block.layers[0].build(1)
        m.body.top_block.layers[0].build(3 + 8)
        m.body.top_block.layers[1].build(8)
        m.output.to_call.build(4)
        return m

    a, b = build(1), build(7)
    for i, p in enumerate(a.parameters()):
        p.state["accumulator"] = torch.full_like(p.data, 0.1 + i)
    assert any(not torch.equal(pa.data, pb.data) for pa, pb in zip(a.parameters(), b.parameters()))
    a.save_weights(tmp_path / "ckpt")
    b.load_weights(tmp_path / "ckpt")
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa.data, pb.data) and torch.equal(pa.state["accumulator"], pb.state["accumulator"])
    c = mm.DLRMModel(schema, embedding_dim=8, bottom_block=mm.MLPBlock([8], device="cpu"), device="cpu")
    with pytest.raises(ValueError):
        c.load_weights(tmp_path / "ckpt")


def test_untagged_checkpoint_of_the_old_dlrm_layout_is_rotated_not_misapplied(tmp_path):
    """Checkpoints carry a layout tag; a file without one was written when the top MLP's input was [interactions | bottom]:
    with legacy_layout="interactions_first" the first top-MLP kernel's last D rows (and its optimizer state) move to the front on
    load; without the argument the untagged file is refused; a file with an unknown tag is refused."""
    import numpy as np
    import torch
    from models_amd import schema as S
    from models_amd.models import CHECKPOINT_FORMAT

    schema = mm.Schema([S.categorical("a", 50), S.categorical("b", 20), S.continuous("x"), S.binary_target("y")])
    D = 8

    def build(seed):
        m = mm.DLRMModel(schema, embedding_dim=D, bottom_block=mm.MLPBlock([D], device="cpu", seed=seed),
                         top_block=mm.MLPBlock([8, 4], device="cpu", seed=seed + 1), device="cpu")
        m.compile(optimizer="adagrad", learning_rate=0.1)
        m.body.bottom_block.layers[0].build(1)
        m.body.top_block.layers[0].build(3 + D)
        m.body.top_block.layers[1].build(8)
        m.output.to_call.build(4)
        return m

    a, b = build(1), build(5)
    k = a.body.top_block.layers[0].kernel
    k.state["accumulator"] = torch.arange(k.data.numel(), dtype=torch.float32).reshape(k.data.shape)
    a.save_weights(tmp_path / "new")
    z = dict(np.load(tmp_path / "new.npz"))
    assert str(z["__format__"]) == CHECKPOINT_FORMAT
    # the same weights as an OLD file: no tag, the kernel's rows in [interactions | bottom] order
    pos = [i for i, p in enumerate(a.parameters()) if p is k][0]
    old = {key: v for key, v in z.items() if key != "__format__"}
    for key in (f"p{pos}", f"s{pos}:accumulator"):
        old[key] = np.concatenate([z[key][D:], z[key][:D]], axis=0)
    np.savez(tmp_path / "old", **old)
    # the file cannot say which of the two untagged layouts it holds: refused without the caller's word (round-5 advisor finding)
    with pytest.raises(ValueError, match="legacy_layout"):
        b.load_weights(tmp_path / "old")
    b.load_weights(tmp_path / "old", legacy_layout="interactions_first")
    kb = b.body.top_block.layers[0].kernel
    assert torch.equal(kb.data, k.data) and torch.equal(kb.state["accumulator"], k.state["accumulator"])
    # an untagged file written after the order changed (bottom first) loads as it is
    c = build(9)
    np.savez(tmp_path / "window", **{key: v for key, v in z.items() if key != "__format__"})
    c.load_weights(tmp_path / "window", legacy_layout="bottom_first")
    kc = c.body.top_block.layers[0].kernel
    assert torch.equal(kc.data, k.data) and torch.equal(kc.state["accumulator"], k.state["accumulator"])
    with pytest.raises(ValueError, match="untagged checkpoints only"):
        c.load_weights(tmp_path / "new", legacy_layout="bottom_first")
    z["__format__"] = np.array("models_amd/99")
    np.savez(tmp_path / "future", **z)
    with pytest.raises(ValueError, match="format"):
        b.load_weights(tmp_path / "future")


def test_embedding_initializer_statistics():
    """tests/unit/tf/inputs/test_embedding.py:372-393, 591-629: the keras Embedding default "uniform" draws from
    U(-0.05, 0.05); the V1 EmbeddingFeatures default is TruncatedNormal(0, 0.05) (two-sigma truncation)."""
    from models_amd import schema as S

    col = S.categorical("item", 20000)
    t = mm.EmbeddingTable(16, col, device="cpu", seed=1)
    w = t.table.embeddings.numpy()
    assert w.shape == (20000, 16) and abs(w.mean()) < 1e-3
    assert w.min() >= -0.05 and w.max() <= 0.05 and abs(w.std() - 0.05 / np.sqrt(3)) < 5e-4
    tn = mm.EmbeddingTable(16, col, device="cpu", seed=1, embeddings_initializer="truncated_normal").table.numpy()
    assert abs(tn.mean()) < 1e-3 and np.abs(tn).max() <= 0.1 + 1e-6  # truncated at two standard deviations
    assert 0.040 < tn.std() < 0.050                                   # 0.05 * 0.8796 for a 2-sigma truncation
    a = mm.EmbeddingTable(16, col, device="cpu", seed=5).table.numpy()
    b = mm.EmbeddingTable(16, col, device="cpu", seed=5).table.numpy()
    assert np.array_equal(a, b)  # seeded


def test_sharded_construction_draws_the_rows_of_the_unsharded_table():
    """distributed.sharded_tables: a table built as a row shard holds rows rank, rank + W, ... of EXACTLY the table an unsharded
    build holds -- for both drawn initializers, below and above the chunked-draw threshold, and for given (pretrained) values
    (round-2 advisor finding: the promise did not hold for tables under 2^24 elements; only 'uniform' could be sharded)."""
    import numpy as np
    import torch

    from models_amd import inputs as I

    dev = torch.device("cpu")
    for init in ("uniform", "truncated_normal"):
        for rows, dim in ((1000, 8), (300_001, 64)):  # 19.2 M elements: drawn in 2^20-row chunks
            full = I._init_table(init, rows, dim, dev, seed=5)
            assert full.shape == (rows, dim) and float(full.abs().max()) <= 0.1 + 1e-6
            for W in (2, 3):
                for rank in range(W):
                    part = I._init_table(init, rows, dim, dev, seed=5, shard=(rank, W))
                    assert torch.equal(part, full[rank::W]), (init, rows, W, rank)
    vals = np.arange(7 * 4, dtype=np.float32).reshape(7, 4)
    assert torch.equal(I._init_table(vals, 7, 4, dev, None, shard=(1, 3)), torch.from_numpy(vals)[1::3])
    assert torch.equal(I._init_table(lambda shape: torch.ones(shape), 7, 4, dev, None, shard=(2, 3)), torch.ones(2, 4))
    with pytest.raises(ValueError):
        I._init_table("orthogonal", 7, 4, dev, None)


def test_parameters_walk_is_the_preorder_of_the_recursive_definition():
    """Block.parameters() / optim._walk are iterative (they run every step); they must list exactly what the recursive definition
    lists, in the same order -- optimizer state and checkpoints are keyed by that order -- including shared tables (listed once)."""
    import models_amd as mm
    from models_amd import optim, schema as S

    def rec_params(b):
        seen, out = set(), []
        for p in b.own_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        for c in b.children():
            for p in rec_params(c):
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out

    def rec_walk(b):
        yield b
        for c in b.children():
            yield from rec_walk(c)

    dev = torch.device("cpu")
    cols = [S.categorical("user_id", 50, [S.Tags.USER, S.Tags.USER_ID]), S.categorical("item_id", 70, [S.Tags.ITEM, S.Tags.ITEM_ID]),
            S.categorical("item_cat", 9, [S.Tags.ITEM]), S.continuous("price", [S.Tags.ITEM]), S.continuous("age", [S.Tags.USER]),
            S.binary_target("click")]
    schema = mm.Schema(cols)
    models = [
        mm.DLRMModel(schema, embedding_dim=16, bottom_block=mm.MLPBlock([32, 16], device=dev), top_block=mm.MLPBlock([32, 8], device=dev), device=dev),
        mm.DCNModel(schema, depth=2, deep_block=mm.MLPBlock([32, 16], device=dev), embedding_dim=8, device=dev),
        mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=dev), embedding_dim=8, device=dev),
    ]
    for m in models:
        assert [id(p) for p in m.parameters()] == [id(p) for p in rec_params(m)]
        assert [id(b) for b in optim._walk(m)] == [id(b) for b in rec_walk(m)]
        assert len(m.parameters()) > 0


def test_reference_names_of_the_hot_path_exist_and_build():
    """merlin/models/tf/__init__.py:42-47, 100-102, 127-132, 165: a reference script importing these names must not fail at import;
    each is a thin class over the code that does the work (models_amd/compat.py)."""
    from models_amd import schema as S

    for n in ("L2Norm", "ItemRetrievalScorer", "LogitsTemperatureScaler", "LazyAdam", "MultiOptimizer", "OptimizerBlocks",
              "split_embeddings_on_size", "BinaryClassificationTask", "ItemRetrievalTask", "RecallAt", "NDCGAt", "MRRAt",
              "PrecisionAt", "AvgPrecisionAt", "TopKMetricsAggregator"):
        assert hasattr(mm, n), n
    schema = mm.Schema([S.categorical("user_id", 500, [S.Tags.USER, S.Tags.USER_ID]),
                        S.categorical("item_id", 300, [S.Tags.ITEM, S.Tags.ITEM_ID]), S.binary_target("click")])
    # V1 retrieval vocabulary -> the same scorer object the V2 vocabulary builds
    task = mm.ItemRetrievalTask(schema, logits_temperature=0.5, store_negative_ids=True)
    m = mm.TwoTowerModel(schema, mm.MLPBlock([16], device="cpu"), embedding_dim=8, device="cpu", prediction_tasks=task)
    assert isinstance(m.output, mm.ItemRetrievalScorer) and isinstance(m.output, mm.ContrastiveOutput)
    assert m.output.logits_temperature == 0.5 and m.output.store_negative_ids and m.output.col_schema.name == "item_id"
    with pytest.raises(ValueError):
        mm.ItemRetrievalTask(mm.Schema([S.categorical("a", 5)]))
    # V1 ranking vocabulary
    t = mm.BinaryClassificationTask("click")
    d = mm.DLRMModel(schema, embedding_dim=8, top_block=mm.MLPBlock([8], device="cpu"), prediction_tasks=t, device="cpu")
    assert isinstance(d.output, mm.BinaryOutput) and d.output.target == "click" and t.task_name == "click/binary_classification_task"
    assert mm.BinaryClassificationTask(schema.select_by_tag(S.Tags.BINARY_CLASSIFICATION) if hasattr(S.Tags, "BINARY_CLASSIFICATION") else "click").target_name == "click"
    # temperature scaler: Prediction in training / testing only
    sc = mm.LogitsTemperatureScaler(0.25)
    import torch

    p = mm.Prediction(torch.ones(2, 3), torch.zeros(2, 3))
    assert torch.equal(sc(p, training=True).outputs, torch.full((2, 3), 4.0)) and sc(p) is p
    # per-block optimizers
    emb = d.body.embeddings
    large, small = mm.split_embeddings_on_size(emb, 400)
    assert [t_.input_dim for t_ in large] == [500] and len(small) == 1
    mo = mm.MultiOptimizer([mm.OptimizerBlocks("sgd", large), mm.OptimizerBlocks(mm.LazyAdam(0.01), small)], default_optimizer="adagrad")
    own = mo._owner(d)
    assert own[id(large[0].table)].name == "sgd" and own[id(small[0].table)].name == "adam" and len(mo.optimizers) == 3
    with pytest.raises(ValueError):
        mm.MultiOptimizer([], "sgd")
    with pytest.raises(ValueError):  # the reference's default "rmsprop" has no fused kernel here: it must be named explicitly
        mm.MultiOptimizer([mm.OptimizerBlocks("sgd", large)], default_optimizer="rmsprop")
    assert [m_.name for m_ in mm.TopKMetricsAggregator.default_metrics([10]).topk_metrics] == \
        ["recall_at_10", "mrr_at_10", "ndcg_at_10", "map_at_10", "precision_at_10"]
