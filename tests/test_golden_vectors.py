"""Golden vectors produced by executing the reference's own torch functions
(tests/golden/make_golden.py): the oracle must reproduce them on CPU, the HIP path on the GPU."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

G = np.load(Path(__file__).resolve().parent / "golden" / "reference_vectors.npz")
ATOL = 1e-4


def test_oracle_scorer_matches_reference_vectors():
    out, tgt = O.contrastive_outputs(G["sc_q"], G["sc_pos"], G["sc_neg"], G["sc_pos_id"], G["sc_neg_id"])
    np.testing.assert_allclose(out, G["sc_logits"], atol=1e-5)
    np.testing.assert_array_equal(tgt, G["sc_target"])
    out, _ = O.contrastive_outputs(G["sc_q"], G["sc_pos"], G["sc_pos"], G["ib_ids"], G["ib_ids"], false_negative_score=-655.04)
    np.testing.assert_allclose(out, G["ib_logits"], atol=1e-5)
    out, _ = O.contrastive_outputs(G["sc_q"], G["sc_pos"], G["sc_neg"], downscore_false_negatives=False)
    np.testing.assert_allclose(out, G["nd_logits"], atol=1e-5)


@pytest.mark.parametrize("f,d", [(27, 64), (4, 8), (17, 32)])
def test_oracle_interaction_matches_reference_vectors(f, d):
    np.testing.assert_allclose(O.dot_interaction(G[f"int_x_{f}_{d}"]), G[f"int_y_{f}_{d}"], atol=1e-4, rtol=1e-5)


def test_oracle_cross_matches_reference_vectors():
    layers = [(G[f"cr_W{i}"], G[f"cr_b{i}"]) for i in range(3)]
    np.testing.assert_allclose(O.cross_block(G["cr_x"], layers), G["cr_y"], atol=1e-4, rtol=1e-5)
    low = [(G[f"crl_U{i}"], G[f"crl_V{i}"], G[f"crl_b{i}"]) for i in range(3)]
    np.testing.assert_allclose(O.cross_block(G["cr_x"], low), G["crl_y"], atol=1e-4, rtol=1e-5)


def test_inferred_embedding_dims_match_reference():
    from models_amd.schema import infer_embedding_dim

    got = [infer_embedding_dim(int(c), 2.0, True) for c in G["emb_card"]]
    np.testing.assert_array_equal(got, G["emb_dim"])
    got = [infer_embedding_dim(int(c), 2.0, False) for c in G["emb_card"]]
    np.testing.assert_array_equal(got, G["emb_dim_raw"])


def test_oracle_losses_match_the_reference_loss_classes():
    """BCE and softmax-CE pinned to the loss modules the reference's torch backend instantiates (BinaryOutput.DEFAULT_LOSS_CLS,
    torch/outputs/classification.py:44; ContrastiveOutput's default loss, torch/outputs/contrastive.py:59)."""
    assert str(G["bce_cls"]) == "BCELoss" and str(G["ce_cls"]) == "CrossEntropyLoss"
    np.testing.assert_allclose(O.binary_crossentropy(G["bce_p"], G["bce_y"]).mean(), G["bce_loss"], atol=1e-6, rtol=1e-6)
    loss, _ = O.softmax_ce_first_column(G["ib_logits"])
    np.testing.assert_allclose(loss.mean(), G["ce_loss"], atol=1e-5, rtol=1e-6)
    np.testing.assert_allclose(G["ce_loss_onehot"], G["ce_loss"], atol=1e-6)  # one-hot column 0 == class index 0


def _dl_stack():
    """The DLRM-layout fixture's inputs as the stacked [B, F, D] tensor in StackFeatures' sorted-key order."""
    names = [str(n) for n in G["dl_names"]]
    feats = {**{n: G[f"dl_emb_{n}"] for n in names}, "bottom_block": G["dl_bottom"]}
    order = sorted(feats)
    return np.stack([feats[k] for k in order], axis=1), order.index("bottom_block"), names


def test_oracle_dlrm_top_input_layout_matches_reference_key_logic():
    """[bottom_block | interactions]: the fixture is what the reference's OWN source produces -- WithShortcut's branch dict,
    ParallelBlock.call's merge, Filter.call and ConcatFeatures.call (tf/core/combinators.py:564-569,669-693,
    tf/core/tabular.py:552-576, tf/core/aggregation.py:54-66) executed by tests/golden/make_golden.py, and the torch twin
    InteractionBlock.forward (torch/blocks/dlrm.py:92-104)."""
    assert [str(k) for k in G["dl_key_order"]] == ["bottom_block", "sequential_block_7"]
    assert "shortcut" not in [str(k) for k in G["dl_merged_keys"]]  # the Filter's dict is merged by update: its key survives
    X, slot, _ = _dl_stack()
    D = X.shape[2]
    np.testing.assert_array_equal(G["dl_top_in"][:, :D], G["dl_bottom"])
    np.testing.assert_allclose(O.dlrm_interaction_concat(X, G["dl_bottom"]), G["dl_top_in"], atol=1e-5)
    np.testing.assert_allclose(O.dlrm_interaction_concat(X, G["dl_bottom"]), G["dl_twin_top_in"], atol=1e-5)


def test_oracle_dense_matches_reference_mlp_block():
    """Dense (a6) pinned through the module sequence the reference's torch MLPBlock.__init__ builds (Linear -> ReLU per layer,
    torch/blocks/mlp.py:51-84), executed from the reference source with fixed weights."""
    assert [str(m) for m in G["mlp_layers"]] == ["Identity", "LazyLinear", "ReLU", "LazyLinear", "ReLU", "LazyLinear", "ReLU"]
    layers = [(G[f"mlp_W{i}"], G[f"mlp_b{i}"], "relu") for i in range(3)]
    np.testing.assert_allclose(O.mlp(G["mlp_x"], layers), G["mlp_y"], atol=1e-5, rtol=1e-5)


def _bn_keras_var(var_torch, var0, m_rows, momentum=0.99):
    """torch feeds the UNBIASED batch variance into running_var, Keras the biased one: the fixture's running_var in Keras'
    convention (same batch, same momentum)."""
    batch_unbiased = (var_torch - momentum * var0) / (1.0 - momentum)
    return momentum * var0 + (1.0 - momentum) * batch_unbiased * (m_rows - 1) / m_rows


def test_oracle_batchnorm_matches_reference_mlp_block():
    """BatchNormalization pinned through the reference's torch MLPBlock (Linear -> ReLU -> the normalization module it was
    handed, torch/blocks/mlp.py:65-76) with Keras' constants: training output, moving mean, moving variance (converted from
    torch's unbiased convention), inference output."""
    assert [str(m) for m in G["bn_layers"]] == ["Identity", "LazyLinear", "ReLU", "BatchNorm1d"]
    h = O.mlp(G["bn_x"], [(G["bn_W"], G["bn_b"], "relu")])
    y, mean1, var1 = O.batchnorm_train(h, G["bn_gamma"], G["bn_beta"], G["bn_mean0"], G["bn_var0"])
    np.testing.assert_allclose(y, G["bn_y_train"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(mean1, G["bn_mean1"], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(var1, _bn_keras_var(G["bn_var1_unbiased"], G["bn_var0"], h.shape[0]), atol=1e-6, rtol=1e-6)
    yi = O.batchnorm_infer(h, G["bn_gamma"], G["bn_beta"], G["bn_mean1"], G["bn_var1_unbiased"])
    np.testing.assert_allclose(yi, G["bn_y_infer"], atol=2e-5, rtol=1e-5)


def test_oracle_sparse_adagrad_matches_reference_optimizer():
    """Adagrad pinned through the optimizer the reference's torch `create_optimizer(module, "adagrad")` builds (fixture ada_*:
    Keras' accumulator start 0.1 and epsilon 1e-7 written into it): two IndexedSlices steps of the oracle -- duplicates summed,
    only looked-up rows touched -- leave the table and the accumulator where the reference's optimizer leaves them."""
    w, acc = G["ada_w0"].copy(), np.full_like(G["ada_w0"], 0.1)
    for step in range(2):
        ids, dy = G["ada_ids"][step], G["ada_dy"][step]
        uniq, inv = np.unique(ids, return_inverse=True)
        assert len(uniq) < len(ids)  # the fixture has duplicate ids
        gsum = np.zeros((len(uniq), w.shape[1]), dtype=np.float32)
        np.add.at(gsum, inv, dy)
        w_rows, s_rows = w[uniq], acc[uniq]
        O._optimizer_update(w_rows, gsum, s_rows, "adagrad", float(G["ada_lr"]))
        w[uniq], acc[uniq] = w_rows, s_rows
        np.testing.assert_allclose(w, G[f"ada_w{step + 1}"], atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(acc, G[f"ada_acc{step + 1}"], atol=1e-6, rtol=1e-5)


# ---- the same vectors through the HIP path ------------------------------------------------------
@pytest.mark.gpu
def test_hip_dlrm_top_input_layout_matches_reference_key_logic(device):
    """The unfused kernel, the fused gather -> interaction kernel and a whole DLRMBlock reproduce the reference's
    [bottom_block | interactions] rows (fixture dl_top_in)."""
    import torch

    import models_amd as mm
    from models_amd import ops, schema as S

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    X, slot, names = _dl_stack()
    B, F, D = X.shape
    out = ops.dot_interaction(t(X), t(G["dl_bottom"]))
    np.testing.assert_allclose(out.cpu().numpy(), G["dl_top_in"], atol=ATOL, rtol=1e-5)
    # fused: every embedding "table" is the fixture's [B, D] block looked up with ids 0..B-1; D = 8 is outside the fused
    # kernel's {16, 32, 64, 128}, so pad the rows with zeros to 16 (dot products unchanged) and compare the pair columns
    pad = lambda a: np.concatenate([a, np.zeros_like(a)], axis=1)
    order = sorted(names + ["bottom_block"])
    tabs = [None if k == "bottom_block" else t(pad(G[f"dl_emb_{k}"])) for k in order]
    ids = [None if k == "bottom_block" else torch.arange(B, dtype=torch.int32, device=device) for k in order]
    fused = ops.dlrm_interaction_fused(tabs, ids, t(pad(G["dl_bottom"]))).cpu().numpy()
    np.testing.assert_array_equal(fused[:, :D], G["dl_bottom"])
    np.testing.assert_allclose(fused[:, 2 * D:], G["dl_top_in"][:, D:], atol=ATOL, rtol=1e-5)
    # a whole DLRMBlock with identity bottom / top blocks would hide nothing either: build one and read its top-MLP input
    cols = [S.categorical(n, B) for n in names] + [S.continuous("I1")]
    block = mm.DLRMBlock(mm.Schema(cols), embedding_dim=D, bottom_block=mm.MLPBlock([D], device=device),
                         top_block=mm.MLPBlock([4], device=device), device=device)
    for n in names:
        block.embeddings.feature_table[n].table.data.copy_(t(G[f"dl_emb_{n}"]))
    dense = block.bottom_block.layers[-1]
    x_in = {n: torch.arange(B, device=device).reshape(B, 1) for n in names}
    x_in["I1"] = torch.rand(B, 1, device=device)
    block(x_in)
    bottom = block.bottom_block(block.continuous(x_in)).cpu().numpy()
    top_in = block._top_in.cpu().numpy()
    feats = {**{n: G[f"dl_emb_{n}"] for n in names}, "bottom_block": bottom}
    ref = O.dlrm_interaction_concat(np.stack([feats[k] for k in sorted(feats)], axis=1), bottom)
    np.testing.assert_array_equal(top_in[:, :D], bottom)
    np.testing.assert_allclose(top_in, ref, atol=ATOL, rtol=1e-5)


@pytest.mark.gpu
def test_hip_dense_matches_reference_mlp_block(device):
    import torch

    from models_amd import ops

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    h = t(G["mlp_x"])
    for i in range(3):
        h = ops.linear(h, t(G[f"mlp_W{i}"]), t(G[f"mlp_b{i}"]), "relu")
    np.testing.assert_allclose(h.cpu().numpy(), G["mlp_y"], atol=ATOL, rtol=1e-5)
    y = ops.mlp_chain(t(G["mlp_x"]), [t(G[f"mlp_W{i}"]) for i in range(3)], [t(G[f"mlp_b{i}"]) for i in range(3)], ["relu"] * 3)
    np.testing.assert_allclose((y[-1] if isinstance(y, (list, tuple)) else y).cpu().numpy(), G["mlp_y"], atol=ATOL, rtol=1e-5)


@pytest.mark.gpu
def test_hip_scorer_matches_reference_vectors(device):
    import torch

    from models_amd import ops

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    r = ops.inbatch_softmax(t(G["sc_q"]), t(G["sc_pos"]), t(G["sc_neg"]), t(G["sc_pos_id"]), t(G["sc_neg_id"]))
    np.testing.assert_allclose(r.logits.cpu().numpy(), G["sc_logits"], atol=ATOL)
    r = ops.inbatch_softmax(t(G["sc_q"]), t(G["sc_pos"]), t(G["sc_pos"]), t(G["ib_ids"]), t(G["ib_ids"]))
    np.testing.assert_allclose(r.logits.cpu().numpy(), G["ib_logits"], atol=ATOL)
    r = ops.inbatch_softmax(t(G["sc_q"]), t(G["sc_pos"]), t(G["sc_neg"]))
    np.testing.assert_allclose(r.logits.cpu().numpy(), G["nd_logits"], atol=ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize("f,d", [(27, 64), (4, 8), (17, 32)])
def test_hip_interaction_matches_reference_vectors(device, f, d):
    import torch

    from models_amd import ops

    out = ops.dot_interaction(torch.from_numpy(G[f"int_x_{f}_{d}"]).to(device)).cpu().numpy()
    np.testing.assert_allclose(out, G[f"int_y_{f}_{d}"], atol=ATOL, rtol=1e-5)


@pytest.mark.gpu
def test_hip_cross_matches_reference_vectors(device):
    import torch

    from models_amd import ops

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    x0 = x = t(G["cr_x"])
    for i in range(3):
        x = ops.cross_layer(x0, x, t(G[f"cr_W{i}"]), t(G[f"cr_b{i}"]))
    np.testing.assert_allclose(x.cpu().numpy(), G["cr_y"], atol=ATOL, rtol=1e-5)
    x = x0  # low-rank layers: h = x U (Dense without bias), then the cross epilogue on h V + b
    for i in range(3):
        h = ops.linear(x, t(G[f"crl_U{i}"]), None, None)
        x = ops.cross_layer_lowrank(x0, x, h, t(G[f"crl_V{i}"]), t(G[f"crl_b{i}"]))
    np.testing.assert_allclose(x.cpu().numpy(), G["crl_y"], atol=ATOL, rtol=1e-5)


@pytest.mark.gpu
def test_bag_lookup_matches_reference_embedding_bag(device):
    """HIP ragged lookup (mh_embedding_bag_fwd) against the reference torch backend's F.embedding_bag outputs."""
    import torch

    from models_amd import ops

    W = torch.from_numpy(G["bag_W"]).to(device)
    vals = torch.from_numpy(G["bag_values"]).to(device)
    offs = torch.from_numpy(G["bag_offsets"]).to(device)
    for mode in ("sum", "mean"):
        got = ops.embedding_bag(W, vals, offs, mode)
        np.testing.assert_allclose(got.cpu().numpy(), G[f"bag_{mode}"], atol=1e-6)


@pytest.mark.gpu
def test_hip_losses_match_the_reference_loss_classes(device):
    """mh_bce_mean_fwd_bwd and the scorer's fused softmax-CE against nn.BCELoss / nn.CrossEntropyLoss outputs."""
    import torch

    from models_amd import ops

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    mean, dlogit = ops.bce(t(G["bce_p"]), t(G["bce_y"]))
    np.testing.assert_allclose(float(mean), G["bce_loss"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(ops.bce_per_sample(t(G["bce_p"]), t(G["bce_y"])).mean().item(), G["bce_loss"], atol=1e-6, rtol=1e-5)
    # d mean / d logit of sigmoid + BCE = (p - y) / M
    np.testing.assert_allclose(dlogit.cpu().numpy(), (G["bce_p"] - G["bce_y"]) / G["bce_p"].shape[0], atol=1e-8, rtol=1e-5)
    ids = t(G["ib_ids"])
    for mat in (True, False):
        r = ops.inbatch_softmax(t(G["sc_q"]), t(G["sc_pos"]), t(G["sc_pos"]), ids, ids, materialize=mat)
        np.testing.assert_allclose(r.loss.mean().item(), G["ce_loss"], atol=1e-5, rtol=1e-5)


@pytest.mark.gpu
def test_hip_batchnorm_mlp_block_matches_reference_mlp_block(device):
    """``mm.MLPBlock([24], normalization="batch_norm")`` with the fixture's weights: training output, every gradient of the
    reference's autograd pass, the moving statistics after one training call, and the inference output."""
    import torch

    import models_amd as mm
    from models_amd import blocks

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    blk = mm.MLPBlock([24], normalization="batch_norm", device=device)
    dense, bn = blk.layers
    x = t(G["bn_x"])
    blk(x)  # build
    dense.kernel.data.copy_(t(G["bn_W"]))
    dense.bias.data.copy_(t(G["bn_b"]))
    bn.gamma.data.copy_(t(G["bn_gamma"]))
    bn.beta.data.copy_(t(G["bn_beta"]))
    bn.moving_mean.copy_(t(G["bn_mean0"]))
    bn.moving_variance.copy_(t(G["bn_var0"]))
    with blocks.tape():
        y = blk(x)
    np.testing.assert_allclose(y.cpu().numpy(), G["bn_y_train"], atol=ATOL, rtol=1e-5)
    dx = blk.backward(t(G["bn_dy"]).clone())
    np.testing.assert_allclose(dx.cpu().numpy(), G["bn_dx"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(dense.kernel.grad.cpu().numpy(), G["bn_dW"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(dense.bias.grad.cpu().numpy(), G["bn_db"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(bn.gamma.grad.cpu().numpy(), G["bn_dgamma"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(bn.beta.grad.cpu().numpy(), G["bn_dbeta"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(bn.moving_mean.cpu().numpy(), G["bn_mean1"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(bn.moving_variance.cpu().numpy(),
                               _bn_keras_var(G["bn_var1_unbiased"], G["bn_var0"], x.shape[0]), atol=1e-6, rtol=1e-5)
    bn.moving_variance.copy_(t(G["bn_var1_unbiased"]))  # the statistics the reference's inference call used
    np.testing.assert_allclose(blk(x).cpu().numpy(), G["bn_y_infer"], atol=ATOL, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("deterministic", ["0", "1"])
def test_hip_sparse_adagrad_matches_reference_optimizer(device, deterministic, monkeypatch):
    """``mh_embedding_gather_bwd`` (Adagrad) on the fixture of the reference's optimizer: table and accumulator after two steps."""
    import torch

    from models_amd import ops

    monkeypatch.setenv("MERLIN_HIP_DETERMINISTIC", deterministic)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    w, acc = t(G["ada_w0"]), torch.full(G["ada_w0"].shape, 0.1, device=device)
    for step in range(2):
        ids, dy = t(G["ada_ids"][step]), t(G["ada_dy"][step])
        ops.embedding_gather_backward([w], [acc], [ids], dy.reshape(dy.shape[0], 1, -1).contiguous(), [0], "adagrad",
                                      float(G["ada_lr"]), 1e-7)
        np.testing.assert_allclose(w.cpu().numpy(), G[f"ada_w{step + 1}"], atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(acc.cpu().numpy(), G[f"ada_acc{step + 1}"], atol=1e-6, rtol=1e-5)
