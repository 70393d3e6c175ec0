"""mm.TwoTowerModel / mm.DCNModel / top-k encoder on the HIP path vs oracle + torch reference."""
import os

import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import schema as S
from oracle import oracle as O
from tests import torch_ref as R

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def _two_tower_schema():
    return mm.Schema([
        S.categorical("user_id", 500, [S.Tags.USER, S.Tags.USER_ID]),
        S.categorical("user_city", 20, [S.Tags.USER]),
        S.categorical("item_id", 300, [S.Tags.ITEM, S.Tags.ITEM_ID]),
        S.categorical("item_category", 12, [S.Tags.ITEM]),
    ])


def _batch(schema, B, g, device):
    x = {}
    for c in schema:
        if S.Tags.CATEGORICAL in c.tags:
            x[c.name] = torch.randint(0, int(c.int_domain.max) + 1, (B, 1), generator=g)
        elif S.Tags.CONTINUOUS in c.tags:
            x[c.name] = torch.rand(B, 1, generator=g)
    return x, {k: v.to(device) for k, v in x.items()}


def _tower_np(tower, x):
    """oracle forward of SequentialBlock[InputBlockV2, MLP] (concat in sorted-name order)."""
    inp, mlp = tower.layers[0], tower.layers[1:]
    feats = {}
    for n in inp.categorical.feature_names:
        feats[n] = O.embedding_lookup(inp.categorical.feature_table[n].table.numpy(), x[n].numpy())
    h = O.concat_features(feats)
    for l in mlp:
        h = O.dense(h, l.kernel.numpy(), l.bias.numpy(), l.activation)
    return h


def test_two_tower_forward_matches_oracle(device):
    schema = _two_tower_schema()
    model = mm.TwoTowerModel(schema, mm.MLPBlock([64, 32], device=device), embedding_dim=32, device=device, logits_temperature=0.5)
    g = torch.Generator().manual_seed(0)
    x, xd = _batch(schema, 257, g, device)
    pred = model(xd, training=True)
    q = _tower_np(model.body.parallel_layers["query"], x)
    it = _tower_np(model.body.parallel_layers["item"], x)
    ids = x["item_id"].numpy().reshape(-1)
    logits, targets = O.contrastive_outputs(q, it, it, ids, ids, temperature=0.5)
    # north_star: logits within 1e-4 (absolute), at the configured temperature; achieved on this batch: ~1e-6
    np.testing.assert_allclose(pred.outputs.cpu().numpy(), logits, atol=ATOL, rtol=0)
    np.testing.assert_array_equal(pred.targets.cpu().numpy(), targets)
    # inference returns only the positive scores [B, 1] (tests/unit/tf/outputs/test_contrastive.py:209-223)
    inf = model(xd)
    assert inf.shape == (257, 1)
    np.testing.assert_allclose(inf.cpu().numpy()[:, 0], (q * it).sum(-1), atol=ATOL)


def test_top_k_encoder_matches_matmul_topk_gather(device):
    # tests/unit/tf/core/test_index.py:76-121
    schema = _two_tower_schema()
    model = mm.TwoTowerModel(schema, mm.MLPBlock([32], device=device), embedding_dim=16, device=device)
    g = torch.Generator().manual_seed(1)
    x, xd = _batch(schema, 64, g, device)
    model(xd)
    cands = torch.randn(300, 32, generator=g).to(device)
    ident = torch.randperm(5000, generator=g)[:300].to(device)
    enc = model.to_top_k_encoder(cands, ident, k=10)
    out = enc(xd)
    q = model.query_embeddings(xd).cpu().numpy()
    vals, ids, _ = O.brute_force_topk(q, cands.cpu().numpy(), ident.cpu().numpy(), 10)
    np.testing.assert_array_equal(out.identifiers.cpu().numpy(), ids)
    np.testing.assert_allclose(out.scores.cpu().numpy(), vals, atol=ATOL)
    assert out.identifiers.dtype == torch.int32
    with pytest.raises(ValueError):
        mm.BruteForce(3).forward(cands)  # not indexed (tests/unit/tf/outputs/test_topk.py:21-78)
    with pytest.raises(ValueError):
        mm.BruteForce(3).index(cands, ident[:5])


@pytest.mark.parametrize("opt,l2", [("sgd", False), ("adagrad", False), ("sgd", True)])
def test_two_tower_train_steps_match_torch(device, opt, l2):
    schema = _two_tower_schema()
    lr, T = 0.05, 0.7
    model = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device), embedding_dim=16, device=device,
                             logits_temperature=T, l2_normalization=l2)
    model.compile(optimizer=opt, learning_rate=lr)
    g = torch.Generator().manual_seed(2)
    batches = [_batch(schema, 200, g, device) for _ in range(3)]
    model(batches[0][1])
    towers = {k: model.body.parallel_layers[k] for k in ("query", "item")}
    ref = {}
    for k, tw in towers.items():
        inp = tw.layers[0]
        ref[k] = {
            "tables": {n: inp.categorical.feature_table[n].table.data.cpu().clone().requires_grad_() for n in inp.categorical.feature_names},
            "mlp": [(l.kernel.data.cpu().clone().requires_grad_(), l.bias.data.cpu().clone().requires_grad_(), l.activation) for l in tw.layers[1:]],
        }

    def tower_t(k, x):
        r = ref[k]
        h = torch.cat([r["tables"][n][x[n].reshape(-1)] for n in sorted(r["tables"])], dim=1)
        for W, b, a in r["mlp"]:
            h = R.act(h @ W + b, a)
        return h

    params = [p for k in ref for p in list(ref[k]["tables"].values()) + [t for l in ref[k]["mlp"] for t in l[:2]]]
    is_table = [True] * 0
    is_table = []
    for k in ref:
        is_table += [True] * len(ref[k]["tables"]) + [False] * (2 * len(ref[k]["mlp"]))
    accs = [torch.full_like(p, 0.1) for p in params]
    for x, xd in batches:
        loss = model.train_step(xd)
        q, it = tower_t("query", x), tower_t("item", x)
        if l2:  # tf.linalg.l2_normalize(x, axis=-1): x * rsqrt(max(sum x^2, 1e-12))
            q = q * torch.rsqrt(torch.clamp((q * q).sum(-1, keepdim=True), min=1e-12))
            it = it * torch.rsqrt(torch.clamp((it * it).sum(-1, keepdim=True), min=1e-12))
        ids = x["item_id"].reshape(-1)
        pos = (q * it).sum(-1, keepdim=True)
        neg = q @ it.T
        neg = torch.where(ids[:, None] == ids[None, :], torch.full_like(neg, O.MIN_FLOAT), neg)
        logits = torch.cat([pos, neg], 1) / T
        ref_loss = (torch.logsumexp(logits, 1) - logits[:, 0]).mean()
        assert abs(loss.item() - ref_loss.item()) < 1e-4
        grads = torch.autograd.grad(ref_loss, params)
        with torch.no_grad():
            for i, (p, gr) in enumerate(zip(params, grads)):
                if opt == "sgd":
                    p -= lr * gr
                else:
                    touched = (gr != 0).any(dim=-1, keepdim=True) if is_table[i] else torch.ones_like(p, dtype=torch.bool)
                    w2, a2 = R.adagrad_update(p, gr, accs[i], lr)
                    p.copy_(torch.where(touched, w2, p))
                    accs[i] = torch.where(touched, a2, accs[i])
    for k, tw in towers.items():
        inp = tw.layers[0]
        for n, t in ref[k]["tables"].items():
            torch.testing.assert_close(inp.categorical.feature_table[n].table.data.cpu(), t.detach(), atol=1e-4, rtol=1e-4)
        for l, (W, b, _) in zip(tw.layers[1:], ref[k]["mlp"]):
            torch.testing.assert_close(l.kernel.data.cpu(), W.detach(), atol=1e-4, rtol=1e-4)


def _dcn_schema():
    cols = [S.categorical(f"C{i}", 30 + i) for i in range(1, 5)] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    return mm.Schema(cols)


def test_dcn_forward_and_train_match_reference(device):
    schema = _dcn_schema()
    lr = 0.05
    model = mm.DCNModel(schema, depth=2, deep_block=mm.MLPBlock([32, 16], device=device), embedding_dim=16, device=device)
    model.compile(optimizer="sgd", learning_rate=lr)
    g = torch.Generator().manual_seed(4)
    x, xd = _batch(schema, 150, g, device)
    y = torch.randint(0, 2, (150, 1), generator=g).float()
    p = model(xd)
    body = model.body
    inp = body.input_block
    tables = {n: inp.categorical.feature_table[n].table.data.cpu().clone().requires_grad_() for n in inp.categorical.feature_names}
    dc = body.cross.layers[0].d  # cross kernels are stored zero-padded to a multiple of 4
    cross = [(l.kernel.data[:dc, :dc].cpu().clone().requires_grad_(), l.bias.data[:dc].cpu().clone().requires_grad_()) for l in body.cross.layers]
    deep = [(l.kernel.data.cpu().clone().requires_grad_(), l.bias.data.cpu().clone().requires_grad_(), l.activation) for l in body.deep.layers]
    hd = model.output.to_call
    head = (hd.kernel.data.cpu().clone().requires_grad_(), hd.bias.data.cpu().clone().requires_grad_())

    def fwd():
        feats = {n: tables[n][x[n].reshape(-1)] for n in tables}
        feats.update({k: v for k, v in x.items() if k.startswith("I")})
        h0 = torch.cat([feats[k] for k in sorted(feats)], dim=1)  # C1..C4 then I1..I3 (sorted)
        h = h0
        for W, b in cross:
            h = h0 * (h @ W + b) + h
        for W, b, a in deep:
            h = R.act(h @ W + b, a)
        return torch.sigmoid(h @ head[0] + head[1])

    # forward parity (oracle cross_block on the same numbers)
    feats_np = {n: O.embedding_lookup(tables[n].detach().numpy(), x[n].numpy()) for n in tables}
    feats_np.update({k: v.numpy() for k, v in x.items() if k.startswith("I")})
    h = O.cross_block(O.concat_features(feats_np), [(W.detach().numpy(), b.detach().numpy()) for W, b in cross])
    h = O.mlp(h, [(W.detach().numpy(), b.detach().numpy(), a) for W, b, a in deep])
    ref_p = O.dense(h, head[0].detach().numpy(), head[1].detach().numpy(), "sigmoid")
    np.testing.assert_allclose(p.cpu().numpy(), ref_p, atol=ATOL)
    # one SGD step
    loss = model.train_step(xd, y.to(device))
    ref_loss = R.keras_bce(fwd(), y)
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    params = list(tables.values()) + [t for l in cross for t in l] + [t for l in deep for t in l[:2]] + list(head)
    grads = torch.autograd.grad(ref_loss, params)
    with torch.no_grad():
        for pp, gr in zip(params, grads):
            pp -= lr * gr
    for n, t in tables.items():
        torch.testing.assert_close(inp.categorical.feature_table[n].table.data.cpu(), t.detach(), atol=1e-4, rtol=1e-4)
    for l, (W, b) in zip(body.cross.layers, cross):
        torch.testing.assert_close(l.kernel.data[:dc, :dc].cpu(), W.detach(), atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(l.bias.data[:dc].cpu(), b.detach(), atol=1e-4, rtol=1e-4)
        assert torch.all(l.kernel.data[dc:] == 0) and torch.all(l.kernel.data[:, dc:] == 0)  # pads stay zero
    for l, (W, b, _) in zip(body.deep.layers, deep):
        torch.testing.assert_close(l.kernel.data.cpu(), W.detach(), atol=1e-4, rtol=1e-4)


def test_dlrm_forward_matches_oracle_criteo_names(device):
    """Sorted stack order C1, C10, ..., C19, C2, ... , bottom_block."""
    names = [f"C{i}" for i in range(1, 13)]
    cols = [S.categorical(n, 40 + i) for i, n in enumerate(names)] + [S.continuous(f"I{i}") for i in range(1, 6)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)
    D = 16
    model = mm.DLRMModel(schema, embedding_dim=D, bottom_block=mm.MLPBlock([32, D], device=device),
                         top_block=mm.MLPBlock([64, 32], device=device), device=device)
    assert model.body.stack_order == sorted(names + ["bottom_block"])
    g = torch.Generator().manual_seed(5)
    x, xd = _batch(schema, 333, g, device)
    p = model(xd).cpu().numpy()
    body = model.body
    lay = lambda blk: [(l.kernel.numpy(), l.bias.numpy(), l.activation) for l in blk.layers]
    hd = model.output.to_call
    ref = O.dlrm_forward({n: x[n].numpy() for n in names}, {k: v.numpy() for k, v in x.items() if k.startswith("I")},
                         {n: body.embeddings.feature_table[n].table.numpy() for n in names},
                         lay(body.bottom_block), lay(body.top_block), (hd.kernel.numpy(), hd.bias.numpy()))
    np.testing.assert_allclose(p, ref["prob"], atol=ATOL)
    np.testing.assert_allclose(body._top_in.cpu().numpy(), ref["top_in"], atol=ATOL)


def test_dlrm_variants_and_errors():
    # tests/unit/tf/blocks/test_dlrm.py:41-112 (construction-time behaviour, no GPU math needed)
    cols = [S.categorical("a", 10), S.categorical("b", 10), S.continuous("x")]
    schema = mm.Schema(cols)
    with pytest.raises(ValueError, match="needs to match"):
        mm.DLRMBlock(schema, embedding_dim=8, bottom_block=mm.MLPBlock([16]))
    with pytest.raises(ValueError, match="bottom_block is required"):
        mm.DLRMBlock(schema, embedding_dim=8)
    with pytest.raises(ValueError, match="requires categorical"):
        mm.DLRMBlock(mm.Schema([S.continuous("x")]), embedding_dim=8, bottom_block=mm.MLPBlock([8]))
    with pytest.raises(ValueError, match="embedding_dim is required"):
        mm.DLRMBlock(schema, bottom_block=mm.MLPBlock([8]))


@pytest.mark.parametrize("optimizer", ["adagrad", "adam"])
def test_distributed_dlrm_world1_matches_plain_model(device, optimizer):
    """DistributedDLRM with forced row-sharding (W = 1: the all-to-alls degenerate to copies) must
    reproduce the plain model's forward and train steps (routing / un-permute / dense-grad path)."""
    from models_amd.distributed import DistributedDLRM

    # replicated tables (C2, C4) are small enough that every row is hit each step: there the dense Adam of the
    # replicated path and the row-wise LazyAdam of the plain model coincide
    cards = {"C1": 5000, "C2": 7, "C3": 3000, "C4": 5}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)
    D = 16

    def build():
        m = mm.DLRMModel(schema, embedding_dim=D, bottom_block=mm.MLPBlock([32, D], device=device, seed=7),
                         top_block=mm.MLPBlock([32, 8], device=device, seed=17), device=device)
        m.output.to_call.seed = 99
        m.compile(optimizer=optimizer, learning_rate=0.05 if optimizer == "adagrad" else 0.01)
        return m

    g = torch.Generator().manual_seed(3)
    batches = []
    for _ in range(3):
        x, xd = _batch(schema, 300, g, device)
        batches.append((xd, torch.randint(0, 2, (300, 1), generator=g).float().to(device)))
    a, b = build(), build()
    pa = a(batches[0][0])
    b(batches[0][0])
    for pa_, pb_ in zip(a.parameters(), b.parameters()):
        pb_.data.copy_(pa_.data)
    db = DistributedDLRM(b, shard_threshold=1000, force_shard=True)
    assert sorted(db.sharded) == ["C1", "C3"]
    torch.testing.assert_close(db(batches[0][0]), a(batches[0][0]), atol=1e-6, rtol=1e-6)
    for xd, y in batches:
        la, lb = a.train_step(xd, y), db.train_step(xd, y)
        assert abs(la.item() - lb.item()) < 1e-5
    torch.testing.assert_close(db(batches[0][0]), a(batches[0][0]), atol=1e-5, rtol=1e-4)
    for n in cards:
        ta = a.body.embeddings.feature_table[n].table.data
        tb = db.sharded[n] if n in db.sharded else b.body.embeddings.feature_table[n].table.data
        torch.testing.assert_close(tb, ta, atol=1e-5, rtol=1e-4)


def test_loader_feeds_train_steps_and_batch_predict(device):
    """models_amd.Loader (pinned staging + copy stream, one batch ahead) -> train_step / TopKEncoder.batch_predict:
    same result as handing the same rows over as device tensors."""
    schema = _two_tower_schema()
    rng = np.random.default_rng(4)
    n = 200
    data = {}
    for c in schema:
        if S.Tags.CATEGORICAL in c.tags:
            data[c.name] = rng.integers(0, int(c.int_domain.max) + 1, size=n).astype(np.int64)
        elif S.Tags.CONTINUOUS in c.tags:
            data[c.name] = rng.random(n).astype(np.float32)

    def build():
        m = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device, seed=3), embedding_dim=16, device=device)
        m.compile(optimizer="adagrad", learning_rate=0.05)
        return m

    a, b = build(), build()
    first = {k: torch.from_numpy(v[:50]).to(device) for k, v in data.items()}
    a(first), b(first)
    for pa_, pb_ in zip(a.parameters(), b.parameters()):
        pb_.data.copy_(pa_.data)
    ld = mm.Loader(data, schema, batch_size=50, shuffle=False, device=device)
    assert ld.prefetch and len(ld) == 4
    for i, (inputs, targets) in enumerate(ld):
        assert targets is None and inputs[next(iter(inputs))].is_cuda
        la = a.train_step(inputs)
        lb = b.train_step({k: torch.from_numpy(v[50 * i:50 * i + 50]).to(device) for k, v in data.items()})
        assert abs(float(la) - float(lb)) < 1e-6
    for pa_, pb_ in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa_.data, pb_.data, atol=1e-6, rtol=1e-6)
    g = torch.Generator().manual_seed(2)
    cands = torch.randn(300, 16, generator=g).to(device)
    ident = torch.randperm(5000, generator=g)[:300].to(device)
    enc = a.to_top_k_encoder(cands, ident, k=7)
    qschema = schema.select_by_tag(S.Tags.USER)
    frame = enc.batch_predict(data, batch_size=64, schema=qschema, output_schema=mm.Schema([qschema.first]))
    whole = enc({c.name: torch.from_numpy(data[c.name]).to(device) for c in qschema})
    assert list(frame)[0] == qschema.first.name and list(frame)[1:] == mm.TopKPrediction.output_names(7)
    np.testing.assert_array_equal(np.stack([frame[f"id_{i}"] for i in range(7)], 1), whole.identifiers.cpu().numpy())
    np.testing.assert_allclose(np.stack([frame[f"score_{i}"] for i in range(7)], 1), whole.scores.cpu().numpy(), atol=1e-6)
    np.testing.assert_array_equal(frame[qschema.first.name], data[qschema.first.name])


@pytest.mark.parametrize("stacked", [True, False, "deep_first"])
@pytest.mark.parametrize("low_rank_dim", [None, 6])
def test_dcn_variants_train_step_matches_torch(device, stacked, low_rank_dim):
    """Low-rank cross kernels (W = U V) and the parallel DCN form (concat(cross, deep), or concat(deep, cross) when the
    reference's two layer names sort that way: DCNBody docstring): forward and one SGD step against torch autograd on the
    reference-shaped weights."""
    schema = _dcn_schema()
    lr = 0.05
    deep_first = stacked == "deep_first"
    stacked = stacked is True
    model = mm.DCNModel(schema, depth=2, deep_block=mm.MLPBlock([32, 16], device=device), embedding_dim=16,
                        device=device, stacked=stacked, low_rank_dim=low_rank_dim,
                        parallel_concat=("deep", "cross") if deep_first else ("cross", "deep"))
    model.compile(optimizer="sgd", learning_rate=lr)
    g = torch.Generator().manual_seed(4)
    x, xd = _batch(schema, 150, g, device)
    y = torch.randint(0, 2, (150, 1), generator=g).float()
    p = model(xd)
    body = model.body
    inp = body.input_block
    tables = {n: inp.categorical.feature_table[n].table.data.cpu().clone().requires_grad_() for n in inp.categorical.feature_names}
    dc, r = body.cross.layers[0].d, low_rank_dim

    def leaf(t):
        return t.cpu().clone().requires_grad_()

    if r is None:
        cross = [(leaf(l.kernel.data[:dc, :dc]), leaf(l.bias.data[:dc])) for l in body.cross.layers]
    else:
        cross = [(leaf(l.kernel_u.data[:dc, :r]), leaf(l.kernel.data[:r, :dc]), leaf(l.bias.data[:dc])) for l in body.cross.layers]
    deep = [(leaf(l.kernel.data), leaf(l.bias.data), l.activation) for l in body.deep.layers]
    hd = model.output.to_call
    head = (leaf(hd.kernel.data), leaf(hd.bias.data))

    def fwd():
        feats = {n: tables[n][x[n].reshape(-1)] for n in tables}
        feats.update({k: v for k, v in x.items() if k.startswith("I")})
        h0 = torch.cat([feats[k] for k in sorted(feats)], dim=1)
        h = h0
        for layer in cross:
            proj = (h @ layer[0] + layer[1]) if r is None else ((h @ layer[0]) @ layer[1] + layer[2])
            h = h0 * proj + h
        dd = h if stacked else h0
        for W, b, a in deep:
            dd = R.act(dd @ W + b, a)
        out = dd if stacked else torch.cat([dd, h] if deep_first else [h, dd], dim=1)
        return torch.sigmoid(out @ head[0] + head[1])

    np.testing.assert_allclose(p.cpu().numpy(), fwd().detach().numpy(), atol=ATOL)
    loss = model.train_step(xd, y.to(device))
    ref_loss = R.keras_bce(fwd(), y)
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    params = list(tables.values()) + [t for l in cross for t in l] + [t for l in deep for t in l[:2]] + list(head)
    grads = torch.autograd.grad(ref_loss, params)
    with torch.no_grad():
        for pp, gr in zip(params, grads):
            pp -= lr * gr
    for n, t in tables.items():
        torch.testing.assert_close(inp.categorical.feature_table[n].table.data.cpu(), t.detach(), atol=1e-4, rtol=1e-4)
    for l, ref in zip(body.cross.layers, cross):
        if r is None:
            torch.testing.assert_close(l.kernel.data[:dc, :dc].cpu(), ref[0].detach(), atol=1e-4, rtol=1e-4)
        else:
            torch.testing.assert_close(l.kernel_u.data[:dc, :r].cpu(), ref[0].detach(), atol=1e-4, rtol=1e-4)
            torch.testing.assert_close(l.kernel.data[:r, :dc].cpu(), ref[1].detach(), atol=1e-4, rtol=1e-4)
            assert torch.all(l.kernel_u.data[dc:] == 0) and torch.all(l.kernel_u.data[:, r:] == 0)  # pads stay zero
        torch.testing.assert_close(l.bias.data[:dc].cpu(), ref[-1].detach(), atol=1e-4, rtol=1e-4)
    for l, (W, b, _) in zip(body.deep.layers, deep):
        torch.testing.assert_close(l.kernel.data.cpu(), W.detach(), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(hd.kernel.data.cpu(), head[0].detach(), atol=1e-4, rtol=1e-4)


def test_predict_and_evaluate_ranking_and_retrieval(device):
    """Minimal Model.predict / evaluate (SURVEY 8b kept surface): BCE loss, accuracy and AUC for the ranking head;
    in-batch top-k metrics for the retrieval model; catalogue top-k metrics for the TopKEncoder."""
    from sklearn.metrics import roc_auc_score

    # --- ranking ---
    schema = _dcn_schema()
    model = mm.DCNModel(schema, depth=1, deep_block=mm.MLPBlock([16], device=device), embedding_dim=8, device=device)
    g = torch.Generator().manual_seed(8)
    batches = []
    for _ in range(3):
        x, xd = _batch(schema, 90, g, device)
        batches.append((xd, torch.randint(0, 2, (90, 1), generator=g).float().to(device)))
    p = model.predict(batches)
    y = np.concatenate([b[1].cpu().numpy() for b in batches])
    assert p.shape == (270, 1)
    np.testing.assert_allclose(p[:90], model(batches[0][0]).cpu().numpy(), atol=1e-7)
    ev = model.evaluate(batches)
    assert abs(ev["loss"] - float(O.binary_crossentropy(p, y).mean())) < 1e-5
    assert abs(ev["binary_accuracy"] - float(((p > 0.5) == (y > 0.5)).mean())) < 1e-7
    assert abs(ev["auc"] - roc_auc_score(y.reshape(-1), p.reshape(-1))) < 1e-9
    # --- retrieval, in-batch ---
    schema = _two_tower_schema()
    tt = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device), embedding_dim=16, device=device, logits_temperature=0.8)
    rb = [_batch(schema, 120, g, device)[1] for _ in range(2)]
    ev = tt.evaluate(rb, k=5)
    want = np.zeros(6)
    loss = 0.0
    for xd in rb:
        logits = tt(xd, testing=True).outputs.cpu().numpy()
        srt = np.argsort(-logits, axis=1, kind="stable")[:, :5]  # ties -> lower index first (tf.math.top_k)
        labels = (srt == 0).astype(np.float32)
        want += O.topk_metrics(labels, np.ones(len(labels)), 5).sum(0)
        lse = np.log(np.exp(logits - logits.max(1, keepdims=True)).sum(1)) + logits.max(1)
        loss += float((lse - logits[:, 0]).sum())
    want /= 240
    for name, w in zip(("recall", "precision", "map", "dcg", "ndcg", "mrr"), want):
        assert abs(ev[f"{name}_at_5"] - w) < 1e-5, name
    assert abs(ev["loss"] - loss / 240) < 1e-4
    # --- retrieval against an indexed catalogue ---
    cands = torch.randn(300, 16, generator=g).to(device)
    ident = torch.arange(300, dtype=torch.int32).to(device)
    enc = tt.to_top_k_encoder(cands, ident, k=10)
    ev = enc.evaluate(rb, item_id="item_id")
    want = np.zeros(6)
    for xd in rb:
        ids = enc(xd).identifiers.cpu().numpy()
        labels = (ids == xd["item_id"].cpu().numpy().reshape(-1, 1)).astype(np.float32)
        want += O.topk_metrics(labels, np.ones(len(labels)), 10).sum(0)
    want /= 240
    for name, w in zip(("recall", "precision", "map", "dcg", "ndcg", "mrr"), want):
        assert abs(ev[f"{name}_at_10"] - w) < 1e-5, name


def test_two_tower_v2_encoders_equal_v1_model(device):
    """TwoTowerModelV2(Encoder, Encoder) is the same computation as the V1 constructor on the same weights."""
    schema = _two_tower_schema()
    v1 = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device), embedding_dim=16, device=device, logits_temperature=0.6)
    q = mm.Encoder(mm.InputBlockV2(schema.select_by_tag(S.Tags.USER), dim=16, device=device), mm.MLPBlock([32, 16], device=device))
    c = mm.Encoder(mm.InputBlockV2(schema.select_by_tag(S.Tags.ITEM), dim=16, device=device), mm.MLPBlock([32, 16], device=device))
    v2 = mm.TwoTowerModelV2(q, c, schema=schema, logits_temperature=0.6)
    for m in (v1, v2):
        m.compile(optimizer="adagrad", learning_rate=0.05)
    g = torch.Generator().manual_seed(6)
    batches = [_batch(schema, 100, g, device)[1] for _ in range(3)]
    v1(batches[0]), v2(batches[0])
    assert len(v1.parameters()) == len(v2.parameters())
    for a, b in zip(v1.parameters(), v2.parameters()):
        b.data.copy_(a.data)
    for xd in batches:
        la, lb = v1.train_step(xd), v2.train_step(xd)
        assert abs(float(la) - float(lb)) < 1e-6
    for a, b in zip(v1.parameters(), v2.parameters()):
        torch.testing.assert_close(a.data, b.data, atol=1e-6, rtol=1e-6)
    emb = q.encode({k: v.cpu().numpy().reshape(-1) for k, v in batches[0].items() if k.startswith("user")}, batch_size=40)
    np.testing.assert_allclose(emb, v2.query_embeddings(batches[0]).cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("mode", ["one_graph", "segmented", "segmented_deterministic", "recorded", "recorded_deterministic"])
def test_graph_replayed_train_steps_equal_eager_steps(device, mode, monkeypatch):
    """The headline number of bench.py is a replay of the captured train step -- ONE hipGraph, or the per-stream graph
    segments of graph.SegmentedStep (what Model.fit uses): replaying it on a sequence of NEW batches must leave the model
    exactly where eager steps on the same batches leave it."""
    from models_amd.graph import GraphedStep, RecordedStep, SegmentedStep

    # "recorded": the launch sequence kept by libmerlin_hip.so itself and replayed by one C call (graph.RecordedStep)
    exact = mode.endswith("_deterministic")  # no float atomics in the sparse update: replay == eager BIT FOR BIT
    monkeypatch.setenv("MERLIN_HIP_DETERMINISTIC", "1" if exact else "0")
    mode = mode.replace("_deterministic", "")
    cards = {"C1": 5000, "C2": 7, "C3": 300, "C4": 50}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)

    def build():
        m = mm.DLRMModel(schema, embedding_dim=16, bottom_block=mm.MLPBlock([32, 16], device=device, seed=7),
                         top_block=mm.MLPBlock([32, 8], device=device, seed=17), device=device)
        m.output.to_call.seed = 99
        m.compile(optimizer="adagrad", learning_rate=0.05)
        return m

    g = torch.Generator().manual_seed(12)
    batches = []
    for _ in range(4):
        x, xd = _batch(schema, 256, g, device)
        batches.append((xd, torch.randint(0, 2, (256, 1), generator=g).float().to(device)))
    a, b = build(), build()
    a(batches[0][0]), b(batches[0][0])
    init = [p.data.clone() for p in a.parameters()]
    for pb, w in zip(b.parameters(), init):
        pb.data.copy_(w)

    def step(inp):
        return b.train_step({k: v for k, v in inp.items() if k != "__label__"}, inp["__label__"])

    static = dict(batches[0][0])
    static["__label__"] = batches[0][1]
    gs = {"one_graph": GraphedStep, "segmented": SegmentedStep, "recorded": RecordedStep}[mode](step, static, warmup=2)  # warm-up steps DO train b: put it back to the initial state in place
    default_cfg = not any(os.environ.get(k) for k in ("MERLIN_HIP_SIDE_STREAMS", "MERLIN_HIP_MLP_CHAIN", "MERLIN_HIP_FUSED_DLRM",
                                                       "MERLIN_HIP_SIDE_ALIAS"))  # the structure asserted below is the default's
    if mode == "recorded":
        # every launch of the step is in the recording, with the hand-offs to and from the side stream; nothing of the step ran
        # outside the library (an aten kernel would be missing from the replay)
        assert gs.n_launches >= 15 and gs.impure_ops == [] and (gs.n_hand_offs >= 4 or not default_cfg)
    if mode == "segmented" and default_cfg:
        streams = {sg["stream"] for sg in gs.segments}
        assert {"main", "sort"} <= streams and len(gs.segments) >= 5  # the step really was cut along its side work (one physical side stream: ops._SideStreams)
        assert all(d < i for i, sg in enumerate(gs.segments) for d in sg["deps"])  # edges point backwards in launch order
    for pb, w in zip(b.parameters(), init):
        pb.data.copy_(w)
        for v in pb.state.values():
            v.fill_(0.1)  # Adagrad initial_accumulator_value; the graph holds these tensors' addresses
    for xd, y in batches:
        la = a.train_step(xd, y)
        new = dict(xd)
        new["__label__"] = y
        lb = gs.replay(new)
        assert abs(float(la) - float(lb)) < 1e-6
    for pa, pb in zip(a.parameters(), b.parameters()):
        if exact:
            assert torch.equal(pb.data, pa.data)
        else:
            torch.testing.assert_close(pb.data, pa.data, atol=1e-6, rtol=1e-5)
    if mode == "segmented":
        print("segments:", [(i, sg["stream"], sg["deps"], bool(sg.get("empty"))) for i, sg in enumerate(gs.segments)])


@pytest.mark.parametrize("mode", ["one_graph", "segmented"])
def test_captured_step_survives_buffer_growth_elsewhere_and_empty_cache(device, mode):
    """A captured step addresses its workspaces and the block's padded input buffer directly.  Other work of the process -- a
    bigger batch through the same model, another model growing the shared workspaces -- replaces those buffers; the old blocks
    must stay allocated (ops.park_replaced), or the next replay writes into memory the caching allocator has handed to someone
    else, and after torch.cuda.empty_cache() dies with a GPU memory fault (bench.py's sustained region did, in segmented mode,
    after its secondary workloads had grown the workspace of the sparse-update preparation).  Checked here: every replaced
    workspace is still alive, blocks handed out afterwards are never written by the replay, and the replayed steps still train
    exactly like eager steps."""
    from models_amd import ops
    from models_amd.graph import GraphedStep, SegmentedStep

    cards = {"C1": 50000, "C2": 7, "C3": 3000, "C4": 50}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)

    def build(seed=7):
        m = mm.DLRMModel(schema, embedding_dim=16, bottom_block=mm.MLPBlock([32, 16], device=device, seed=seed),
                         top_block=mm.MLPBlock([32, 8], device=device, seed=seed + 10), device=device)
        m.output.to_call.seed = 99
        m.compile(optimizer="adagrad", learning_rate=0.05)
        return m

    g = torch.Generator().manual_seed(21)
    def batch(B):
        x, xd = _batch(schema, B, g, device)
        return xd, torch.randint(0, 2, (B, 1), generator=g).float().to(device)

    ops._WS.clear()  # start from fresh workspaces whatever earlier tests grew (their captured steps are not replayed again)
    B = 16384  # workspaces of a few MB: blocks of their own in the caching allocator
    batches = [batch(B) for _ in range(3)]
    a, b = build(), build()
    a(batches[0][0]), b(batches[0][0])
    init = [p.data.clone() for p in a.parameters()]
    for pb, w in zip(b.parameters(), init):
        pb.data.copy_(w)

    def step(inp):
        return b.train_step({k: v for k, v in inp.items() if k != "__label__"}, inp["__label__"])

    static = dict(batches[0][0])
    static["__label__"] = batches[0][1]
    gs = (GraphedStep if mode == "one_graph" else SegmentedStep)(step, static, warmup=2)
    for pb, w in zip(b.parameters(), init):
        pb.data.copy_(w)
        for v in pb.state.values():
            v.fill_(0.1)
    before = {k: (v.data_ptr(), v.numel()) for k, v in ops._WS.items()}
    marked = {k for k, v in ops._WS.items() if getattr(v, "_mh_captured", False)}  # the workspaces the captured step addresses
    parked_before = len(ops._PARKED)
    # --- the rest of the process: bigger batches grow the shared workspaces and replace b's padded top-MLP input buffer
    other = build(seed=3)
    big = batch(4 * B)
    other(big[0])
    other.train_step(*big)
    with torch.no_grad():
        b(big[0])       # same model, other batch size: its persistent input buffer is re-made
    del other, big
    torch.cuda.synchronize()
    grown = [k for k, (ptr, n) in before.items() if ops._WS[k].data_ptr() != ptr]
    assert grown, "the scenario must replace at least one workspace"
    if mode == "segmented" and os.environ.get("MERLIN_HIP_SIDE_STREAMS") is None:
        assert marked & set(grown), "... one that the captured step addresses (the sparse-update preparation's)"
    # kept: the old input buffer and every replaced workspace the captured step addresses; freed: the rest
    fused = os.environ.get("MERLIN_HIP_FUSED_DLRM", "1") != "0"  # the fused block owns the persistent padded input buffer
    assert len(ops._PARKED) - parked_before == len(marked & set(grown)) + (1 if fused else 0)
    torch.cuda.empty_cache()
    # whatever was freed is handed out again: blocks of exactly the replaced sizes, filled with a pattern the replay must not touch
    junk = [torch.full((n,), 0xAB, dtype=torch.uint8, device=device) for k in grown for n in [before[k][1]] * 3]
    junk.append(torch.full((B, 64), 7.0, device=device))
    for xd, y in batches:
        la = a.train_step(xd, y)
        new = dict(xd)
        new["__label__"] = y
        lb = gs.replay(new)
        assert abs(float(la) - float(lb)) < 1e-6
    torch.cuda.synchronize()
    assert all(bool((t == 0xAB).all()) for t in junk[:-1]) and bool((junk[-1] == 7.0).all())
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pb.data, pa.data, atol=1e-6, rtol=1e-5)
    # buffers that were never touched by a capture are freed as usual: alternating batch sizes in eager mode must not pile up
    c = build(seed=5)
    parked = len(ops._PARKED)
    with torch.no_grad():
        for _ in range(4):
            c(batches[0][0])
            c(batch(3000)[0])
    assert len(ops._PARKED) == parked


def _tt_schema():
    return mm.Schema([S.categorical("user_id", 500, [S.Tags.USER, S.Tags.USER_ID]), S.categorical("user_age", 10, [S.Tags.USER]),
                      S.categorical("item_id", 300, [S.Tags.ITEM, S.Tags.ITEM_ID]), S.categorical("item_cat", 20, [S.Tags.ITEM])])


def _tt_batch(device, B, seed):
    g = torch.Generator().manual_seed(seed)
    return {"user_id": torch.randint(0, 500, (B, 1), generator=g).to(device), "user_age": torch.randint(0, 10, (B, 1), generator=g).to(device),
            "item_id": torch.randint(0, 300, (B, 1), generator=g).to(device), "item_cat": torch.randint(0, 20, (B, 1), generator=g).to(device)}


def test_two_tower_with_popularity_logits_correction_and_cross_batch_negatives(device):
    """SURVEY 8f-4 through the mm surface: TwoTowerModel with (a) the PopularityLogitsCorrection post block -- logits
    equal the oracle's, training reduces the loss; (b) in-batch + cached cross-batch negatives -- the negative set
    grows by the cached rows of the previous batches and the step still trains."""
    torch.manual_seed(0)
    schema = _tt_schema()
    freq = torch.arange(300, 0, -1).float()
    post = mm.PopularityLogitsCorrection(freq, schema=schema, reg_factor=0.5)
    model = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device), embedding_dim=16, post_logits=post, device=device)
    model.compile(optimizer="adagrad", learning_rate=0.05)
    b = _tt_batch(device, 256, 1)
    pred = model(b, training=True)
    emb = model.body(b)
    ids = b["item_id"].reshape(-1).cpu().numpy()
    p = (freq / freq.sum()).numpy()
    want, _ = O.contrastive_outputs(emb["query"].cpu().numpy(), emb["item"].cpu().numpy(), emb["item"].cpu().numpy(), ids, ids,
                                    post_positive_prob=p[ids], post_negative_prob=p[ids], post_reg_factor=0.5)
    np.testing.assert_allclose(pred.outputs.cpu().numpy(), want, atol=1e-4)
    losses = [float(model.train_step(b)) for _ in range(8)]
    assert losses[-1] < losses[0]
    # (b)
    sampler = mm.CachedCrossBatchSampler(capacity=300)
    model2 = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device), embedding_dim=16,
                              samplers=["in-batch", sampler], device=device)
    model2.compile(optimizer="adagrad", learning_rate=0.05)
    l0 = float(model2.train_step(_tt_batch(device, 128, 2)))
    assert model2.output._neg.embedding.shape[0] == 128 if hasattr(model2.output, "_neg") else True
    l1 = float(model2.train_step(_tt_batch(device, 128, 3)))  # 128 in-batch + 128 cached
    pr = model2(_tt_batch(device, 128, 4), training=True)     # 128 in-batch + 256 cached (the pending batch landed)
    assert pr.outputs.shape == (128, 1 + 128 + 256)
    assert np.isfinite(l0) and np.isfinite(l1)
    ls = [float(model2.train_step(_tt_batch(device, 128, 2))) for _ in range(8)]
    assert ls[-1] < ls[2]  # from ls[2] on the queue is full (300 cached rows): same number of negatives, comparable losses


def test_contrastive_output_over_candidate_table_with_popularity_sampler(device):
    """ContrastiveOutput(to_call=EmbeddingTable, PopularityBasedSamplerV2, logq_sampling_correction=True)
    (tests/unit/tf/outputs/test_contrastive.py:104-133): sampled-softmax logits over the table's rows equal the oracle's
    with the sampler's probabilities."""
    torch.manual_seed(1)
    col = S.categorical("item_category", 101, [S.Tags.ITEM])
    table = mm.EmbeddingTable(16, col, device=device)
    sampler = mm.PopularityBasedSamplerV2(max_id=100, max_num_samples=20, min_id=1, seed=5)
    out = mm.ContrastiveOutput(table, negative_samplers=sampler, logq_sampling_correction=True, store_negative_ids=True)
    g = torch.Generator().manual_seed(2)
    q = torch.randn(50, 16, generator=g).to(device)
    tgt = torch.randint(1, 101, (50, 1), generator=g).to(device)
    pred = out({"query": q}, features={}, targets=tgt, training=True)
    nid = pred.negative_candidate_ids.cpu().numpy()
    assert pred.outputs.shape == (50, 21) and len(np.unique(nid)) == 20 and nid.min() >= 1
    W = table.table.data.cpu().numpy()
    dist = sampler.sampling_dist.cpu().numpy()
    t = tgt.reshape(-1).cpu().numpy()
    want, _ = O.contrastive_outputs(q.cpu().numpy(), W[t], W[nid], t, nid, positive_sampling_prob=dist[t],
                                    negative_sampling_prob=dist[nid])
    np.testing.assert_allclose(pred.outputs.cpu().numpy(), want, atol=1e-4)


def test_fit_auto_graph_and_targetless_batches(device):
    """Model.fit: (inputs, targets) batches for ranking, inputs-only batches for retrieval (no AttributeError on a
    missing target); the static-shape steps replay a captured hipGraph and give the same weights as eager steps; a
    trailing partial batch falls back to eager."""
    torch.manual_seed(0)
    schema = mm.Schema([S.categorical("a", 40), S.categorical("b", 17), S.continuous("x"), S.binary_target("y")])

    def build():
        mm.set_seed(3)  # the lazily-built output layer draws its seed from the construction counter
        m = mm.DLRMModel(schema, embedding_dim=8, bottom_block=mm.MLPBlock([8], device=device, seed=1),
                         top_block=mm.MLPBlock([8], device=device, seed=2), device=device)
        m.compile(optimizer="adagrad", learning_rate=0.05)
        return m

    def batches(sizes):
        g = torch.Generator().manual_seed(9)
        out = []
        for B in sizes:
            x = {"a": torch.randint(0, 40, (B, 1), generator=g).to(device), "b": torch.randint(0, 17, (B, 1), generator=g).to(device),
                 "x": torch.rand(B, 1, generator=g).to(device)}
            out.append((x, torch.randint(0, 2, (B, 1), generator=g).float().to(device)))
        return out

    data = batches([64, 64, 64, 64, 20])
    m1, m2 = build(), build()
    h1 = m1.fit(data, graph=None)
    h2 = m2.fit(data, graph=False)
    assert len(h1["loss"]) == 1 and h1["examples_per_sec"][0] > 0
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(p1.data, p2.data, atol=1e-6, rtol=1e-5)
    assert abs(h1["loss"][0] - h2["loss"][0]) < 1e-5
    tt = mm.TwoTowerModel(_tt_schema(), mm.MLPBlock([16], device=device), embedding_dim=8, device=device)
    tt.compile(optimizer="adagrad", learning_rate=0.05)
    hist = tt.fit([_tt_batch(device, 96, i) for i in range(4)])  # inputs only
    assert np.isfinite(hist["loss"][0]) and tt.graph_capturable
    tt2 = mm.TwoTowerModel(_tt_schema(), mm.MLPBlock([16], device=device), embedding_dim=8,
                           samplers=["in-batch", mm.CachedCrossBatchSampler(64)], device=device)
    assert not tt2.graph_capturable
    assert np.isfinite(tt2.fit([_tt_batch(device, 96, i) for i in range(3)])["loss"][0])


def test_sharded_list_features_equal_the_plain_model_on_the_hip_kernels(device):
    """The row-sharded route with list features on the HIP kernels (forced sharding on one GPU: the code an N-GPU job runs,
    exchange = identity): a ragged history sharing the item table with the one-hot item id + a dense list over another table,
    DCN-v2 body, Adagrad -- equal to the plain model step for step (tf/distributed/embedding.py:144-148)."""
    from models_amd import distributed as D

    cols = [S.categorical("item_id", 2003, domain_name="item"),
            S.categorical("item_hist", 2003, domain_name="item", is_list=True, is_ragged=True),
            S.categorical("tags", 1500, is_list=True), S.categorical("small", 7), S.continuous("I1"), S.binary_target("label")]
    schema = mm.Schema(cols)

    def build():
        m = mm.DCNModel(schema, depth=1, deep_block=mm.MLPBlock([16, 8], device=device, seed=5), embedding_dim=8, device=device)
        m.compile(optimizer="adagrad", learning_rate=0.05)
        return m

    g = torch.Generator().manual_seed(8)
    B = 300

    def batch():
        lens = torch.randint(0, 6, (B,), generator=g)
        lens[5] = 0
        offs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])
        vals = torch.randint(0, 2003, (int(offs[-1]),), generator=g)
        vals[torch.rand(vals.shape, generator=g) < 0.05] = -1
        x = {"item_id": torch.randint(0, 2003, (B, 1), generator=g).to(device), "item_hist": mm.Ragged(vals.to(device), offs.to(device)),
             "tags": torch.randint(0, 1500, (B, 3), generator=g).to(device), "small": torch.randint(0, 7, (B, 1), generator=g).to(device),
             "I1": torch.rand(B, 1, generator=g).to(device)}
        return x, torch.randint(0, 2, (B, 1), generator=g).float().to(device)

    batches = [batch() for _ in range(3)]
    a, b = build(), build()
    a(batches[0][0]), b(batches[0][0])
    for pa, pb in zip(a.parameters(), b.parameters()):
        pb.data.copy_(pa.data)
    dm = D.DistributedModel(b, shard_threshold=1000, force_shard=True)
    assert sum(len(ns) for sh in dm.shards for _, ns in sh.groups.values()) == 3  # item_id, item_hist, tags
    for x, y in batches:
        la, lb = a.train_step(x, y), dm.train_step(x, y)
        assert abs(float(la) - float(lb)) < 1e-5
    dm.check_overflow()
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pb.data, pa.data, atol=2e-6, rtol=2e-5)


def test_cross_batch_queue_step_replays_from_a_graph_once_the_queue_is_full(device):
    """in-batch + cached cross-batch negatives (blocks/sampling/queue.py): the FIFO ring's head is device state, so once the
    queue is full the train step is a fixed launch sequence -- Model.fit captures it and the replayed steps leave the model
    where eager steps leave it (the queue contents included)."""
    def build():
        mm.set_seed(11)
        m = mm.TwoTowerModel(_tt_schema(), mm.MLPBlock([16], device=device, seed=4), embedding_dim=8,
                             samplers=["in-batch", mm.CachedCrossBatchSampler(128)], device=device)
        m.compile(optimizer="adagrad", learning_rate=0.05)
        return m

    data = [_tt_batch(device, 64, 100 + i) for i in range(9)]
    m1, m2 = build(), build()
    assert not m1.graph_capturable
    h1 = m1.fit(data, graph=None)   # eager until the queue is full (step 3), captured and replayed from then on
    h2 = m2.fit(data, graph=False)
    assert m1.graph_capturable and abs(h1["loss"][0] - h2["loss"][0]) < 1e-5
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(p1.data, p2.data, atol=1e-6, rtol=1e-5)
    s1, s2 = m1.output.negative_samplers[1], m2.output.negative_samplers[1]
    assert torch.equal(s1._ids.list_all(), s2._ids.list_all())
    torch.testing.assert_close(s1._emb.list_all(), s2._emb.list_all(), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("D", [16, 8])  # 16: the fused gather -> interaction kernel; 8: the unfused pair
def test_dlrm_loads_reference_layout_weights(device, D):
    """The drop-in case: weights exported from the reference load unchanged.  The first top-MLP kernel of the reference has
    its rows ordered [bottom_block (D) | interactions (P)] (tf/blocks/dlrm.py:126-130 through tf/core/combinators.py:564-569
    and tf/core/aggregation.py:54-66; pinned by the dl_* fixtures).  The logits are checked against a statement that never
    forms the concatenated row: relu(bottom @ K[:D] + interactions @ K[D:] + b)."""
    cards = {"C1": 40, "C10": 9, "C2": 300, "Z": 5, "a": 3}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    B = 333
    model = mm.DLRMModel(mm.Schema(cols), embedding_dim=D, bottom_block=mm.MLPBlock([24, D], device=device),
                         top_block=mm.MLPBlock([32, 8], device=device), device=device)
    g = torch.Generator().manual_seed(3)
    x = {n: torch.randint(0, v, (B, 1), generator=g) for n, v in cards.items()}
    x.update({f"I{i}": torch.rand(B, 1, generator=g) for i in range(1, 4)})
    xd = {k: v.to(device) for k, v in x.items()}
    model(xd)  # build
    body = model.body
    F = len(cards) + 1
    P = F * (F - 1) // 2
    rng = np.random.default_rng(8)
    K_bottom = rng.normal(size=(D, 32)).astype(np.float32) * 0.3   # rows that multiply the bottom-MLP output
    K_inter = rng.normal(size=(P, 32)).astype(np.float32) * 0.1    # rows that multiply the pairwise dots
    top0 = body.top_block.layers[0]
    assert tuple(top0.kernel.shape) == (D + P, 32)
    top0.kernel.data.copy_(torch.from_numpy(np.concatenate([K_bottom, K_inter], axis=0)).to(device))  # the reference's row order
    p = model(xd).cpu().numpy()
    # independent statement
    cont = O.concat_features({k: v.numpy() for k, v in x.items() if k.startswith("I")})
    bottom = O.mlp(cont, [(l.kernel.numpy(), l.bias.numpy(), l.activation) for l in body.bottom_block.layers])
    feats = {n: O.embedding_lookup(body.embeddings.feature_table[n].table.numpy(), x[n].numpy()) for n in cards}
    feats["bottom_block"] = bottom
    order = sorted(feats)
    assert order == ["C1", "C10", "C2", "Z", "a", "bottom_block"]  # ASCII: upper case first
    inter = np.stack([(feats[order[i]] * feats[order[j]]).sum(-1) for i in range(F) for j in range(i + 1, F)], axis=1)
    h = np.maximum(bottom @ K_bottom + inter @ K_inter + top0.bias.numpy(), 0.0)
    h = O.mlp(h, [(l.kernel.numpy(), l.bias.numpy(), l.activation) for l in body.top_block.layers[1:]])
    hd = model.output.to_call
    ref = O.dense(h, hd.kernel.numpy(), hd.bias.numpy(), "sigmoid")
    np.testing.assert_allclose(p, ref, atol=ATOL, rtol=0)
    # and the opposite row order must NOT match (the test can tell the two layouts apart)
    top0.kernel.data.copy_(torch.from_numpy(np.concatenate([K_inter, K_bottom], axis=0)).to(device))
    assert np.abs(model(xd).cpu().numpy() - ref).max() > 1e-3


@pytest.mark.parametrize("optimizer", ["adagrad", "sgd", "adam"])
def test_pipelined_train_steps_equal_plain_steps_bit_for_bit(device, optimizer, monkeypatch):
    """Model.pipelined_updates(): the first top-MLP layer's dW GEMM and dense update of step t run at the start of step t + 1
    (blocks.DLRMBlock.backward).  Deterministic sparse update on both sides: after the flush every parameter and every optimizer
    state tensor equals the un-pipelined run BIT FOR BIT; an evaluate in the middle sees flushed weights; with Adam (step counter)
    the context changes nothing."""
    monkeypatch.setenv("MERLIN_HIP_DETERMINISTIC", "1")
    # 12 + 1 stacked features of width 64: the first top layer reads 78 + 64 = 142 columns -- too wide for the fused MLP chain, so it
    # is a layer of its own with its own dW GEMM (as the 415 -> 128 layer of configs[1] is)
    cards = {f"C{j}": v for j, v in enumerate([5000, 7, 300, 50, 1000, 3, 64, 900, 12, 4000, 33, 256], 1)}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)

    def build():
        m = mm.DLRMModel(schema, embedding_dim=64, bottom_block=mm.MLPBlock([32, 64], device=device, seed=7),
                         top_block=mm.MLPBlock([64, 16], device=device, seed=17), device=device)
        m.output.to_call.seed = 99
        m.compile(optimizer=optimizer, learning_rate=0.05)
        return m

    g = torch.Generator().manual_seed(21)
    batches = []
    for _ in range(6):
        x, xd = _batch(schema, 256, g, device)
        batches.append((xd, torch.randint(0, 2, (256, 1), generator=g).float().to(device)))
    a, b = build(), build()
    a(batches[0][0]), b(batches[0][0])
    for pb, pa in zip(b.parameters(), a.parameters()):
        pb.data.copy_(pa.data)
    dlrm = [blk for blk in b._pipeline_blocks()]
    assert len(dlrm) == 1
    deferred_seen = 0
    losses_a, losses_b = [], []
    with b.pipelined_updates():
        for i, (xd, y) in enumerate(batches):
            losses_a.append(float(a.train_step(xd, y)))
            losses_b.append(float(b.train_step(xd, y)))
            deferred_seen += getattr(dlrm[0], "_deferred", None) is not None
            if i == 2:  # a forward outside a train step flushes first
                ea, eb = a.evaluate([(xd, y)]), b.evaluate([(xd, y)])
                assert ea == eb
                assert getattr(dlrm[0], "_deferred", None) is None
    assert getattr(dlrm[0], "_deferred", None) is None and getattr(dlrm[0], "pipeline_dw", None) is None
    assert deferred_seen == (0 if optimizer == "adam" else len(batches))
    assert losses_a == losses_b
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa.data, pb.data), pa.name
        for k in pa.state:
            assert torch.equal(pa.state[k], pb.state[k]), (pa.name, k)
    # fit() inside a caller's context runs pipelined eager steps and leaves nothing behind
    with b.pipelined_updates():
        h = b.fit(batches, epochs=1, graph=False)
    a_h = a.fit(batches, epochs=1, graph=False)
    assert h["loss"] == a_h["loss"] and getattr(dlrm[0], "_deferred", None) is None
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa.data, pb.data), pa.name
