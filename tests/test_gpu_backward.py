"""Backward / optimizer HIP kernels vs a plain PyTorch fp32 CPU reference."""
import numpy as np
import pytest
import torch

from models_amd import ops
from oracle import oracle as O
from tests import torch_ref as R

pytestmark = pytest.mark.gpu



def _tol(ref, **f32_tol):
    """Tolerance of a check that pins the fp32 GEMMs; under the opt-in MERLIN_HIP_GEMM_ARITH=bf16x3 the wide layers run the 3-term
    bf16 split, whose error scales with the operands: 1e-4 of the largest reference entry (tests/test_gpu_gemm_split.py)."""
    if ops.gemm_arith() == "bf16x3":
        return dict(atol=1e-4 * float(ref.abs().max()), rtol=1e-4)
    return f32_tol


@pytest.mark.parametrize("M,K,N", [(300, 13, 128), (1000, 128, 64), (700, 415, 128), (129, 64, 32), (515, 32, 1), (65, 100, 200)])
@pytest.mark.parametrize("act", [None, "relu", "sigmoid"])
def test_linear_backward(device, M, K, N, act):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g, requires_grad=True)
    W = (torch.randn(K, N, generator=g) * 0.2).requires_grad_()
    b = (torch.randn(N, generator=g) * 0.1).requires_grad_()
    dy = torch.randn(M, N, generator=g)
    y = R.act(x @ W + b, act)
    y.backward(dy)
    yd = ops.linear(x.detach().to(device), W.detach().to(device), b.detach().to(device), act)
    dx, dW, db = ops.linear_backward(x.detach().to(device), W.detach().to(device), yd, dy.clone().to(device), act)
    tol = dict(atol=2e-4 * max(1.0, M / 256) ** 0.5, rtol=1e-4)
    torch.testing.assert_close(dx.cpu(), x.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(dW.cpu(), W.grad, **tol)
    torch.testing.assert_close(db.cpu(), b.grad, **tol)


def test_linear_backward_strided_dy_and_no_dx(device):
    g = torch.Generator().manual_seed(1)
    M, K, N = 260, 13, 64
    x = torch.randn(M, K, generator=g)
    W = (torch.randn(K, N, generator=g) * 0.2).requires_grad_()
    big = torch.randn(M, 5, N, generator=g)
    y = torch.relu(x @ W)
    y.backward(big[:, 2])
    bd = big.clone().to(device)
    yd = ops.linear(x.to(device), W.detach().to(device), None, "relu")
    dx, dW, db = ops.linear_backward(x.to(device), W.detach().to(device), yd, bd[:, 2], "relu", need_dx=False, need_db=False)
    assert dx is None and db is None
    torch.testing.assert_close(dW.cpu(), W.grad, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("F,D", [(27, 64), (5, 8), (16, 32), (17, 128), (3, 40)])
@pytest.mark.parametrize("with_tail", [False, True])
def test_dot_interaction_backward(device, F, D, with_tail):
    g = torch.Generator().manual_seed(F * D)
    B = 131
    X = torch.randn(B, F, D, generator=g, requires_grad=True)
    slot = F - 2
    tail = X[:, slot] if with_tail else None
    out = R.dot_interaction(X, tail)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    dx = ops.dot_interaction_backward(X.detach().to(device), dout.to(device), slot if with_tail else -1, D if with_tail else 0)
    torch.testing.assert_close(dx.cpu(), X.grad, atol=2e-4 * max(1.0, D / 64), rtol=1e-4)
    if with_tail:  # the appended column order: the same gradient from the swapped row
        dswap = torch.cat([dout[:, D:], dout[:, :D]], dim=1)
        dx2 = ops.dot_interaction_backward(X.detach().to(device), dswap.to(device), slot, D, tail_first=False)
        assert torch.equal(dx2, dx)


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
def test_embedding_backward_dedup_and_optimizers(device, opt, idt):
    g = torch.Generator().manual_seed(5)
    rows = [4, 1000, 37, 50000]  # tiny tables -> very long duplicate runs crossing many chunks
    B, D, S = 3001, 64, 6
    tabs = [torch.randn(r, D, generator=g) for r in rows]
    ids = [torch.randint(0, r, (B,), generator=g).to(idt) for r in rows]
    ids[1][:5] = torch.tensor([-1, rows[1], 7, 7, 7]).to(idt)  # OOR ids are skipped
    grad = torch.randn(B, S, D, generator=g)
    slots = [4, 0, 2, 5]
    lr, eps, acc0 = 0.05, 1e-7, 0.1
    exp_w, exp_acc = [], []
    for t, i, s in zip(tabs, ids, slots):
        i64 = i.long()
        ok = (i64 >= 0) & (i64 < t.shape[0])
        gsum = torch.zeros_like(t).index_add_(0, i64[ok], grad[ok][:, s])
        touched = torch.zeros(t.shape[0], dtype=torch.bool)
        touched[i64[ok]] = True
        if opt == "sgd":
            exp_w.append(t - lr * gsum)
            exp_acc.append(None)
        else:
            acc = torch.full_like(t, acc0)
            w2, a2 = R.adagrad_update(t, gsum, acc, lr, eps)
            exp_w.append(torch.where(touched[:, None], w2, t))
            exp_acc.append(torch.where(touched[:, None], a2, acc))
    dt = [t.clone().to(device) for t in tabs]
    st = [torch.full_like(t, acc0) for t in dt] if opt == "adagrad" else None
    ops.embedding_gather_backward(dt, st, [i.to(device) for i in ids], grad.to(device), [s * D for s in slots], opt, lr, eps)
    for k in range(len(rows)):
        torch.testing.assert_close(dt[k].cpu(), exp_w[k], atol=2e-4, rtol=1e-4)
        if opt == "adagrad":
            torch.testing.assert_close(st[k].cpu(), exp_acc[k], atol=2e-3, rtol=1e-4)


def test_embedding_backward_shared_table(device):
    """Two features sharing one table: their gradients are summed before the single update."""
    g = torch.Generator().manual_seed(6)
    B, D = 500, 16
    t = torch.randn(20, D, generator=g)
    ia, ib = torch.randint(0, 20, (B,), generator=g), torch.randint(0, 20, (B,), generator=g)
    grad = torch.randn(B, 2, D, generator=g)
    gsum = torch.zeros_like(t).index_add_(0, ia, grad[:, 0]).index_add_(0, ib, grad[:, 1])
    acc = torch.full_like(t, 0.1)
    w2, _ = R.adagrad_update(t, gsum, acc, 0.1)
    touched = torch.zeros(20, dtype=torch.bool)
    touched[ia] = True
    touched[ib] = True
    td = t.clone().to(device)
    sd = torch.full_like(td, 0.1)
    ops.embedding_gather_backward([td, td], [sd, sd], [ia.to(device), ib.to(device)], grad.to(device), [0, D], "adagrad", 0.1, 1e-7)
    torch.testing.assert_close(td.cpu(), torch.where(touched[:, None], w2, t), atol=2e-4, rtol=1e-4)


def test_bce_matches_keras_definition(device):
    g = torch.Generator().manual_seed(7)
    p = torch.rand(1000, 1, generator=g)
    p[:3, 0] = torch.tensor([0.0, 1.0, 0.5])
    y = torch.randint(0, 2, (1000, 1), generator=g).float()
    loss, dlogit = ops.bce(p.to(device), y.to(device))
    assert abs(loss.item() - R.keras_bce(p, y).item()) < 1e-5
    torch.testing.assert_close(dlogit.cpu(), (p - y) / 1000, atol=1e-8, rtol=1e-6)


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_dlrm_train_steps_match_torch(device, opt):
    """3 explicit train steps of mm.DLRMModel vs torch autograd + (Keras) SGD / Adagrad."""
    import models_amd as mm
    from models_amd import schema as S

    cards = {"C1": 50, "C10": 7, "C2": 1000, "a": 3}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)
    D, B, lr = 16, 257, 0.05
    model = mm.DLRMModel(schema, embedding_dim=D, bottom_block=mm.MLPBlock([32, D], device=device),
                         top_block=mm.MLPBlock([32, 8], device=device), device=device)
    model.compile(optimizer=opt, learning_rate=lr)
    g = torch.Generator().manual_seed(11)
    batches = []
    for _ in range(3):
        x = {n: torch.randint(0, v, (B, 1), generator=g) for n, v in cards.items()}
        x.update({f"I{i}": torch.rand(B, 1, generator=g) for i in range(1, 4)})
        y = torch.randint(0, 2, (B, 1), generator=g).float()
        batches.append((x, y))
    dev = lambda d: {k: v.to(device) for k, v in d.items()}
    model(dev(batches[0][0]))  # build lazy layers
    body = model.body
    tables = {n: body.embeddings.feature_table[n].table.data.cpu().clone().requires_grad_() for n in cards}
    lay = lambda blk: [(l.kernel.data.cpu().clone().requires_grad_(), l.bias.data.cpu().clone().requires_grad_(), l.activation) for l in blk.layers]
    bottom, top = lay(body.bottom_block), lay(body.top_block)
    hd = model.output.to_call
    head = (hd.kernel.data.cpu().clone().requires_grad_(), hd.bias.data.cpu().clone().requires_grad_())
    params = list(tables.values()) + [t for l in bottom + top for t in l[:2]] + list(head)
    accs = [torch.full_like(p, 0.1) for p in params]
    for x, y in batches:
        loss = model.train_step(dev(x), y.to(device))
        cat = {n: x[n] for n in cards}
        cont = {k: v for k, v in x.items() if k.startswith("I")}
        ref_loss = R.keras_bce(R.dlrm_forward(cat, cont, tables, bottom, top, head), y)
        assert abs(loss.item() - ref_loss.item()) < 1e-4
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
        with torch.no_grad():
            for k, (p, gr) in enumerate(zip(params, grads)):
                if gr is None:
                    continue
                if opt == "sgd":
                    p -= lr * gr
                else:
                    touched = (gr != 0).any(dim=-1, keepdim=True) if k < len(tables) else torch.ones_like(p, dtype=torch.bool)
                    w2, a2 = R.adagrad_update(p, gr, accs[k], lr)
                    p.copy_(torch.where(touched, w2, p))
                    accs[k] = torch.where(touched, a2, accs[k])
    for n in cards:
        torch.testing.assert_close(body.embeddings.feature_table[n].table.data.cpu(), tables[n].detach(), atol=1e-4, rtol=1e-4)
    for l, (W, b, _) in zip(body.bottom_block.layers + body.top_block.layers, bottom + top):
        torch.testing.assert_close(l.kernel.data.cpu(), W.detach(), atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(l.bias.data.cpu(), b.detach(), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(hd.kernel.data.cpu(), head[0].detach(), atol=1e-4, rtol=1e-4)


def test_dlrm_train_steps_adam_lazyadam_match_reference_formulas(device):
    """Dense tensors: keras Adam; embedding tables: LazyAdam (only touched rows move their moments),
    merlin/models/tf/blocks/optimizer.py:412-437.  3 steps vs torch autograd + the same formulas."""
    import models_amd as mm
    from models_amd import schema as S

    cards = {"C1": 50, "C2": 1000, "a": 3}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)
    D, B, lr, b1, b2, eps = 16, 200, 0.01, 0.9, 0.999, 1e-7
    model = mm.DLRMModel(schema, embedding_dim=D, bottom_block=mm.MLPBlock([32, D], device=device),
                         top_block=mm.MLPBlock([32, 8], device=device), device=device)
    model.compile(optimizer="adam", learning_rate=lr)
    g = torch.Generator().manual_seed(21)
    batches = []
    for _ in range(3):
        x = {n: torch.randint(0, v, (B, 1), generator=g) for n, v in cards.items()}
        x.update({f"I{i}": torch.rand(B, 1, generator=g) for i in range(1, 4)})
        batches.append((x, torch.randint(0, 2, (B, 1), generator=g).float()))
    dev = lambda d: {k: v.to(device) for k, v in d.items()}
    model(dev(batches[0][0]))
    body = model.body
    tables = {n: body.embeddings.feature_table[n].table.data.cpu().clone().requires_grad_() for n in cards}
    lay = lambda blk: [(l.kernel.data.cpu().clone().requires_grad_(), l.bias.data.cpu().clone().requires_grad_(), l.activation) for l in blk.layers]
    bottom, top = lay(body.bottom_block), lay(body.top_block)
    hd = model.output.to_call
    head = (hd.kernel.data.cpu().clone().requires_grad_(), hd.bias.data.cpu().clone().requires_grad_())
    params = list(tables.values()) + [t for l in bottom + top for t in l[:2]] + list(head)
    ms = [torch.zeros_like(p) for p in params]
    vs = [torch.zeros_like(p) for p in params]
    for step, (x, y) in enumerate(batches, start=1):
        loss = model.train_step(dev(x), y.to(device))
        cat = {n: x[n] for n in cards}
        cont = {k: v for k, v in x.items() if k.startswith("I")}
        ref_loss = R.keras_bce(R.dlrm_forward(cat, cont, tables, bottom, top, head), y)
        assert abs(loss.item() - ref_loss.item()) < 1e-4
        grads = torch.autograd.grad(ref_loss, params)
        lr_t = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
        with torch.no_grad():
            names = list(tables)
            for k, (p, gr) in enumerate(zip(params, grads)):
                if k < len(tables):
                    # the rows of the IndexedSlices are the LOOKED-UP rows, whatever their gradient values: a sample whose ReLUs
                    # are all dead contributes a zero gradient row, and LazyAdam still decays that row's moments and moves it
                    # (tf/blocks/optimizer.py:412-437 scatters over `indices`); `gr != 0` would call such a row untouched
                    touched = torch.zeros(p.shape[0], 1, dtype=torch.bool)
                    touched[x[names[k]].reshape(-1)] = True
                else:
                    touched = torch.ones_like(p, dtype=torch.bool)
                m2 = b1 * ms[k] + (1 - b1) * gr
                v2 = b2 * vs[k] + (1 - b2) * gr * gr
                w2 = p - lr_t * m2 / (v2.sqrt() + eps)
                ms[k] = torch.where(touched, m2, ms[k])
                vs[k] = torch.where(touched, v2, vs[k])
                p.copy_(torch.where(touched, w2, p))
    for n in cards:
        torch.testing.assert_close(body.embeddings.feature_table[n].table.data.cpu(), tables[n].detach(), atol=1e-4, rtol=1e-4)
    for l, (W, b, _) in zip(body.bottom_block.layers + body.top_block.layers, bottom + top):
        torch.testing.assert_close(l.kernel.data.cpu(), W.detach(), atol=1e-4, rtol=1e-4)


def test_lazy_adam_reproduces_the_reference_known_answers(device):
    """The two sparse LazyAdam scenarios of the reference's own suite (tests/unit/tf/blocks/test_optimizer.py):
    `test_lazy_adam_sparse` (:353-395: rows {0, 2} of a 3-row variable get gradients, row 1 must not move, three
    steps against `adam_update_numpy`) and `test_lazy_adam_sparse_repeated_indices` (:420-445: a repeated index is
    the same as its aggregated gradient).  Scalars of the reference become 4-wide rows (D % 4 == 0 on the HIP path)."""
    from models_amd import optim

    D = 4
    opt = optim.Adam(learning_rate=0.001)
    for var_np, grad_val in ((np.array([1.0, 1.0, 2.0], np.float32), 0.1), (np.array([3.0, 3.0, 4.0], np.float32), 0.01)):
        table = torch.from_numpy(np.repeat(var_np[:, None], D, 1).copy()).to(device)
        m, v = torch.zeros_like(table), torch.zeros_like(table)
        ids = torch.tensor([0, 2], dtype=torch.int32, device=device)
        grad = torch.full((2, 1, D), grad_val, device=device)
        ref, m_np, v_np = var_np.astype(np.float64).copy(), np.zeros(3), np.zeros(3)
        opt._step_dev = None
        for t in range(3):
            opt.begin_step(device)
            ops.embedding_gather_backward([table], [m], [ids], grad, [0], "adam", opt.learning_rate, opt.epsilon, [v],
                                          opt.beta_1, opt.beta_2, opt.lr_device)
            for r in (0, 2):
                ref[r], m_np[r], v_np[r] = O.adam_update(ref[r], grad_val, t, m_np[r], v_np[r])
            np.testing.assert_allclose(table.cpu().numpy(), np.repeat(ref[:, None], D, 1), rtol=1e-6, atol=1e-6)
        assert np.all(table.cpu().numpy()[1] == var_np[1])  # the untouched row never moves (lazy)
    # repeated index == aggregated gradient
    rep, agg = (torch.tensor([[1.0] * D, [2.0] * D], device=device) for _ in range(2))
    st = [torch.zeros(2, D, device=device) for _ in range(4)]
    o1, o2 = optim.Adam(), optim.Adam()
    for _ in range(3):
        o1.begin_step(device)
        ops.embedding_gather_backward([rep], [st[0]], [torch.tensor([1, 1], dtype=torch.int64, device=device)],
                                      torch.full((2, 1, D), 0.1, device=device), [0], "adam", o1.learning_rate, o1.epsilon,
                                      [st[1]], o1.beta_1, o1.beta_2, o1.lr_device)
        o2.begin_step(device)
        ops.embedding_gather_backward([agg], [st[2]], [torch.tensor([1], dtype=torch.int64, device=device)],
                                      torch.full((1, 1, D), 0.2, device=device), [0], "adam", o2.learning_rate, o2.epsilon,
                                      [st[3]], o2.beta_1, o2.beta_2, o2.lr_device)
        np.testing.assert_allclose(agg.cpu().numpy(), rep.cpu().numpy(), rtol=1e-6)


# ---- second-generation GEMM core in the backward: dX (NT) for K, N >= 256 and dW (split-M TN) for K >= 256, N >= 128 --------
# (4100, 415, 128): the DLRM top layer's backward shape at more than 4096 rows; (4133, 200, 96): narrow odd-sized layer
@pytest.mark.parametrize("M,K,N,ldx", [(513, 300, 260, 300), (700, 415, 128, 416), (1030, 512, 256, 512), (2500, 260, 388, 264),
                                       (4100, 415, 128, 416), (4133, 200, 96, 200)])
@pytest.mark.parametrize("act,x_act", [(None, None), ("relu", "relu"), ("sigmoid", "sigmoid")])
def test_wide_linear_backward(device, M, K, N, ldx, act, x_act):
    g = torch.Generator().manual_seed(M + N + K)
    x0 = torch.randn(M, K, generator=g)
    x = R.act(x0, x_act).detach().requires_grad_()  # x is the output of a layer with activation x_act
    W = (torch.randn(K, N, generator=g) * 0.1).requires_grad_()
    b = (torch.randn(N, generator=g) * 0.1).requires_grad_()
    dy = torch.randn(M, N, generator=g)
    y = R.act(x @ W + b, act)
    y.backward(dy)
    buf = torch.zeros(M, ldx)
    buf[:, :K] = x.detach()
    xd = buf.to(device)[:, :K]  # ld-padded operand (the interaction output feeding the top MLP has ld 416 for K = 415)
    yd = ops.linear(xd, W.detach().to(device), b.detach().to(device), act)
    dx, dW, db = ops.linear_backward(xd, W.detach().to(device), yd, dy.clone().to(device), act, x_activation=x_act)
    # reference dx folds the producer's derivative: d/dz_prev = (dz W^T) * x_act'(x)
    gx = x.grad
    if x_act == "relu":
        gx = gx * (x.detach() > 0)
    elif x_act == "sigmoid":
        gx = gx * x.detach() * (1 - x.detach())
    tol = dict(atol=3e-4 * max(1.0, M / 256) ** 0.5, rtol=1e-4)
    torch.testing.assert_close(dx.cpu(), gx, atol=2e-4, rtol=1e-4)
    torch.testing.assert_close(dW.cpu(), W.grad, **tol)
    torch.testing.assert_close(db.cpu(), b.grad, **tol)


@pytest.mark.parametrize("M,K,N", [(200, 512, 256), (2000, 1024, 1024), (700, 1024, 512)])
def test_linear_backward_slab_reductions(device, M, K, N):
    """dW / db when the batch is ONE slab (the GEMM writes them itself, nothing to reduce), and when a long dW is split into a
    few slabs (<= 8: the float4 slab reduction instead of the general 64-outputs-per-workgroup kernel)."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(device)
    W = (torch.randn(K, N, generator=g) * 0.05).to(device)
    dy = torch.randn(M, N, generator=g).to(device)
    dx, dW, db = ops.linear_backward(x, W, None, dy.clone(), None)
    for got, want in ((dW, x.double().T @ dy.double()), (db, dy.double().sum(0)), (dx, dy.double() @ W.double().T)):
        torch.testing.assert_close(got.double(), want, **_tol(want, atol=1e-4, rtol=1e-4))


@pytest.mark.parametrize("M,d", [(300, 20), (1000, 132), (33000, 512)])  # first-generation NT core ... second-generation core (fills the chip)
@pytest.mark.parametrize("accumulate", [False, True])
def test_cross_layer_backward_fused(device, M, d, accumulate):
    """mh_cross_layer_bwd: g = dout * x0, dx0_acc (+)= dout * p in one pass, dx = g W^T + dout (residual add in the GEMM epilogue),
    dW = x^T g, db -- against fp64 autograd of out = x0 * (x W + b) + x (tf/blocks/cross.py:188-202)."""
    g = torch.Generator().manual_seed(M + d)
    x0 = torch.randn(M, d, generator=g).to(device)
    x = torch.randn(M, d, generator=g).to(device)
    W = (torch.randn(d, d, generator=g) / d ** 0.5).to(device)
    b = (torch.randn(d, generator=g) * 0.1).to(device)
    dout = torch.randn(M, d, generator=g).to(device)
    prev = torch.randn(M, d, generator=g).to(device) if accumulate else None
    out, p = ops.cross_layer(x0, x, W, b, save_p=True)
    x064, x64, W64, b64 = (t.double().requires_grad_() for t in (x0, x, W, b))
    ref = x064 * (x64 @ W64 + b64) + x64
    torch.testing.assert_close(out.double(), ref.detach(), **_tol(ref.detach(), atol=1e-4, rtol=1e-4))
    ref.backward(dout.double())
    acc_in = None if prev is None else prev.clone()
    dx0_acc, dx, dW, db = ops.cross_layer_backward(x0, x, p, dout, W, acc_in)
    want0 = x064.grad if prev is None else x064.grad + prev.double()
    tol = dict(atol=2e-4 * max(1.0, (M / 1000) ** 0.5), rtol=1e-4)
    torch.testing.assert_close(dx0_acc.double(), want0, **_tol(want0, atol=1e-5, rtol=1e-5))
    if accumulate:
        assert dx0_acc.data_ptr() == acc_in.data_ptr()  # accumulated in place: the running sum of a CrossBlock
    torch.testing.assert_close(dx.double(), x64.grad, **_tol(x64.grad, atol=1e-4, rtol=1e-4))
    torch.testing.assert_close(dW.double(), W64.grad, **_tol(W64.grad, **tol))
    torch.testing.assert_close(db.double(), b64.grad, **_tol(b64.grad, **tol))


def test_cross_lowrank_dx_phase(device):
    g = torch.Generator().manual_seed(9)
    M, d, r = 777, 36, 8
    dh = torch.randn(M, r, generator=g).to(device)
    U = torch.randn(d, r, generator=g).to(device)
    dout = torch.randn(M, d, generator=g).to(device)
    dx = ops.cross_lowrank_dx(dh, U, dout)
    torch.testing.assert_close(dx.double(), dh.double() @ U.double().T + dout.double(), atol=1e-5, rtol=1e-5)


def test_eltwise_vector_and_scalar_paths(device):
    g = torch.Generator().manual_seed(3)
    for n in (4096 * 3, 1001):  # float4 path / scalar path (n % 4 != 0)
        a, b, c = (torch.randn(n, generator=g).to(device) for _ in range(3))
        assert torch.equal(ops.eltwise("mul", a, b), a * b)
        assert torch.equal(ops.eltwise("add", a, b), a + b)
        torch.testing.assert_close(ops.eltwise("fma", a, b, c), torch.addcmul(c, a, b), atol=1e-6, rtol=1e-6)


def test_wide_backward_equals_first_generation_core(device):
    """y and dX are bit-identical between the two GEMM cores (one k-ascending chain per output); dW sums the same slices
    of the batch in the same row order -> identical too; db groups its partial sums differently."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import torch, sys
        from models_amd import ops
        g = torch.Generator().manual_seed(5)
        out = []
        for M, K, N in ((1500, 512, 384), (4200, 416, 128)):  # the second shape: the DLRM top layer (K = 416, short contraction in dX)
            x = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(K, N, generator=g) * 0.1).cuda()
            dy = torch.randn(M, N, generator=g).cuda()
            y = ops.linear(x, W, None, "relu")
            dx, dW, db = ops.linear_backward(x, W, y, dy, "relu", x_activation="relu" if M > 4000 else None)
            out.append([y.cpu(), dx.cpu(), dW.cpu(), db.cpu()])
        torch.save(out, sys.argv[1])
    ''')
    import os, tempfile
    outs = []
    for v1 in (False, True):
        env = dict(os.environ)
        env.pop("MERLIN_HIP_GEMM_V1", None)
        if v1:
            env["MERLIN_HIP_GEMM_V1"] = "1"
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            subprocess.run([sys.executable, "-c", code, f.name], check=True, env=env, cwd=os.path.dirname(os.path.dirname(__file__)))
            outs.append(torch.load(f.name))
    for (y0, dx0, dW0, db0), (y1, dx1, dW1, db1) in zip(*outs):
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1) and torch.equal(dW0, dW1)
        torch.testing.assert_close(db0, db1, atol=1e-4, rtol=1e-5)  # column sums: 16- vs 32-row partial sums per k-tile


def test_dense_adam_reproduces_the_reference_known_answers(device):
    """The DENSE scenario of the reference's optimizer suite (tests/unit/tf/blocks/test_optimizer.py:448-487: var0 = [1, 2]
    with grads [0.1, 0.1], var1 = [3, 4] with grads [0.01, 0.01], three steps against `adam_update_numpy`) through
    mh_dense_optimizer_step -- the Keras Adam that the dense tensors of a model take (tf/models/base.py:476-508)."""
    from models_amd import optim
    from models_amd.core import Parameter

    opt = optim.Adam(learning_rate=0.001)
    scen = [(np.array([1.0, 2.0], np.float32), np.array([0.1, 0.1], np.float32)),
            (np.array([3.0, 4.0], np.float32), np.array([0.01, 0.01], np.float32))]
    params = [Parameter(torch.from_numpy(v.copy()).to(device), name=f"var{i}") for i, (v, _) in enumerate(scen)]
    ref = [(v.astype(np.float64).copy(), np.zeros(2), np.zeros(2)) for v, _ in scen]
    for t in range(3):
        opt.begin_step(device)
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(scen[i][1]).to(device)
            ops.dense_optimizer_step(opt, p)
            ref[i] = O.adam_update(ref[i][0], scen[i][1].astype(np.float64), t, ref[i][1], ref[i][2])
            np.testing.assert_allclose(p.data.cpu().numpy(), ref[i][0], rtol=1e-6, atol=1e-6)
