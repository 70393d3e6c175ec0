"""Fused chains of small Dense layers (mh_mlp_chain_fwd / _bwd) vs the layer-by-layer kernels and the oracle."""
import os

import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import ops, schema as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu

CHAINS = [
    # dims, activations
    ([13, 128, 64], ["relu", "relu"]),            # DLRM bottom MLP (C2)
    ([128, 64, 32, 1], ["relu", "relu", "sigmoid"]),  # tail of the top MLP + BinaryOutput head (C2)
    ([128, 64, 32], ["relu", "relu"]),
    ([4, 64, 32], ["relu", None]),
    ([53, 64, 32, 1], ["relu", "relu", "sigmoid"]),
    ([10, 20, 9], ["sigmoid", "relu"]),           # ragged widths: padded signature <16,32,16>
    ([64, 30, 16, 7], ["relu", "sigmoid", None]),
]


def _mk(dims, M, seed, device, ldx=None):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(M, dims[0])).astype(np.float32)
    Ws = [O.glorot_uniform(rng, dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
    bs = [(rng.normal(size=dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return x, Ws, bs, t(x), [t(w) for w in Ws], [t(b) for b in bs]


@pytest.mark.parametrize("dims,acts", CHAINS)
@pytest.mark.parametrize("M", [1, 16, 1000, 4099])
def test_chain_forward_equals_layer_by_layer(device, dims, acts, M):
    assert ops.mlp_chain_supported(dims)
    x, Ws, bs, xd, Wd, bd = _mk(dims, M, sum(dims) + M, device)
    ys = ops.mlp_chain(xd, Wd, bd, acts)
    h, ref = xd, x
    for l, (W, b, a) in enumerate(zip(Wd, bd, acts)):
        h = ops.linear(h, W, b, a)
        ref = O.dense(ref, Ws[l], bs[l], a)
        if W.shape[1] > 4:  # MFMA layers: the same k-ascending fmaf chain -> identical bits
            assert torch.equal(ys[l], h), f"layer {l}"
        else:  # N <= 4 heads use 16 partial chains in the layer-by-layer kernel
            torch.testing.assert_close(ys[l], h, atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(ys[l].cpu().numpy(), ref, atol=1e-4, rtol=1e-5)
        h = ys[l]  # keep the two paths on identical inputs below the head


def test_chain_forward_strided_input_and_destination(device):
    dims, acts = [13, 128, 64], ["relu", "relu"]
    M = 777
    x, Ws, bs, xd, Wd, bd = _mk(dims, M, 5, device)
    big = torch.zeros(M, 20, device=device)
    big[:, 3:16] = xd
    stack = torch.full((M, 5, 64), -7.0, device=device)
    ys = ops.mlp_chain(big[:, 3:16], Wd, bd, acts, outs=[None, stack[:, 2]])
    ref = ops.linear(ops.linear(xd, Wd[0], bd[0], "relu"), Wd[1], bd[1], "relu")
    assert ys[1].data_ptr() == stack[:, 2].data_ptr()
    assert torch.equal(stack[:, 2], ref)
    assert float(stack[:, 1].max()) == -7.0 and float(stack[:, 3].max()) == -7.0  # neighbours untouched


@pytest.mark.parametrize("dims,acts", CHAINS)
@pytest.mark.parametrize("M,pre_masked,need_dx,x_act", [(1000, False, True, "relu"), (4099, True, True, None),
                                                        (530, False, False, None), (16, True, False, None)])
def test_chain_backward_equals_layer_by_layer_and_autograd(device, dims, acts, M, pre_masked, need_dx, x_act):
    x, Ws, bs, xd, Wd, bd = _mk(dims, M, 3 * sum(dims) + M, device)
    if x_act == "relu":
        xd = torch.relu(xd)  # x is then the output of a relu layer: its derivative is folded into dx
    L = len(Wd)
    ys = ops.mlp_chain(xd, Wd, bd, acts)
    g = torch.from_numpy(np.random.default_rng(M).normal(size=(M, dims[-1])).astype(np.float32)).to(device) / M
    dx, dWs, dbs = ops.mlp_chain_backward(xd, Wd, ys, acts, g, pre_masked=pre_masked, need_dx=need_dx, x_activation=x_act)
    # layer by layer (dy is overwritten in place there: clone)
    grad, pm = g.clone(), pre_masked
    xs = [xd] + ys[:-1]
    ref_dW, ref_db = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        prev = acts[l - 1] if l > 0 else x_act
        grad, ref_dW[l], ref_db[l] = ops.linear_backward(xs[l], Wd[l], ys[l], grad, None if pm else acts[l],
                                                        need_dx=(l > 0) or need_dx, x_activation=prev)
        pm = prev is not None
    if need_dx:
        assert torch.equal(dx, grad)  # dX: same k-ascending chains, same masks -> identical bits
    else:
        assert dx is None
    for l in range(L):
        torch.testing.assert_close(dWs[l], ref_dW[l], atol=2e-6, rtol=2e-4)
        torch.testing.assert_close(dbs[l], ref_db[l], atol=2e-6, rtol=2e-4)
    # torch autograd on the same graph (fp64 on the host)
    xt = torch.from_numpy(x).double()
    if x_act == "relu":
        xt = xt.requires_grad_(True)
        h = torch.relu(xt)  # exact in both precisions (a comparison, no arithmetic)
    else:
        xt = xt.requires_grad_(True)
        h = xt
    Wt = [torch.from_numpy(w).double().requires_grad_(True) for w in Ws]
    bt = [torch.from_numpy(b).double().requires_grad_(True) for b in bs]
    for l in range(L):
        z = h @ Wt[l] + bt[l]
        last = l == L - 1
        a = acts[l]
        if a is None or (last and pre_masked):
            h = z
        elif a == "relu":  # the mask of the fp32 forward (a pre-activation within rounding of 0 may flip in fp64)
            h = z * (ys[l].cpu() > 0).double()
        else:
            h = torch.sigmoid(z)
    h.backward(g.cpu().double())
    for l in range(L):
        np.testing.assert_allclose(dWs[l].cpu().numpy(), Wt[l].grad.numpy(), atol=1e-5, rtol=1e-3)
        np.testing.assert_allclose(dbs[l].cpu().numpy(), bt[l].grad.numpy(), atol=1e-5, rtol=1e-3)
    if need_dx:
        np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), atol=1e-6, rtol=1e-3)


def test_unsupported_chains_are_refused(device):
    assert not ops.mlp_chain_supported([415, 128, 64])  # first width > 128: the big layer stays on the tiled GEMM
    assert not ops.mlp_chain_supported([128, 64])       # a single layer gains nothing
    assert not ops.mlp_chain_supported([512, 256, 128])
    x = torch.zeros(8, 415, device=device)
    with pytest.raises(Exception, match="no fused kernel"):
        ops.mlp_chain(x, [torch.zeros(415, 128, device=device), torch.zeros(128, 64, device=device)], [None, None],
                      ["relu", "relu"])


def _dlrm(device, chain: bool):
    os.environ["MERLIN_HIP_MLP_CHAIN"] = "1" if chain else "0"
    mm.set_seed(11)
    names = [f"C{i}" for i in range(1, 9)]
    cols = [S.categorical(n, 50 + 13 * i) for i, n in enumerate(names)] + [S.continuous(f"I{i}") for i in range(1, 14)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)
    model = mm.DLRMModel(schema, embedding_dim=64, bottom_block=mm.MLPBlock([128, 64], device=device),
                         top_block=mm.MLPBlock([128, 64, 32], device=device), device=device)
    model.compile(optimizer="adagrad", learning_rate=0.05)
    return model, cols


def test_dlrm_train_steps_with_and_without_fused_chains(device):
    """The C2-shaped model (bottom 13->128->64, top ->128->64->32, head) trained with the fused chains equals the
    layer-by-layer path to float-reassociation error in dW only."""
    B = 1000
    g = torch.Generator().manual_seed(3)
    try:
        res = []
        for chain in (False, True):
            model, cols = _dlrm(device, chain)
            x = {c.name: torch.randint(0, int(c.int_domain.max) + 1, (B, 1), generator=torch.Generator().manual_seed(1)).to(device)
                 for c in cols[:8]}
            x.update({f"I{i}": torch.rand(B, 1, generator=torch.Generator().manual_seed(100 + i)).to(device) for i in range(1, 14)})
            y = torch.randint(0, 2, (B, 1), generator=torch.Generator().manual_seed(2)).float().to(device)
            losses = [float(model.train_step(x, y)) for _ in range(3)]
            p = model(x).cpu().numpy()
            res.append((losses, p, [q.data.clone() for q in model.parameters()]))
            if chain:  # the fused path really ran
                assert model.body.bottom_block.layers[0]._chain_plan == [(0, 2)]
                assert model.body.top_block.layers[0]._chain_plan == [(0, 1), (1, 3)]
        (l0, p0, w0), (l1, p1, w1) = res
        np.testing.assert_allclose(l0, l1, rtol=1e-5)
        np.testing.assert_allclose(p0, p1, atol=1e-5)
        for a, b in zip(w0, w1):
            torch.testing.assert_close(a, b, atol=1e-5, rtol=1e-4)
    finally:
        os.environ.pop("MERLIN_HIP_MLP_CHAIN", None)
