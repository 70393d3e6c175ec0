"""Host logic of the fused Dense chains (blocks.mlp_forward / mlp_backward): how a run of _Dense layers is partitioned into
fused launches and single layers, and that the backward mirrors the forward's partition.  CPU only: the chain ops are
replaced by framework-op statements (tests/ops_shim-style), the kernels themselves are covered by tests/test_gpu_mlp_chain.py."""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import blocks, ops
from tests import ops_shim


def _act(x, a):
    return ops_shim._act(x, a)


@pytest.fixture()
def fake_chain(monkeypatch):
    """ops.mlp_chain / mlp_chain_backward as torch statements; 'supported' = 2-3 layers with every width <= 128."""
    calls = {"fwd": [], "bwd": []}

    def mlp_chain(x, Ws, bs, acts, outs=None):
        calls["fwd"].append([x.shape[1]] + [W.shape[1] for W in Ws])
        ys, h = [], x
        for l, (W, b, a) in enumerate(zip(Ws, bs, acts)):
            h = _act(h @ W + (0 if b is None else b), a)
            if outs is not None and outs[l] is not None:
                outs[l].copy_(h)
                h = outs[l]
            ys.append(h)
        return ys

    def mlp_chain_backward(x, Ws, ys, acts, grad, pre_masked=False, need_dx=True, need_db=None, x_activation=None):
        calls["bwd"].append([x.shape[1]] + [W.shape[1] for W in Ws])
        L = len(Ws)
        xs = [x] + list(ys[:-1])
        g = grad
        dWs, dbs = [None] * L, [None] * L
        for l in range(L - 1, -1, -1):
            dz = g if (pre_masked and l == L - 1) else ops_shim._act_grad(ys[l], g, acts[l])
            dWs[l], dbs[l] = xs[l].t() @ dz, dz.sum(0)
            g = dz @ Ws[l].t()
        dx = ops_shim._act_grad(x, g, x_activation) if need_dx else None
        return dx, dWs, dbs

    for name in ("linear", "linear_backward"):
        monkeypatch.setattr(ops, name, getattr(ops_shim, name))
    monkeypatch.setattr(ops, "mlp_chain", mlp_chain)
    monkeypatch.setattr(ops, "mlp_chain_backward", mlp_chain_backward)
    monkeypatch.setattr(blocks, "_chain_supported", lambda dims: 3 <= len(dims) <= 4 and max(dims) <= 128)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))  # the partition only fuses device tensors
    return calls


def _layers(dims, acts):
    ls = [blocks._Dense(d, activation=a, device="cpu", seed=3 + i) for i, (d, a) in enumerate(zip(dims[1:], acts))]
    return ls


@pytest.mark.parametrize("dims,acts,plan", [
    ([13, 128, 64], ["relu", "relu"], [(0, 2)]),                                   # DLRM bottom MLP
    ([415, 128, 64, 32, 1], ["relu", "relu", "relu", "sigmoid"], [(0, 1), (1, 3)]),  # top MLP + head: big layer alone, tail fused
    ([415, 128, 64, 32], ["relu", "relu", "relu"], [(0, 1), (1, 2)]),
    ([512, 256, 128], ["relu", "relu"], [(0, 1), (1, 1)]),                           # two-tower towers: nothing to fuse
    ([64, 32, 16, 8, 4], ["relu"] * 4, [(0, 3), (3, 1)]),
])
def test_partition_and_gradients(fake_chain, dims, acts, plan):
    torch.manual_seed(0)
    M = 37
    layers = _layers(dims, acts)
    x = torch.randn(M, dims[0])
    y = blocks.mlp_forward(layers, x)
    assert layers[0]._chain_plan == plan
    assert fake_chain["fwd"] == [dims[i:i + r + 1] for i, r in plan if r > 1]
    # reference: plain autograd over the same parameters
    xr = x.clone().requires_grad_()
    h = xr
    params = []
    for l, a in zip(layers, acts):
        W, b = l.kernel.data.clone().requires_grad_(), l.bias.data.clone().requires_grad_()
        params.append((W, b))
        h = _act(h @ W + b, a)
    torch.testing.assert_close(y, h.detach())
    g = torch.randn_like(y)
    h.backward(g)
    dx = blocks.mlp_backward(layers, g.clone(), need_dx=True)
    assert fake_chain["bwd"] == [dims[i:i + r + 1] for i, r in reversed(plan) if r > 1]
    torch.testing.assert_close(dx, xr.grad, atol=1e-5, rtol=1e-4)
    for l, (W, b) in zip(layers, params):
        torch.testing.assert_close(l.kernel.grad, W.grad, atol=1e-5, rtol=1e-4)
        torch.testing.assert_close(l.bias.grad, b.grad, atol=1e-5, rtol=1e-4)


def test_backward_without_matching_plan_goes_layer_by_layer(fake_chain):
    """A forward over [tail + head] followed by a backward over the tail only (plan length mismatch) must not reuse the
    fused partition."""
    dims, acts = [128, 64, 32, 1], ["relu", "relu", "sigmoid"]
    layers = _layers(dims, acts)
    x = torch.randn(9, 128)
    blocks.mlp_forward(layers, x)
    assert layers[0]._chain_plan == [(0, 3)]
    fake_chain["bwd"].clear()
    g = torch.randn(9, 32)
    blocks.mlp_backward(layers[:2], g, need_dx=True)
    assert fake_chain["bwd"] == []  # two single-layer backward calls instead


def test_last_layer_destination_is_honoured(fake_chain):
    dims, acts = [13, 128, 64], ["relu", "relu"]
    layers = _layers(dims, acts)
    stack = torch.full((11, 3, 64), -5.0)
    out = blocks.mlp_forward(layers, torch.randn(11, 13), out_last=stack[:, 1])
    assert out.data_ptr() == stack[:, 1].data_ptr()
    assert float(stack[:, 0].max()) == -5.0 and float(stack[:, 2].max()) == -5.0
