"""Tower layers (N = 128, K <= 1024, batch >= 4096) in the fp32-grade six-term split "bf16x6" (mh_tower_split.hip), the default for those
shapes: against float64 on the host and against the exact-chain fp32 kernels -- the new arithmetic must be AS CLOSE to the real product as the
fp32 chain is (that is the claim that makes it a default and not an opt-in like bf16x3)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from models_amd import ops

pytestmark = pytest.mark.gpu


def _f32_chain(fn_src: str, tmp_path, name: str):
    """the same call under MERLIN_HIP_GEMM_ARITH=f32 in a fresh process (the switch is read per call, but the library workspaces are not shared)"""
    f = tmp_path / f"{name}.pt"
    env = dict(os.environ, MERLIN_HIP_GEMM_ARITH="f32")
    subprocess.run([sys.executable, "-c", fn_src, str(f)], check=True, env=env, cwd=os.path.dirname(os.path.dirname(__file__)))
    return torch.load(f)


@pytest.mark.parametrize("K,act", [(415, "relu"), (256, None), (128, "sigmoid"), (1000, "relu"), (40, None)])
def test_tower_forward_is_fp32_grade(device, K, act, monkeypatch):
    monkeypatch.delenv("MERLIN_HIP_GEMM_ARITH", raising=False)
    g = torch.Generator().manual_seed(K)
    M, N = 8192 + 77, 128
    ld = (K + 3) // 4 * 4
    xh = torch.randn(M, ld, generator=g)
    xh[:, K:] = 0
    # a wide dynamic range: products of very different magnitudes inside one dot product
    xh[:, :K] *= torch.exp(torch.randn(M, K, generator=g) * 2)
    Wh = torch.randn(K, N, generator=g) * 0.1
    bh = torch.randn(N, generator=g)
    x, W, b = xh.to(device)[:, :K], Wh.to(device), bh.to(device)
    assert ops._tower_ok(M, K, N, x)
    y = ops.linear(x, W, b, act)
    z = xh[:, :K].double() @ Wh.double() + bh.double()
    ref = torch.relu(z) if act == "relu" else (torch.sigmoid(z) if act == "sigmoid" else z)
    scale = (xh[:, :K].double().abs() @ Wh.double().abs())  # sum_k |x_k w_k|: what a rounding error of the dot product is relative to
    err = ((y.cpu().double() - ref).abs() / (scale + 1e-30)).max().item() if act != "sigmoid" else (y.cpu().double() - ref).abs().max().item()
    # fp32 chain: ~K * 2^-24 in the worst case, ~sqrt(K) * 2^-24 typically; the six-term split must sit in the same place
    assert err < (2e-5 if act == "sigmoid" else 3e-6), err
    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", "f32")
    assert not ops._tower_ok(M, K, N, x)
    y32 = ops.linear(x, W, b, act)
    err32 = ((y32.cpu().double() - ref).abs() / (scale + 1e-30)).max().item() if act != "sigmoid" else (y32.cpu().double() - ref).abs().max().item()
    assert err <= max(4 * err32, 2e-7), (err, err32)   # as accurate as the chain (which is itself within ~1e-7 of the real product here)
    torch.testing.assert_close(y, y32, rtol=2e-5, atol=2e-5 * float(scale.max()) if act != "sigmoid" else 1e-5)


def test_tower_forward_edges(device, monkeypatch):
    """rows past a multiple of the 128-row tile, a strided output, NaN-free padding columns, no bias"""
    monkeypatch.delenv("MERLIN_HIP_GEMM_ARITH", raising=False)
    g = torch.Generator().manual_seed(1)
    M, K, N = 4096 + 1, 415, 128
    buf = torch.full((M, 416), float("nan"))
    buf[:, :K] = torch.randn(M, K, generator=g)
    buf[:, K:] = 0.0  # the product's layout keeps the alignment column zeroed (ops.zero_pad_columns)
    x = buf.to(device)[:, :K]
    W = (torch.randn(K, N, generator=g) * 0.1).to(device)
    out = torch.full((M, 200), -7.0, device=device)
    y = ops.linear(x, W, None, None, out=out[:, 8:136])
    ref = x.cpu().double() @ W.cpu().double()
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-5)
    assert torch.all(out[:, :8] == -7.0) and torch.all(out[:, 136:] == -7.0)


@pytest.mark.parametrize("K,act", [(415, "relu"), (256, None), (64, "relu")])
def test_tower_backward_dx_is_fp32_grade(device, K, act, monkeypatch):
    """dX = dz W^T of a tower layer by the A-stationary six-term kernel: against float64, and as close to it as the exact fp32 chain is;
    dW / db (still the chain kernels) unchanged bit for bit."""
    monkeypatch.delenv("MERLIN_HIP_GEMM_ARITH", raising=False)
    g = torch.Generator().manual_seed(K + 7)
    M, N = 8192 + 33, 128
    ld = (K + 3) // 4 * 4
    xh = torch.randn(M, ld, generator=g)
    xh[:, K:] = 0
    Wh = torch.randn(K, N, generator=g) * 0.1
    dyh = torch.randn(M, N, generator=g) * torch.exp(torch.randn(M, N, generator=g))
    x, W = xh.to(device)[:, :K], Wh.to(device)
    y = ops.linear(x, W, None, act)
    dzh = dyh.double() * ((y.cpu().double() > 0) if act == "relu" else 1.0)
    ref = dzh @ Wh.double().t()
    scale = dzh.abs() @ Wh.double().abs().t()
    dx, dW, db = ops.linear_backward(x, W, y, dyh.to(device), act)
    err = ((dx.cpu().double() - ref).abs() / (scale + 1e-30)).max().item()
    assert err < 3e-6, err
    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", "f32")
    dx32, dW32, db32 = ops.linear_backward(x, W, y, dyh.to(device), act)
    err32 = ((dx32.cpu().double() - ref).abs() / (scale + 1e-30)).max().item()
    assert err <= max(4 * err32, 2e-7), (err, err32)
    assert torch.equal(dW, dW32) and torch.equal(db, db32)
    if dx.shape[1] != dx.stride(0):  # the alignment column behind K stays what the caller made it (zero)
        full = torch.as_strided(dx, (M, dx.stride(0)), (dx.stride(0), 1))
        assert torch.all(full[:, K:] == 0)
