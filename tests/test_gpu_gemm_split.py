"""The DCN-v2 cross layer and wide Dense layers on the split-bf16 GEMM (mh_cross_layer_fwd_split / _bwd_split, mh_linear_bias_act_*_split) in
both of its arithmetics -- the DEFAULT six-term bf16x6 (fp32-grade: as close to float64 as the exact fp32 chain) and the opt-in three-term bf16x3
(1e-4 of the scale) -- against the exact-fp32 kernels on the same inputs and against float64 (Cross.call and its gradients,
tf/blocks/cross.py:188-202)."""
import numpy as np
import pytest
import torch

from models_amd import ops

pytestmark = pytest.mark.gpu


def _check(arith, n, got, ref, w):
    scale = float(w.abs().max())
    if arith == "bf16x3":
        # bf16x3 against float64: 1e-4 of the largest entry (dW sums M products per entry: its fp32 accumulation error alone is ~1e-5 of scale)
        torch.testing.assert_close(got, w, atol=1e-4 * scale, rtol=1e-4, msg=lambda m, n=n: f"{n} vs float64: {m}")
        return
    # bf16x6 is fp32-grade: never further from float64 than a few times what the exact fp32 chain itself is
    e6, e32 = float((got - w).abs().max()), float((ref - w).abs().max())
    assert e6 <= max(4 * e32, 2e-6 * scale), (n, e6, e32, scale)


@pytest.mark.parametrize("arith", ["bf16x6", "bf16x3"])
@pytest.mark.parametrize("M,d", [(512, 128), (1000, 200), (700, 3344), (4096, 448)])
def test_cross_layer_split_matches_fp32_and_float64(device, monkeypatch, M, d, arith):
    g = torch.Generator().manual_seed(M + d)
    x0 = torch.randn(M, d, generator=g)
    x = torch.randn(M, d, generator=g)
    W = torch.randn(d, d, generator=g) * (1.0 / np.sqrt(d))
    b = torch.randn(d, generator=g) * 0.1
    dout = torch.randn(M, d, generator=g)
    acc0 = torch.randn(M, d, generator=g)
    dev = lambda t: t.to(device)

    def run():
        out, p = ops.cross_layer(dev(x0), dev(x), dev(W), dev(b), save_p=True)
        out2 = ops.cross_layer(dev(x0), dev(x), dev(W), dev(b))
        dx0, dx, dW, db = ops.cross_layer_backward(dev(x0), dev(x), p, dev(dout), dev(W), dev(acc0).clone())
        dx0n, _, _, _ = ops.cross_layer_backward(dev(x0), dev(x), p, dev(dout), dev(W))
        return [t.cpu().double() for t in (out, p, out2, dx0, dx, dW, db, dx0n)]

    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", "f32")
    f32 = run()
    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", arith)
    sp = run()
    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", "f32")
    assert any(not torch.equal(a, b_) for a, b_ in zip(f32, sp)), "the switch did not change the arithmetic"
    x064, x64, W64, b64, d64 = (t.double() for t in (x0, x, W, b, dout))
    p64 = x64 @ W64 + b64
    out64 = x064 * p64 + x64
    g64 = d64 * x064
    want = [out64, p64, out64, acc0.double() + d64 * p64, g64 @ W64.T + d64, x64.T @ g64, g64.sum(0), d64 * p64]
    names = ("out", "p", "out (no p)", "dx0_acc", "dx", "dW", "db", "dx0 (fresh)")
    for n, got, ref, w in zip(names, sp, f32, want):
        scale = float(w.abs().max())
        _check(arith, n, got, ref, w)
        torch.testing.assert_close(got, ref, atol=1e-4 * scale, rtol=1e-4, msg=lambda m, n=n: f"{n} vs fp32 kernels: {m}")


@pytest.mark.parametrize("arith", ["bf16x6", "bf16x3"])
@pytest.mark.parametrize("M,K,N,act,x_act", [(1536, 3341, 512, "relu", None), (1100, 1024, 768, "relu", "relu"), (2048, 1200, 600, None, None),
                                              (1024, 1000, 640, "sigmoid", "relu")])
def test_dense_layer_split_matches_fp32_and_float64(device, monkeypatch, M, K, N, act, x_act, arith):
    """Wide Dense layers on the split-bf16 GEMM (mh_linear_bias_act_fwd_split / _bwd_split): y, dz, dx (with the
    producer's activation mask), dW, db against float64 and against the exact-fp32 kernels; ragged K (3341 = the DCN-v2 tower input)."""
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    if x_act == "relu":
        x = x.clamp_min(0.0)  # x is the producer's activated output
    W = torch.randn(K, N, generator=g) * (1.0 / np.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(M, N, generator=g)
    dev = lambda t: t.to(device)

    def run():
        y = ops.linear(dev(x), dev(W), dev(b), act)
        dyc = dev(dy).clone()
        dx, dW, db = ops.linear_backward(dev(x), dev(W), y, dyc, act, True, True, x_act)
        return [t.cpu().double() for t in (y, dyc, dx, dW, db)]

    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", "f32")
    f32 = run()
    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", arith)
    assert ops._linear_split_ok(M, K, N)
    sp = run()
    monkeypatch.setenv("MERLIN_HIP_GEMM_ARITH", "f32")
    assert any(not torch.equal(a, b_) for a, b_ in zip(f32, sp)), "the switch did not change the arithmetic"
    x64, W64, b64, d64 = (t.double() for t in (x, W, b, dy))
    z64 = x64 @ W64 + b64
    if act == "relu":
        y64, dz64 = z64.clamp_min(0.0), d64 * (z64 > 0)
    elif act == "sigmoid":
        y64 = torch.sigmoid(z64)
        dz64 = d64 * y64 * (1 - y64)
    else:
        y64, dz64 = z64, d64
    # a pre-activation within the arithmetic's error of zero lands on either side of the relu gate, and ONE flipped gate moves a whole
    # row of dx by dy * W[:, n]: the GEMMs of the backward are checked on the dz this run produced (its gates), dz itself where |z| is clear
    dz_run = sp[1] if act == "relu" else dz64
    dx64 = dz_run @ W64.T
    if x_act == "relu":
        dx64 = dx64 * (x64 > 0)
    want = [y64, dz64, dx64, x64.T @ dz_run, dz_run.sum(0)]
    for n, got, ref, w in zip(("y", "dz", "dx", "dW", "db"), sp, f32, want):
        scale = float(w.abs().max())
        if n == "dz" and act == "relu":
            clear = (z64.abs() > 1e-4)
            got, ref, w = got * clear, ref * clear, w * clear
        if arith == "bf16x3" or act == "sigmoid" or n in ("dz",):
            # (sigmoid: both kernels use the fast exponential; dz at a relu gate within the arithmetic's error of zero: masked above)
            torch.testing.assert_close(got, w, atol=1e-4 * scale, rtol=1e-4, msg=lambda m, n=n: f"{n} vs float64: {m}")
        else:
            _check(arith, n, got, f32[("y", "dz", "dx", "dW", "db").index(n)] if act != "relu" or n == "y" else got, w)
        if act != "relu" or n in ("y", "dz"):
            torch.testing.assert_close(got, ref, atol=1e-4 * scale, rtol=1e-4, msg=lambda m, n=n: f"{n} vs fp32 kernels: {m}")
