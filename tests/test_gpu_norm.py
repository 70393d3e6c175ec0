"""Dropout / BatchNormalization behind the Dense layers of an MLPBlock (tf/blocks/mlp.py:108-137) on the HIP kernels."""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import blocks, ops
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,rate", [(1000, 0.2), (4099, 0.5), (7, 0.9), (65536 * 64, 0.1)])
def test_dropout_mask_equals_the_oracle_and_backward_reuses_it(device, n, rate):
    seed = 77 + n
    st = torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g).to(device)
    dy = torch.randn(n, generator=g).to(device)
    for call in range(2):
        y = ops.dropout(x, rate, st)
        want = O.dropout(x.cpu().numpy(), rate, seed, call)
        np.testing.assert_array_equal(y.cpu().numpy(), want)        # mask bit for bit, kept values scaled by 1 / (1 - rate)
        dx = ops.dropout(dy, rate, st, backward=True)                 # the mask of the LAST forward
        keep = O.dropout_keep_mask(n, rate, seed, call)
        np.testing.assert_array_equal(dx.cpu().numpy(), np.where(keep, dy.cpu().numpy() * O.dropout_scale(rate), np.float32(0)))
    assert st.tolist() == [seed, 2, 1]
    frac = float((y != 0).float().mean())
    if n > 1000:
        assert abs(frac - (1 - rate)) < 0.03


@pytest.mark.parametrize("M,N", [(257, 128), (65536, 64), (1000, 37), (5, 512)])
def test_batchnorm_forward_backward(device, M, N):
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, N, generator=g) * 1.5 + 0.7)
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    mm0, mv0 = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5
    dy = torch.randn(M, N, generator=g) / M
    mmean, mvar = mm0.clone().to(device), mv0.clone().to(device)
    y, sm, si = ops.batchnorm(x.to(device), gamma.to(device), beta.to(device), mmean, mvar, training=True)
    wy, wm, wv = O.batchnorm_train(x.numpy(), gamma.numpy(), beta.numpy(), mm0.numpy(), mv0.numpy())
    np.testing.assert_allclose(y.cpu().numpy(), wy, atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(mmean.cpu().numpy(), wm, atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(mvar.cpu().numpy(), wv, atol=1e-6, rtol=1e-5)
    # backward against fp64 autograd of the same statement
    xt, gt, bt = x.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    yt = (xt - xt.mean(0)) / torch.sqrt(xt.var(0, unbiased=False) + 1e-3) * gt + bt
    yt.backward(dy.double())
    dx, dgamma, dbeta = ops.batchnorm_backward(x.to(device), dy.to(device), gamma.to(device), sm, si, training=True)
    torch.testing.assert_close(dx.cpu().double(), xt.grad, atol=2e-7, rtol=2e-4)
    torch.testing.assert_close(dgamma.cpu().double(), gt.grad, atol=1e-6, rtol=1e-4)
    torch.testing.assert_close(dbeta.cpu().double(), bt.grad, atol=1e-6, rtol=1e-4)
    # inference: the moving statistics normalise, nothing is updated
    before = (mmean.clone(), mvar.clone())
    yi, _, sii = ops.batchnorm(x.to(device), gamma.to(device), beta.to(device), mmean, mvar, training=False)
    np.testing.assert_allclose(yi.cpu().numpy(), O.batchnorm_infer(x.numpy(), gamma.numpy(), beta.numpy(), wm, wv), atol=1e-4, rtol=1e-5)
    assert torch.equal(mmean, before[0]) and torch.equal(mvar, before[1])
    dxi, _, _ = ops.batchnorm_backward(x.to(device), dy.to(device), gamma.to(device), mmean, sii, training=False)
    torch.testing.assert_close(dxi.cpu(), dy * gamma * sii.cpu(), atol=1e-9, rtol=1e-5)


def test_mlp_block_with_dropout_and_batch_norm_trains_like_the_statement(device):
    """mm.MLPBlock([...], dropout=, normalization="batch_norm") inside a DLRM: layer order Dense -> Dropout -> BatchNormalization
    (mlp.py:116-135), dropout only under the tape, one Adagrad step equal to torch autograd of the same statement with the
    oracle's masks."""
    from models_amd import schema as S

    mm.set_seed(5)
    blk = mm.MLPBlock([16, 8], dropout=0.25, normalization="batch_norm", device=device, seed=3)
    kinds = [type(l).__name__ for l in blk.layers]
    assert kinds == ["_Dense", "Dropout", "BatchNormalization", "_Dense", "Dropout", "BatchNormalization"]
    blk2 = mm.MLPBlock([16, 8], dropout=0.25, no_activation_last_layer=True, device=device)
    assert [type(l).__name__ for l in blk2.layers] == ["_Dense", "Dropout", "_Dense"]  # none behind the linear last layer
    with pytest.raises(ValueError):
        mm.MLPBlock([8], normalization="layer_norm")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(300, 12, generator=g).to(device)
    y_eval = blk(x)                                   # inference: no dropout, moving statistics (mean 0, var 1)
    d0, d1 = blk.layers[0], blk.layers[3]
    h = torch.relu(x @ d0.kernel.data + d0.bias.data) / np.sqrt(1 + 1e-3)
    want = torch.relu(h @ d1.kernel.data + d1.bias.data) / np.sqrt(1 + 1e-3)
    torch.testing.assert_close(y_eval, want, atol=1e-5, rtol=1e-5)
    with blocks.tape():
        y_train = blk(x)
    assert float((y_train - y_eval).abs().max()) > 1e-3
    dy = torch.randn(300, 8, generator=g).to(device)
    dx = blk.backward(dy.clone())
    # torch statement with the oracle's masks (call 0 of each dropout layer)
    xs = x.cpu().double().requires_grad_()
    W0, b0, W1, b1 = (t.data.cpu().double().requires_grad_() for t in (d0.kernel, d0.bias, d1.kernel, d1.bias))
    def bn(t):
        return (t - t.mean(0)) / torch.sqrt(t.var(0, unbiased=False) + 1e-3)
    m0 = torch.from_numpy(O.dropout_keep_mask(300 * 16, 0.25, blk.layers[1].seed, 0).reshape(300, 16))
    m1 = torch.from_numpy(O.dropout_keep_mask(300 * 8, 0.25, blk.layers[4].seed, 0).reshape(300, 8))
    h = bn(torch.relu(xs @ W0 + b0) * m0 / 0.75)
    out = bn(torch.relu(h @ W1 + b1) * m1 / 0.75)
    torch.testing.assert_close(y_train.cpu().double(), out.detach(), atol=1e-4, rtol=1e-4)
    out.backward(dy.cpu().double())
    torch.testing.assert_close(dx.cpu().double(), xs.grad, atol=1e-5, rtol=1e-3)
    torch.testing.assert_close(d0.kernel.grad.cpu().double(), W0.grad, atol=1e-5, rtol=1e-3)
    torch.testing.assert_close(d1.kernel.grad.cpu().double(), W1.grad, atol=1e-5, rtol=1e-3)
    # through a whole model: the loss goes down and the moving statistics move
    cols = [S.categorical("a", 50), S.categorical("b", 30), S.continuous("x"), S.binary_target("y")]
    model = mm.DLRMModel(mm.Schema(cols), embedding_dim=8, bottom_block=mm.MLPBlock([8], device=device),
                         top_block=mm.MLPBlock([16, 8], dropout=0.1, normalization="batch_norm", device=device), device=device)
    model.compile(optimizer="adagrad", learning_rate=0.05)
    xb = {"a": torch.randint(0, 50, (256, 1), generator=g).to(device), "b": torch.randint(0, 30, (256, 1), generator=g).to(device),
          "x": torch.rand(256, 1, generator=g).to(device)}
    yb = (xb["a"] % 2).float()
    losses = [float(model.train_step(xb, yb)) for _ in range(30)]
    assert losses[-1] < losses[0] * 0.9
    bnl = [l for l in model.body.top_block.layers if isinstance(l, mm.BatchNormalization)][0]
    assert float(bnl.moving_mean.abs().max()) > 0
    # save -> load -> eval round trip: the moving statistics are weights of the layer (Keras save_weights includes non-trainable
    # weights); a restored model must normalise with the trained statistics, not with mean 0 / variance 1
    import os
    import tempfile

    want = model(xb).cpu()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.npz")
        model.save_weights(path)
        fresh = mm.DLRMModel(mm.Schema(cols), embedding_dim=8, bottom_block=mm.MLPBlock([8], device=device),
                             top_block=mm.MLPBlock([16, 8], dropout=0.1, normalization="batch_norm", device=device), device=device)
        fresh.compile(optimizer="adagrad", learning_rate=0.05)
        fresh(xb)  # build
        assert not torch.allclose(fresh(xb).cpu(), want, atol=1e-3)
        fresh.load_weights(path)
    fbn = [l for l in fresh.body.top_block.layers if isinstance(l, mm.BatchNormalization)][0]
    assert torch.equal(fbn.moving_mean, bnl.moving_mean) and torch.equal(fbn.moving_variance, bnl.moving_variance)
    torch.testing.assert_close(fresh(xb).cpu(), want, atol=1e-6, rtol=0)
    names = [p.name for p in model.parameters()]
    assert any(n.endswith("/moving_mean") for n in names) and any(n.endswith("/moving_variance") for n in names)
    assert all(not p.trainable for p in model.parameters() if "moving_" in p.name)


@pytest.mark.parametrize("name", ["tanh", "elu", "selu", "softplus", "swish", "gelu", "leaky_relu", "relu6"])
def test_elementwise_activation_layer(device, name):
    """mh_activation forward against the oracle's Keras definitions and torch; backward against torch autograd; through an
    MLPBlock(activation=name) (Dense(linear) + Activation) forward + backward."""
    tfn = {"tanh": torch.tanh, "elu": torch.nn.functional.elu, "selu": torch.nn.functional.selu,
           "softplus": torch.nn.functional.softplus, "swish": torch.nn.functional.silu,
           "gelu": lambda t: torch.nn.functional.gelu(t), "leaky_relu": lambda t: torch.nn.functional.leaky_relu(t, 0.2),
           "relu6": torch.nn.functional.relu6}[name]
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(301, 37, generator=g) * 3).requires_grad_()
    dy = torch.randn(301, 37, generator=g)
    y = tfn(x)
    y.backward(dy)
    yd = ops.activation(x.detach().to(device), name)
    np.testing.assert_allclose(yd.cpu().numpy(), O._act(x.detach().numpy(), name), atol=2e-6, rtol=1e-5)
    torch.testing.assert_close(yd.cpu(), y.detach(), atol=2e-6, rtol=1e-5)
    dx = ops.activation(x.detach().to(device), name, dy=dy.to(device))
    torch.testing.assert_close(dx.cpu(), x.grad, atol=2e-6, rtol=1e-5)
    # as the activation of an MLPBlock: structure Dense(linear) -> Activation per layer; forward and input gradient vs torch
    blk = mm.MLPBlock([24, 8], activation=name, device=device, seed=3)
    xin = torch.randn(65, 13, generator=g)
    with blocks.tape():
        out = blk(xin.to(device))
    assert [type(l).__name__ for l in blk.layers] == ["_Dense", "Activation", "_Dense", "Activation"]
    xt = xin.clone().requires_grad_()
    h = xt
    for l in blk.layers:
        if isinstance(l, mm.Activation):
            h = tfn(h)
        else:
            h = h @ l.kernel.data.cpu() + l.bias.data.cpu()
    torch.testing.assert_close(out.cpu(), h.detach(), atol=1e-5, rtol=1e-4)
    go = torch.randn(65, 8, generator=g)
    h.backward(go)
    gin = blk.backward(go.to(device))
    torch.testing.assert_close(gin.cpu(), xt.grad, atol=1e-5, rtol=1e-4)
