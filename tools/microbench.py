#!/usr/bin/env python
"""Per-kernel timing at the BASELINE config-2 shapes (development tool; run through gpurun)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from models_amd import ops

dev = torch.device("cuda")
B, F, D = 65536, 27, 64


def timeit(name, fn, nbytes=None, flops=None, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    extra = ""
    if nbytes:
        extra += f"  {nbytes / ms / 1e6:8.1f} GB/s"
    if flops:
        extra += f"  {flops / ms / 1e9:8.1f} TF/s"
    print(f"{name:40s} {ms * 1e3:9.1f} us{extra}", flush=True)
    return ms


which = set(sys.argv[1:])
g = torch.Generator(device="cpu").manual_seed(0)
if not which or "inter" in which:
    x = torch.randn(B, F, D, device=dev)
    tail = x[:, 5]
    P = F * (F - 1) // 2
    out = torch.empty(B, 416, device=dev)
    timeit("dot_interaction fwd", lambda: ops.dot_interaction(x, tail, out=out[:, :415]), nbytes=B * (F * D * 4 + 415 * 4))
    dout = torch.randn(B, 415, device=dev)
    timeit("dot_interaction bwd", lambda: ops.dot_interaction_backward(x, dout, 5, D), nbytes=B * (2 * F * D * 4 + 415 * 4))
if not which or "linbwd" in which:
    for (K, N) in [(415, 128), (128, 64), (64, 32), (13, 128), (32, 1)]:
        xx = torch.randn(B, (K + 3) // 4 * 4, device=dev)[:, :K]  # the product's layout: rows padded to 16 bytes
        W = torch.randn(K, N, device=dev) * 0.1
        y = ops.linear(xx, W, None, "relu" if N > 1 else None)
        dy = torch.randn(B, N, device=dev)
        timeit(f"linear fwd {K}x{N}", lambda: ops.linear(xx, W, None, "relu"), flops=2 * B * K * N, nbytes=4 * B * (K + N))
        timeit(f"linear bwd {K}x{N}", lambda: ops.linear_backward(xx, W, y, dy, "relu" if N > 1 else None), flops=4 * B * K * N,
               nbytes=4 * B * (2 * K + 2 * N))
if not which or "embbwd" in which:
    from models_amd.synthetic import CRITEO_CARDINALITIES

    tabs = [torch.rand(v, D, device=dev) for v in CRITEO_CARDINALITIES]
    st = [torch.full_like(t, 0.1) for t in tabs]
    ids = [torch.randint(0, v, (B,), dtype=torch.int32, device=dev) for v in CRITEO_CARDINALITIES]
    grad = torch.randn(B, F, D, device=dev)
    offs = [i * D for i in range(26)]
    timeit("embedding bwd adagrad (26 tables)", lambda: ops.embedding_gather_backward(tabs, st, ids, grad, offs, "adagrad", 0.01, 1e-7))
    timeit("embedding bwd sgd (26 tables)", lambda: ops.embedding_gather_backward(tabs, None, ids, grad, offs, "sgd", 0.01, 1e-7))
    outb = torch.empty(B, F, D, device=dev)
    timeit("embedding gather fwd", lambda: ops.embedding_gather(tabs, ids, out=outb), nbytes=B * 26 * (2 * D * 4 + 4))
    big = torch.rand(50_000_000, D, device=dev)  # 12.8 GB >> 256 MiB MALL
    idb = [torch.randint(0, 50_000_000, (B * 8,), dtype=torch.int32, device=dev)]
    outl = torch.empty(B * 8, 1, D, device=dev)
    timeit("gather fwd, one 12.8 GB table, 512K ids", lambda: ops.embedding_gather([big], idb, out=outl), nbytes=B * 8 * (2 * D * 4 + 4))
if "embada" in which or "embsgd" in which:  # ONE optimizer, C2 shapes: the selection the isolated rocprofv3 traces are taken on
    from models_amd.synthetic import CRITEO_CARDINALITIES

    tabs = [torch.rand(v, D, device=dev) for v in CRITEO_CARDINALITIES]
    st = [torch.full_like(t, 0.1) for t in tabs]
    ids = [torch.randint(0, v, (B,), dtype=torch.int32, device=dev) for v in CRITEO_CARDINALITIES]
    grad = torch.randn(B, F, D, device=dev)
    offs = [i * D for i in range(26)]
    if "embada" in which:
        timeit("embedding bwd adagrad (26 tables)", lambda: ops.embedding_gather_backward(tabs, st, ids, grad, offs, "adagrad", 0.01, 1e-7))
    else:
        timeit("embedding bwd sgd (26 tables)", lambda: ops.embedding_gather_backward(tabs, None, ids, grad, offs, "sgd", 0.01, 1e-7))
if "emb1m" in which:
    tabs = [torch.rand(1_000_000, D, device=dev) for _ in range(26)]
    ids = [torch.randint(0, 1_000_000, (B,), dtype=torch.int32, device=dev) for _ in range(26)]
    grad = torch.randn(B, F, D, device=dev)
    offs = [i * D for i in range(26)]
    timeit("embedding bwd sgd, 26 x 1M-row tables", lambda: ops.embedding_gather_backward(tabs, None, ids, grad, offs, "sgd", 0.01, 1e-7))
if "towers" in which:  # the tower GEMMs the MFMA-utilisation target is quoted on: DLRM top layer at M = 64 K, two-tower layers at M = 32 K
    for (M_, K, N) in [(B, 415, 128), (32768, 512, 256), (32768, 256, 256), (32768, 256, 128)]:
        xx = torch.randn(M_, (K + 3) // 4 * 4, device=dev)[:, :K]
        W = torch.randn(K, N, device=dev) * 0.1
        bb = torch.zeros(N, device=dev)
        y = ops.linear(xx, W, bb, "relu")
        dy = torch.randn(M_, N, device=dev)
        timeit(f"tower fwd {M_}x{K}x{N}", lambda: ops.linear(xx, W, bb, "relu", out=y), flops=2 * M_ * K * N)
        timeit(f"tower bwd {M_}x{K}x{N}", lambda: ops.linear_backward(xx, W, y, dy, "relu"), flops=4 * M_ * K * N)
if "fused" in which:
    from models_amd.synthetic import CRITEO_CARDINALITIES

    tabs = [torch.rand(v, D, device=dev) for v in CRITEO_CARDINALITIES]
    ids = [torch.randint(0, v, (B,), dtype=torch.int32, device=dev) for v in CRITEO_CARDINALITIES]
    dense = torch.randn(B, D, device=dev)
    stack = torch.empty(B, F, D, device=dev)
    out = torch.empty(B, 416, device=dev)
    dout = torch.randn(B, 415, device=dev)
    alg_f = B * (26 * (D * 4 + 4) + D * 4 + 415 * 4)
    def unfused_fwd():
        ops.embedding_gather(tabs, ids, out=stack)
        ops.dot_interaction(stack, dense, out=out[:, :415])
    timeit("gather + interaction fwd (unfused pair)", unfused_fwd, nbytes=alg_f)
    slot_t, slot_i = tabs + [None], ids + [None]
    timeit("fused gather->interaction fwd", lambda: ops.dlrm_interaction_fused(slot_t, slot_i, dense, out=out[:, :415]), nbytes=alg_f)
    timeit("interaction bwd (unfused, reads the stack)", lambda: ops.dot_interaction_backward(stack, dout, 26, D), nbytes=alg_f + B * F * D * 4)
    timeit("fused gather->interaction bwd", lambda: ops.dlrm_interaction_fused_backward(slot_t, slot_i, dense, dout), nbytes=alg_f + B * F * D * 4)
if "embbig" in which:
    tabs = [torch.rand(1_000_000, D, device=dev) for _ in range(26)]
    ids = [torch.randint(0, 1_000_000, (B,), dtype=torch.int32, device=dev) for _ in range(26)]
    grad = torch.randn(B, F, D, device=dev)
    offs = [i * D for i in range(26)]
    timeit("embedding bwd sgd, 26 x 1M-row tables", lambda: ops.embedding_gather_backward(tabs, None, ids, grad, offs, "sgd", 0.01, 1e-7))
    tabs = [torch.rand(16, D, device=dev) for _ in range(26)]
    ids = [torch.randint(0, 16, (B,), dtype=torch.int32, device=dev) for _ in range(26)]
    timeit("embedding bwd sgd, 26 x 16-row tables", lambda: ops.embedding_gather_backward(tabs, None, ids, grad, offs, "sgd", 0.01, 1e-7))
if "scorer" in which:
    import os

    Bs, E = 32768, 128
    q = torch.randn(Bs, E, device=dev) * 0.1
    it = torch.randn(Bs, E, device=dev) * 0.1
    ids = torch.randperm(1_000_000, device=dev)[:Bs].to(torch.int32)
    r = ops.inbatch_softmax(q, it, it, ids, ids, materialize=False)
    timeit("scorer fwd fused 32Kx32Kx128", lambda: ops.inbatch_softmax(q, it, it, ids, ids, materialize=False), flops=2 * Bs * Bs * E, iters=5)
    timeit("scorer fwd+dq 32Kx32Kx128", lambda: ops.inbatch_softmax_train(q, it, it, ids, ids), flops=4 * Bs * Bs * E, iters=5)
    timeit("scorer bwd column pass", lambda: ops.inbatch_softmax_backward(q, it, it, r.lse, ids, ids, need_dq=False), flops=4 * Bs * Bs * E, iters=5)
    timeit("scorer bwd (row + column passes)", lambda: ops.inbatch_softmax_backward(q, it, it, r.lse, ids, ids),
           flops=8 * Bs * Bs * E, iters=3)
if "topk" in which:
    N, E, Bq, k = 1_000_000, 128, 4096, 100
    c = torch.randn(N, E, device=dev)
    qq = torch.randn(Bq, E, device=dev)
    timeit("topk 4096 x 1M x 128, k=100", lambda: ops.topk_dot(qq, c, None, k), flops=2 * Bq * N * E, iters=3)
if "chain" in which:
    for dims, acts, need_dx, pre in [([13, 128, 64], ["relu", "relu"], False, False), ([128, 64, 32, 1], ["relu", "relu", "sigmoid"], True, True)]:
        xc = torch.rand(B, dims[0], device=dev)
        Ws = [(torch.rand(dims[i], dims[i + 1], device=dev) - 0.5) * 0.2 for i in range(len(dims) - 1)]
        bs = [torch.zeros(dims[i + 1], device=dev) for i in range(len(dims) - 1)]
        ys = ops.mlp_chain(xc, Ws, bs, acts)
        gr = torch.rand(B, dims[-1], device=dev)
        tag = "x".join(map(str, dims))
        timeit(f"mlp_chain fwd {tag}", lambda: ops.mlp_chain(xc, Ws, bs, acts, ys), flops=2 * B * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)))
        timeit(f"mlp_chain bwd {tag}", lambda: ops.mlp_chain_backward(xc, Ws, ys, acts, gr, pre_masked=pre, need_dx=need_dx,
                                                                      x_activation="relu" if need_dx else None),
               flops=4 * B * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)))
if "cross" in which:
    for d in (3344,):
        x0 = torch.randn(B, d, device=dev)
        W = torch.randn(d, d, device=dev) * 0.02
        bb = torch.zeros(d, device=dev)
        timeit(f"cross layer 64K x {d} x {d}", lambda: ops.cross_layer(x0, x0, W, bb), flops=2 * B * d * d, iters=3)
if "bagbwd" in which:  # the multi-hot update of bench.py's `embedding_bag` secondary (26 ragged features, nnz ~ Poisson(20)), one call
    import numpy as np
    from models_amd.synthetic import CRITEO_CARDINALITIES

    rng = np.random.default_rng(99)
    tabs = [torch.rand(int(v), D, device=dev) - 0.5 for v in CRITEO_CARDINALITIES]
    accs = [torch.full_like(t, 0.1) for t in tabs]
    vs, os_ = [], []
    for v in CRITEO_CARDINALITIES:
        lens = np.maximum(rng.poisson(20, size=B), 1)
        offs = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(lens, out=offs[1:])
        vs.append(torch.from_numpy(rng.integers(0, int(v), size=int(offs[-1])).astype(np.int32)).to(dev))
        os_.append(torch.from_numpy(offs).to(dev))
    nF = len(tabs)
    grad = torch.rand(B, nF * D, device=dev) - 0.5
    nnz = sum(int(v.numel()) for v in vs)
    uniq = sum(int(torch.unique(v).numel()) for v in vs)
    by = nF * B * (D * 4 + 8) + nnz * 4 + uniq * 4 * D * 4
    timeit(f"bag_bwd_multi adagrad ({nnz} values, {uniq} unique)", lambda: ops.embedding_bag_backward_multi(
        tabs, accs, vs, os_, grad, [f * D for f in range(nF)], "mean", optimizer="adagrad", lr=0.0), nbytes=by, iters=6)
if "tower" in which:  # the N = 128 tower layers: six-term split (default) against the exact fp32 chain
    import os
    for (K, N) in [(415, 128), (256, 128)]:
        xx = torch.randn(B, (K + 3) // 4 * 4, device=dev)[:, :K]
        W = torch.randn(K, N, device=dev) * 0.1
        bb = torch.randn(N, device=dev)
        for mode in ("", "f32"):
            if mode:
                os.environ["MERLIN_HIP_GEMM_ARITH"] = mode
            else:
                os.environ.pop("MERLIN_HIP_GEMM_ARITH", None)
            timeit(f"linear fwd {K}x{N} [{mode or 'bf16x6'}]", lambda: ops.linear(xx, W, bb, "relu"), flops=2 * B * K * N, nbytes=4 * B * (K + N))
            yy = ops.linear(xx, W, bb, "relu")
            dyy = torch.randn(B, N, device=dev)
            timeit(f"linear bwd {K}x{N} [{mode or 'bf16x6 dX'}]", lambda: ops.linear_backward(xx, W, yy, dyy, "relu"), flops=4 * B * K * N, nbytes=4 * B * (2 * K + 2 * N))
        os.environ.pop("MERLIN_HIP_GEMM_ARITH", None)
if "ttlin" in which:  # the TwoTower tower layers (512 -> 256 -> 128, 256 -> 256) at batch 64 K / 32 K under each GEMM arithmetic
    import os

    for M in (65536, 32768):
        for (K, N) in [(512, 256), (256, 256), (256, 128)]:
            xx = torch.randn(M, K, device=dev)
            W = torch.randn(K, N, device=dev) * 0.1
            bb = torch.zeros(N, device=dev)
            dy = torch.randn(M, N, device=dev)
            for mode in ("f32", "bf16x6", "bf16x3"):
                os.environ["MERLIN_HIP_GEMM_ARITH"] = mode
                y = ops.linear(xx, W, bb, "relu")
                timeit(f"M={M} linear fwd {K}x{N} [{mode}]", lambda: ops.linear(xx, W, bb, "relu"), flops=2 * M * K * N)
                timeit(f"M={M} linear bwd {K}x{N} [{mode}]", lambda: ops.linear_backward(xx, W, y, dy, "relu"), flops=4 * M * K * N)
            os.environ.pop("MERLIN_HIP_GEMM_ARITH", None)
