"""Fixed vs per-strip cost of the fused Dense chains: time forward / backward at several batch sizes.
usage (GPU box): python tools/exp/chain_scaling.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from models_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for dims, acts, need_dx, pre in [([13, 128, 64], ["relu", "relu"], False, False),
                                 ([128, 64, 32, 1], ["relu", "relu", "sigmoid"], True, True)]:
    for M in [int(m) for m in os.environ.get("CHAIN_M", "4096,16384,32768,65536,131072,262144").split(",")]:
        g = torch.Generator(device="cpu").manual_seed(0)
        x = torch.rand(M, dims[0], generator=g).to(dev)
        Ws = [(torch.rand(dims[i], dims[i + 1], generator=g) - 0.5).to(dev) for i in range(len(dims) - 1)]
        bs = [torch.zeros(dims[i + 1], device=dev) for i in range(len(dims) - 1)]
        ys = ops.mlp_chain(x, Ws, bs, acts)
        outs = [torch.empty_like(y) for y in ys]
        gr = torch.rand(M, dims[-1], generator=g).to(dev)
        tf = timeit(lambda: ops.mlp_chain(x, Ws, bs, acts, outs))
        tb = timeit(lambda: ops.mlp_chain_backward(x, Ws, ys, acts, gr, pre_masked=pre, need_dx=need_dx, x_activation="relu" if need_dx else None))
        print(f"{'x'.join(map(str, dims)):>16s} M={M:7d}  fwd {tf:7.1f} us   bwd {tb:7.1f} us", flush=True)
