"""dlrm_fused_fwd / _bwd kernel time against the number of stacked features F (B = 65536, D = 64, 100 K-row tables, the last
slot the dense tail): where the F = 27 -> 28 step of the backward (0.180 -> 0.276 ms) comes from."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from models_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, D = int(os.environ.get("PROBE_B", 65536)), 64
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for F in ([int(v) for v in sys.argv[1:]] or (24, 25, 26, 27, 28, 29, 30, 31, 32)):
    tabs = [torch.rand((100_000, D), device=dev, generator=g) for _ in range(F - 1)] + [None]
    ids = [torch.randint(0, 100_000, (B, 1), dtype=torch.int32, device=dev, generator=g) for _ in range(F - 1)] + [None]
    dense = torch.rand((B, D), device=dev, generator=g)
    P = F * (F - 1) // 2
    ld = (P + D + 3) // 4 * 4
    out = torch.zeros((B, ld), device=dev)[:, :P + D]
    dout = torch.rand((B, ld), device=dev, generator=g)[:, :P + D]
    tf = timed(lambda: ops.dlrm_interaction_fused(tabs, ids, dense, append_dense=True, out=out))
    tb = timed(lambda: ops.dlrm_interaction_fused_backward(tabs, ids, dense, dout, tail_to_dense=True))
    per_f = lambda t: t / F * 1e3
    print(f"F = {F:2d}  P + D = {P + D:3d} (ld {ld})  fwd {tf * 1e3:7.1f} us ({per_f(tf):5.2f} per feature)   bwd {tb * 1e3:7.1f} us ({per_f(tb):5.2f} per feature)")
