"""What the de-duplicating route costs on one GPU (no exchange): mh_route_build vs mh_route_build_dedup, and the two ways the
gradient rows reach the send buffer (gather by src_row vs the segment sum over pos_of), at the shapes of configs[1] / [3]."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from models_amd import ops  # noqa: E402
from models_amd.distributed import _hip_segment_sum  # noqa: E402
from models_amd.synthetic import lognormal_ids  # noqa: E402


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    dev = torch.device("cuda:0")
    B, D, W = 65536, 64, 8
    rng = np.random.default_rng(0)
    for F in (9, 26):
        for dist in ("uniform", "lognormal"):
            ids = [torch.from_numpy((rng.integers(0, 1_000_000, size=B) if dist == "uniform" else lognormal_ids(rng, B, 999_999))
                                    .astype(np.int32)).to(dev) for _ in range(F)]
            over = torch.zeros(1, dtype=torch.int32, device=dev)
            k, p, s, c = ops.route_build(ids, W)
            kd, pd, _, cd = ops.route_build(ids, W, dedup=True)
            uniq = int(cd.sum())
            cap_plain = (int(c.max()) * 5 // 4 + 63) // 64 * 64
            cap_dedup = (int(cd.max()) * 5 // 4 + 63) // 64 * 64
            t_plain = timed(lambda: ops.route_build(ids, W, capacity=cap_plain, overflow=over))
            t_dedup = timed(lambda: ops.route_build(ids, W, capacity=cap_dedup, overflow=over, dedup=True))
            kf, pf, sf, _ = ops.route_build(ids, W, capacity=cap_plain, overflow=over)
            kdf, pdf, _, _ = ops.route_build(ids, W, capacity=cap_dedup, overflow=over, dedup=True)
            dstack = torch.randn(B, F, D, device=dev)
            t_gather = timed(lambda: ops.embedding_gather([dstack.reshape(B * F, D)], [sf]))
            t_sum = timed(lambda: _hip_segment_sum(dstack, list(range(F)), pdf, kdf.numel()))
            back_plain = torch.randn(kf.numel(), D, device=dev)
            back_dedup = torch.randn(kdf.numel(), D, device=dev)
            out = torch.empty(B, F, D, device=dev)
            t_fwd_plain = timed(lambda: ops.embedding_gather([back_plain] * F, [pf[f] for f in range(F)], out=out, out_slot=list(range(F))))
            t_fwd_dedup = timed(lambda: ops.embedding_gather([back_dedup] * F, [pdf[f] for f in range(F)], out=out, out_slot=list(range(F))))
            row = W * D * 4
            print(f"F={F:2d} {dist:9s} requests {F * B:8d} distinct {uniq:8d} ({uniq / (F * B):.2f})  window {cap_plain} -> {cap_dedup} slots; "
                  f"rows + gradient rows on the wire per rank: {2 * cap_plain * row / 1e6:.0f} -> {2 * cap_dedup * row / 1e6:.0f} MB | "
                  f"route {t_plain:.0f} -> {t_dedup:.0f} us, gradient rows into the send buffer {t_gather:.0f} (gather) -> {t_sum:.0f} us "
                  f"(segment sum), returned rows into the stack {t_fwd_plain:.0f} -> {t_fwd_dedup:.0f} us", flush=True)


if __name__ == "__main__":
    main()
