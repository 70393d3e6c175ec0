#!/usr/bin/env python
"""bench.py -- samples/sec of the DLRM hot path (BASELINE.json config 2) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one synthetic batch of 65 536 samples that is already
resident in HBM: 26 categorical lookups (Criteo cardinalities capped at 1 M rows, D = 64) + 13
dense features -> bottom MLP [128, 64] -> pairwise dot interaction -> top MLP [128, 64, 32] ->
sigmoid head (``--mode fwd``), plus loss, backward and the optimizer update (``--mode train``).
Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
``roofline`` (the launch that takes the most time of the step -- the fused embedding backward in train mode, HBM-bound;
``roofline_gather`` carries the multi-table gather next to it) and ``cpu_baseline`` (the numpy oracle timed on the
host cores on a bounded sample of the same workload, at N = 1).  At N > 1 large tables are row-sharded and the
step runs eagerly (RCCL all-to-all needs host-side split sizes).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def _cat_columns(extra_rows=0):
    """(name, cardinality) of the categorical features: the 26 Criteo columns, plus one big table for config C4."""
    from models_amd.synthetic import CRITEO_CARDINALITIES, CRITEO_CAT_NAMES

    cols = list(zip(CRITEO_CAT_NAMES, CRITEO_CARDINALITIES))
    if extra_rows:
        cols.append(("C27", int(extra_rows)))
    return cols


def build_model(device, emb_dim=64, seed=0, dcn=False, extra_rows=0):
    import models_amd as mm
    from models_amd import schema as S
    from models_amd.synthetic import CRITEO_CONT_NAMES

    cols = [S.categorical(n, v) for n, v in _cat_columns(extra_rows)]
    cols += [S.continuous(n) for n in CRITEO_CONT_NAMES]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)
    if dcn:
        return mm.DCNModel(schema, depth=3, deep_block=mm.MLPBlock([512, 256], device=device), embedding_dim=emb_dim,
                           device=device), schema
    model = mm.DLRMModel(schema, embedding_dim=emb_dim, bottom_block=mm.MLPBlock([128, emb_dim], device=device),
                         top_block=mm.MLPBlock([128, 64, 32], device=device), device=device)
    return model, schema


def make_batch(device, B, rank, dist="uniform", extra_rows=0):
    from models_amd.synthetic import CRITEO_CONT_NAMES, lognormal_ids

    rng = np.random.default_rng(1234 + rank)
    batch = {}
    for n, v in _cat_columns(extra_rows):
        ids = rng.integers(0, v, size=B) if dist == "uniform" else lognormal_ids(rng, B, v - 1)
        batch[n] = torch.from_numpy(ids.astype(np.int32)).to(device)
    dense = rng.random(size=(B, len(CRITEO_CONT_NAMES)), dtype=np.float32)
    for i, n in enumerate(CRITEO_CONT_NAMES):
        batch[n] = torch.from_numpy(np.ascontiguousarray(dense[:, i:i + 1])).to(device)
    label = torch.from_numpy(rng.integers(0, 2, size=(B, 1)).astype(np.float32)).to(device)
    return batch, label


def cpu_baseline(model, batch, label, B_cpu, mode, optimizer, budget_s=20.0):
    """numpy oracle ("port") of the SAME step on the host cores, bounded sample.  Works on host copies of
    the model state, so the timed GPU model is untouched."""
    from oracle import oracle as O

    body = model.body
    tables = {n: body.embeddings.feature_table[n].table.data.cpu().numpy() for n in body.cat_names}
    cat = {n: batch[n][:B_cpu].cpu().numpy() for n in body.cat_names}
    cont = {n: batch[n][:B_cpu].cpu().numpy() for n in body.continuous.features}
    y = label[:B_cpu].cpu().numpy()
    lay = lambda blk: [(l.kernel.numpy().copy(), l.bias.numpy().copy(), l.activation) for l in blk.layers]
    head = model.output.to_call
    bottom, top, hd = lay(body.bottom_block), lay(body.top_block), (head.kernel.numpy().copy(), head.bias.numpy().copy())
    fwd_args = (cat, cont, tables, bottom, top, hd)
    ref = O.dlrm_forward(*fwd_args)  # parity reference (before any CPU update)
    state = {"s": None}

    def one():
        if mode == "fwd":
            O.dlrm_forward(*fwd_args)
        else:
            _, state["s"] = O.dlrm_train_step(cat, cont, y, tables, bottom, top, hd, state["s"], optimizer, 0.01)

    one()  # first step discarded (tf/logging/callbacks.py:174-189)
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info

        cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    what = "dlrm_forward" if mode == "fwd" else f"dlrm_train_step ({optimizer})"
    return {"value": B_cpu * n / dt, "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"numpy oracle {what}, {n} steps x {B_cpu} samples (same tables/ids as the GPU batch; "
                      "BLAS threads for the GEMMs, single-threaded gather/scatter)"}, ref


MFMA_F32_PEAK_TF = 157.3  # v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md


def _time_steps(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def extra_workload(args, device):
    """Secondary BASELINE.json configurations on one GPU (reported in BASELINE.md; the driver's default
    invocation never takes this branch)."""
    import models_amd as mm
    from models_amd import ops, schema as S

    g = torch.Generator(device="cpu").manual_seed(0)
    if args.workload == "twotower":  # configs[2]: 1M-item catalogue, D=128, in-batch sampled softmax, B=32K
        B = args.batch if args.batch != 65536 else 32768
        cols = [S.categorical("user_id", 1_000_000, [S.Tags.USER, S.Tags.USER_ID]),
                S.categorical("user_city", 1000, [S.Tags.USER]), S.categorical("user_age", 100, [S.Tags.USER]),
                S.categorical("user_gender", 4, [S.Tags.USER]),
                S.categorical("item_id", 1_000_000, [S.Tags.ITEM, S.Tags.ITEM_ID]),
                S.categorical("item_category", 1000, [S.Tags.ITEM])]
        schema = mm.Schema(cols)
        model = mm.TwoTowerModel(schema, mm.MLPBlock([256, 128], device=device), embedding_dim=128, device=device)
        model.compile(optimizer=args.optimizer, learning_rate=0.01)
        batch = {c.name: torch.randint(0, int(c.int_domain.max) + 1, (B, 1), generator=g, dtype=torch.int32).to(device) for c in cols}
        batch["item_id"] = torch.randperm(1_000_000, generator=g)[:B].to(torch.int32).reshape(B, 1).to(device)  # no duplicate ids
        model(batch)
        step = (lambda: model.train_step(batch)) if args.mode == "train" else (lambda: model(batch, training=True))
        dt = _time_steps(step, args.steps, args.warmup)
        ops.TIMER.enable()
        for _ in range(3):
            step()
        km = ops.TIMER.summary()
        ops.TIMER.disable()
        ms = km["inbatch_softmax_fwd"]["avg_ms"]
        tf = 2.0 * B * B * 128 / (ms * 1e-3) / 1e12
        return {"metric": "samples/sec at batch 32K (TwoTower)", "value": B * args.steps / dt, "unit": "samples/s",
                "ms_per_step": dt / args.steps * 1e3, "config": {"workload": f"BASELINE configs[2]: TwoTower 1M-item catalogue, emb_dim=128, towers [256,128], in-batch sampled softmax, B={B}, {args.mode}", "launch": "eager"},
                "roofline": {"kernel": "scorer_kernel (q x items^T + mask + online LSE)", "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF,
                             "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF, "traffic": None, "avg_launch_ms": ms},
                "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in km.items()}}
    if args.workload == "topk":  # configs[2] retrieval: 4096 queries x 1M candidates x 128, k=100
        N, E, Bq, k = 1_000_000, 128, 4096, 100
        c = torch.randn(N, E, generator=g).to(device)
        q = torch.randn(Bq, E, generator=g).to(device)
        layer = mm.BruteForce(k).index(c)
        dt = _time_steps(lambda: layer(q), args.steps, args.warmup)
        tf = 2.0 * Bq * N * E * args.steps / dt / 1e12
        return {"metric": "queries/sec, brute-force top-100 over 1M x 128", "value": Bq * args.steps / dt, "unit": "queries/s",
                "ms_per_step": dt / args.steps * 1e3, "config": {"workload": "BASELINE configs[2] retrieval: 4096 queries x 1M candidates, emb_dim=128, k=100"},
                "roofline": {"kernel": "gemm_nt_kernel + topk_select_kernel", "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF,
                             "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF, "traffic": None}}
    if args.workload == "dcn":  # configs[4] on ONE GPU: DCN-v2 depth 3, D=128 (d = 3341), deep [512,256], B=64K
        model, schema = build_model(device, emb_dim=128, dcn=True)
        model.compile(optimizer=args.optimizer, learning_rate=0.01)
        batch, label = make_batch(device, args.batch, 0, args.ids)
        model(batch)
        step = (lambda: model.train_step(batch, label)) if args.mode == "train" else (lambda: model(batch))
        dt = _time_steps(step, args.steps, args.warmup)
        ops.TIMER.enable()
        for _ in range(2):
            step()
        km = ops.TIMER.summary()
        ops.TIMER.disable()
        d4 = 3344
        ms = km.get(f"cross_{d4}", {}).get("avg_ms")
        tf = 2.0 * args.batch * 3341 * 3341 / (ms * 1e-3) / 1e12 if ms else None
        return {"metric": "samples/sec at batch 64K (DCN-v2)", "value": args.batch * args.steps / dt, "unit": "samples/s",
                "ms_per_step": dt / args.steps * 1e3, "config": {"workload": f"BASELINE configs[4] on one GPU: DCN-v2 depth 3 (d=3341), emb_dim=128, deep [512,256], {args.mode}", "launch": "eager"},
                "roofline": {"kernel": "linear_fwd_kernel<128,128,4,2> (cross epilogue)", "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TF,
                             "unit": "TFLOP/s", "frac": (tf / MFMA_F32_PEAK_TF) if tf else None, "traffic": None, "avg_launch_ms": ms},
                "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in km.items()}}
    raise ValueError(args.workload)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["dlrm", "twotower", "topk", "dcn"], default="dlrm",
                    help="dlrm = BASELINE configs[1] (the headline metric); the others are secondary configs")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--mode", choices=["fwd", "train"], default="train",
                    help="train = fwd + BCE + bwd + optimizer update (the reference's fit() throughput)")
    ap.add_argument("--optimizer", choices=["sgd", "adagrad", "adam"], default="adagrad")
    ap.add_argument("--shard-threshold", type=int, default=200_000, help="rows >= this are row-sharded when N > 1")
    ap.add_argument("--ids", choices=["uniform", "lognormal"], default="uniform")
    ap.add_argument("--extra-table-rows", type=int, default=0,
                    help="BASELINE configs[3] (C4): add one table of this many rows (100000000 = 25.6 GB fp32 at D=64), "
                         "row-sharded over the ranks; the default 0 is the headline config C2")
    ap.add_argument("--eager", action="store_true", help="launch from Python instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16384)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from models_amd import ops

    if args.workload != "dlrm":
        res = extra_workload(args, device)
        res.update({"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "dtype": "f32", "data": "synthetic", "cpu_baseline": None})
        print(json.dumps(res))
        return

    model, schema = build_model(device, extra_rows=args.extra_table_rows)
    model.compile(optimizer=args.optimizer, learning_rate=0.01)
    batch, label = make_batch(device, args.batch, rank, args.ids, args.extra_table_rows)
    model(batch)  # builds the lazily-shaped dense layers

    from models_amd.graph import GraphedStep

    static = dict(batch)
    static["__label__"] = label
    runner = model
    force = os.environ.get("MH_FORCE_DISTRIBUTED") == "1"  # exercise the sharded code path on one GPU
    if world > 1 or force:
        from models_amd.distributed import DistributedDLRM

        # replicated small tables + row-sharded large tables (all-to-all over xGMI), dense bucket reduce
        runner = DistributedDLRM(model, shard_threshold=args.shard_threshold, force_shard=force)

    def eager(inp):
        feats = {k: v for k, v in inp.items() if k != "__label__"}
        if args.mode == "fwd":
            return runner(feats)
        return runner.train_step(feats, inp["__label__"])

    if args.eager or world > 1 or force:  # the sharded lookup needs host-side split sizes: not graph-capturable
        step = lambda: eager(static)
    else:
        graphed = GraphedStep(eager, static)  # whole step captured once into a hipGraph
        step = graphed.replay

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # distribution of single-step times (median and p10 / p90), hipEvent pair around each step, after
    # the timed region so that the event records do not perturb `value`
    n_ev = min(max(args.steps, 1), 100)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for a_ev, b_ev in evs:
        a_ev.record()
        step()
        b_ev.record()
    torch.cuda.synchronize()
    step_ms = sorted(a_ev.elapsed_time(b_ev) for a_ev, b_ev in evs)
    pct = lambda q: step_ms[min(n_ev - 1, int(q * n_ev))]
    step_stats = {"p10": pct(0.10), "p50": pct(0.50), "p90": pct(0.90), "n": n_ev, "timing": "hipEvent pair per step"}
    # per-kernel durations: the same launches issued eagerly with a hipEvent pair around each
    # C-ABI call on the launch stream (events cannot be read back from inside a replayed graph)
    ops.TIMER.enable()
    for _ in range(min(args.steps, 20)):
        eager(static)
    kernel_ms = ops.TIMER.summary()
    ops.TIMER.disable()
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
            return
    F, D, B = len(model.body.cat_names), model.body.dim, args.batch
    Fs = F + 1
    P = Fs * (Fs - 1) // 2
    # algorithmic bytes per launch, keyed by the op names of models_amd/ops.py
    alg_bytes = {
        "embedding_gather": B * (F * (D * 4 + D * 4) + F * 4),            # 13 416 B/sample at F=26, D=64, int32 ids
        "embedding_bwd": B * F * (5 * D * 4 + 4),                          # grad r + weight r/w + state r/w (+ id)
        "dot_interaction": B * (Fs * D * 4 + (P + D) * 4),
        "dot_interaction_bwd": B * (2 * Fs * D * 4 + (P + D) * 4),
    }

    try:  # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.sh; not re-measured live)
        pmc = json.load(open(ROOT / "profiles" / "r1_pmc_traffic.json"))
    except Exception:
        pmc = {}

    def hbm_roofline(name, kernel):
        ms = kernel_ms.get(name, {}).get("avg_ms")
        if not ms:
            return None
        ach = alg_bytes[name] / (ms * 1e-3) / 1e9
        traffic = pmc.get(name, {}).get("traffic_bytes") if (B == 65536 and args.ids == "uniform" and not args.extra_table_rows) else None
        return {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/r1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2)" if traffic else None,
                "algorithmic_bytes_per_launch": alg_bytes[name],
                "avg_launch_ms": ms, "timing": "hipEvent pair around the launch, eager pass after the timed region"}

    kernels = {"embedding_gather": "gather_fwd_kernel", "embedding_bwd": "mh_embedding_gather_bwd = build_keys + rocPRIM Onesweep sort (3 passes) + chunk_flags/scan + "
                                "piece_list + piece_reduce_apply_kernel (dominant) + carry_apply: ONE C-ABI launch",
               "dot_interaction": "dot_interaction_fwd_pipe_kernel", "dot_interaction_bwd": "dot_interaction_bwd_pipe_kernel"}
    dominant = max((k for k in kernels if k in kernel_ms), key=lambda k: kernel_ms[k]["avg_ms"], default=None)
    roofline = hbm_roofline(dominant, kernels[dominant]) if dominant else None
    roofline_gather = hbm_roofline("embedding_gather", kernels["embedding_gather"])
    res = {
        "metric": "samples/sec at batch 64K (DLRM)", "value": world * B * args.steps / dt, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[1]: DLRM 26 cat (Criteo cardinalities capped 1M) + 13 dense, "
                                f"emb_dim=64, bottom [128,64], top [128,64,32], {args.mode}, ids={args.ids}"
                                if not args.extra_table_rows else
                                f"BASELINE configs[3]: configs[1] + one {args.extra_table_rows}-row table (row-sharded at N > 1), "
                                f"{args.mode}, ids={args.ids}"),
                   "global_batch": world * B, "per_gpu_batch": B, "mode": args.mode,
                   "optimizer": args.optimizer if args.mode == "train" else None,
                   "launch": "eager" if (args.eager or world > 1 or force) else "hipGraph replay", "parallelism": f"dp{world}"},
        "roofline": roofline,
        "roofline_gather": roofline_gather,
        "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in kernel_ms.items()},
        "step_ms": step_stats,
    }
    # CPU baseline on rank 0 at N = 1 only: at N > 1 the tables are sharded and a forward is a collective
    if not args.no_cpu_baseline and world == 1 and not args.extra_table_rows:
        got = runner(batch)[: min(args.cpu_batch, B)].cpu().numpy()  # GPU probabilities with the CURRENT weights
        base, ref = cpu_baseline(model, batch, label, min(args.cpu_batch, B), args.mode, args.optimizer)
        res["cpu_baseline"] = base
        res["max_abs_err_vs_oracle"] = float(np.abs(got - ref["prob"]).max())
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
