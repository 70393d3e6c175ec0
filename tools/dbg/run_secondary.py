"""Run ONE secondary of bench.py by name and print its object (GPU box; tools/calls scripts).
usage: python tools/dbg/run_secondary.py embedding_bag | fit_from_parquet | topk | topk_f32 [key=value ...]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1]
kw = {k: (int(v) if v.lstrip("-").isdigit() else float(v) if v.replace(".", "", 1).isdigit() else v)
      for k, v in (a.split("=", 1) for a in sys.argv[2:])}
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = argparse.Namespace(batch=65536, optimizer="adagrad", mode="train", ids="uniform", shard_threshold=200_000, tt_batch=32768,
                          batches=8, eager=False, steps=20, warmup=5, sustain=0.0)
tm = bench.Timing(1, device)
fns = {"embedding_bag": lambda: bench.run_embedding_bag(device, **kw),
       "fit_from_parquet": lambda: bench.run_fit_from_parquet(args, device, **kw),
       "topk": lambda: bench.run_topk(args, device, steps=8, warmup=4),
       "topk_f32": lambda: bench.run_topk(args, device, steps=6, warmup=4, mode="f32"),
       "c4_one_gpu": lambda: bench.run_c4_one_gpu(args, device, tm, **kw),
       "dcn_train": lambda: bench.run_dcn(argparse.Namespace(**dict(vars(args), steps=6, warmup=2, batches=2)), device, tm),
       "cross_gemm": lambda: bench.run_cross_gemm(device, **kw),
       "twotower": lambda: bench.run_twotower(args, device, tm, steps=20, warmup=3, sustain=0.0, **kw)}
print(json.dumps(fns[name](), indent=None))
