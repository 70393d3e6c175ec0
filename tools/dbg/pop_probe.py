import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import models_amd as mm
from models_amd import ops, schema as S
dev = torch.device("cuda")
B, n_neg, V, E = 32768, 8192, 1_000_000, 128
col = S.categorical("item_id", V, [S.Tags.ITEM, S.Tags.ITEM_ID])
table = mm.EmbeddingTable(E, col, device=dev)
sampler = mm.PopularityBasedSamplerV2(max_id=V - 1, max_num_samples=n_neg, seed=7)
head = mm.ContrastiveOutput(table, negative_samplers=sampler, logq_sampling_correction=True)
q = (torch.randn(B, E) * 0.1).to(dev)
tgt = torch.randint(0, V - 1, (B, 1)).to(dev)
fwd = lambda: head({"query": q}, features={}, targets=tgt, training=True, materialize=False)
for _ in range(3): fwd()
torch.cuda.synchronize()
ops.TIMER.enable()
for _ in range(5): fwd()
print({k: round(v["avg_ms"], 4) for k, v in ops.TIMER.summary().items()})
ops.TIMER.disable()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): fwd()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14))
