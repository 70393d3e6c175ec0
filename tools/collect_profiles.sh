#!/bin/bash
# After `gpurun -- 'bash tools/refresh_profiles.sh'`: copy gpurun_out/refresh/* into profiles/ under the round's names (run from the repo root)
R=gpurun_out/refresh; P=${1:-r3}
for sel in embada emb1m fused; do
  hdr=$(head -1 profiles/${P}_isolated_$sel.txt)
  per=1; [ $sel != fused ] && per=2
  { echo "$hdr"; python tools/trace_summary.py $R/${sel}_kernel_trace.csv $per | grep -v "at::native"; grep -v "^W2026\|amdgpu.ids\|^$" $R/$sel.log | grep -i "us "; echo; echo "# rocprofv3 --stats (kernel_stats.csv)"; cat $R/${sel}_kernel_stats.csv; } > profiles/${P}_isolated_$sel.txt
done
for w in train twotower dcn; do h=$(head -1 profiles/${P}_bench_${w}_kernel_stats.csv); { echo "$h"; cat $R/bench_${w}_kernel_stats.csv; } > profiles/${P}_bench_${w}_kernel_stats.csv.new && mv profiles/${P}_bench_${w}_kernel_stats.csv.new profiles/${P}_bench_${w}_kernel_stats.csv; done
cp $R/pmc_traffic.json profiles/pmc_traffic.json; cp $R/pmc_traffic.json profiles/${P}_pmc_traffic.json; cp $R/pmc_traffic.txt profiles/${P}_pmc_traffic_summary.txt
cp $R/pytest_gpu.log profiles/${P}_pytest_gpu.log; cp $R/secondary.jsonl profiles/${P}_secondary_workloads.jsonl; cp $R/bench_lines.jsonl profiles/${P}_bench_lines.jsonl
python - <<'PY'
import json
from models_amd.build import source_hash
print("traffic stamp matches the sources:", json.load(open("profiles/pmc_traffic.json"))["source_hash"] == source_hash())
d = json.loads(open("profiles/r3_bench_lines.jsonl").readline())
print("default line:", round(d["ms_per_step"], 4), "ms", round(d["value"] / 1e6, 2), "M samples/s", d["config"]["launch"], "| sustained", round(d["sustained"]["ms_per_step"], 4),
      "| roofline", round(d["roofline"]["frac"], 3), "traffic MB", round(d["roofline"]["traffic"] / 1e6), "| cpu", round(d["cpu_baseline"]["value"]))
PY
