#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c2; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_topk_split.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -30 > $O/pytest.txt
cut -c1-300 $O/pytest.txt | tail -15
for m in split f32; do
  MERLIN_HIP_TOPK=$m timeout 300 python bench.py --workload topk --steps 8 --warmup 4 2>/dev/null | tail -1 > $O/topk_$m.json
  python - <<PY
import json
d=json.load(open("$O/topk_$m.json")); print("$m", round(d["ms_per_step"],3), d.get("dtype","")[:20], d.get("bit_identical_to_f32_pipeline"), d.get("index_split_ms"), d["roofline"]["frac"])
PY
done
for sp in 16 64; do for gr in 4 8 16; do
  MERLIN_HIP_TOPK_SPLITS=$sp MERLIN_HIP_TOPK_GROWTH=$gr timeout 300 python bench.py --workload topk --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('splits $sp growth $gr', round(d['ms_per_step'],3), d.get('bit_identical_to_f32_pipeline'))"
done; done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o topk -- python bench.py --workload topk --steps 8 --warmup 4 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-220
