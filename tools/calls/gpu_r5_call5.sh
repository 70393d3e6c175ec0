#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c5; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_topk_split.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -30 > $O/pytest.txt
cut -c1-300 $O/pytest.txt | tail -8
run() { timeout 300 python bench.py --workload topk --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d.get('bit_identical_to_f32_pipeline'), d['kernels_ms'])"; }
run default
for n0 in 2048; do for gr in 4 8; do
  MERLIN_HIP_TOPK_N0=$n0 MERLIN_HIP_TOPK_GROWTH=$gr run "n0 $n0 growth $gr"
done; done
for sp in 32 128; do MERLIN_HIP_TOPK_SPLITS=$sp run "splits $sp"; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o topk -- python bench.py --workload topk --steps 8 --warmup 4 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/topk_kernel_stats.csv && python - <<PY
import csv
rows=list(csv.DictReader(open("$O/topk_kernel_stats.csv")))
for r in rows[:9]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(4), f"{float(r['AverageNs'])/1e3:10.1f} us avg", f"min {float(r['MinNs'])/1e3:.1f} max {float(r['MaxNs'])/1e3:.1f}")
PY
exit 0
