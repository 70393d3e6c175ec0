#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c19; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for bm in 256 128; do
MERLIN_HIP_GEMM_SPLIT_BM=$bm timeout 900 python -m pytest tests/test_gpu_gemm_split.py -m gpu -x -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error" | head -10 | cut -c1-300
MERLIN_HIP_GEMM_SPLIT_BM=$bm MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py dcn_train 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('BM=$bm bf16x3 dcn_train', round(d['ms_per_step'],2), {k:v for k,v in d['kernels_ms'].items() if 'cross' in k}, d.get('roofline',{}).get('frac'))"
done
MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python bench.py --workload dcn --no-cpu-baseline --steps 3 --warmup 1 --sustain 0 > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/dcn_bf16x3_kernel_stats.csv && python - <<PY
import csv
rows=list(csv.DictReader(open("$O/dcn_bf16x3_kernel_stats.csv")))
for r in rows[:12]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(4), f"{float(r['AverageNs'])/1e3:10.1f} us avg", r['Percentage'])
PY
exit 0
