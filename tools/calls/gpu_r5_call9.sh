#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c9; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python -m pytest tests/test_gpu_compat.py tests/test_gpu_mlp_chain.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -12 | cut -c1-400
ab() { timeout 200 python bench.py --no-cpu-baseline --no-secondary --sustain 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); lp=d['config']['launch_probe']
print('$1', 'step', round(d['ms_per_step'],4), d['config']['launch'][:12], 'graph', round(lp['hipGraph_replay_ms'],4), 'seg', round(lp.get('segmented_replay_ms',0),4), 'rec', round(lp.get('recorded_replay_ms',0),4), 'eager', round(lp['eager_side_streams_ms'],4))"; }
for rep in 1 2; do
for late in 0 1; do for res in 5 4 3; do
  MERLIN_HIP_DW_LATE=$late MERLIN_HIP_APPLY_RESIDENT=$res ab "late=$late resident=$res"
done; done; done
for v in "1 4" "1 5"; do set -- $v
MERLIN_HIP_DW_LATE=$1 MERLIN_HIP_APPLY_RESIDENT=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python bench.py --eager --no-secondary --no-cpu-baseline --steps 30 --warmup 10 --sustain 0 > $O/bench.json 2>$O/err.txt
f=$(find $O/tl -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f concat_columns 20 > $O/timeline_late$1_res$2.txt 2>&1
rm -rf $O/tl
cat $O/timeline_late$1_res$2.txt
done
exit 0
