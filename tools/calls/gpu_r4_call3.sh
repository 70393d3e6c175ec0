#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c3/bench_default.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('frac','frac_survey_8d','avg_launch_ms')})
s=d['secondary']
for k,v in s.items():
    if isinstance(v,dict):
        print(k, {kk:vv for kk,vv in v.items() if kk in ('ms','ms_per_step','value','error','launch_modes_ms','warmup_ms_per_step','roofline','frac_of_peak','GBps')})
PY
