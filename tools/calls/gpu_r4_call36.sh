#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c36; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 400 python bench.py 2>/dev/null | grep "^{" > $O/default_line.jsonl; python -c "
import json; d=json.loads(open('$O/default_line.jsonl').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
