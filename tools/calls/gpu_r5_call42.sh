#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c42; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python bench.py --launch pipelined --no-secondary --no-cpu-baseline --steps 30 --warmup 10 --sustain 0 > $O/bench.json 2>$O/err.txt
f=$(find $O/tl -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f concat_columns 20 > $O/timeline_pipelined.txt 2>&1
rm -rf $O/tl
cat $O/timeline_pipelined.txt
tail -1 $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['launch'])"
exit 0
