#!/bin/bash
# kernel trace of the eager DLRM train step and the timeline of one steady-state step (tools/step_timeline.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/timeline; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python bench.py --no-cpu-baseline --no-secondary --sustain 0 --steps 30 --warmup 10 --launch auto > /dev/null 2>&1
f=$(find $O/t -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f > $O/step_timeline.txt 2>&1; head -80 $O/step_timeline.txt
