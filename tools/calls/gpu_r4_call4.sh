#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 10 --sustain 0 --launch auto > $O/bench.json 2>$O/err.txt
f=$(find $O/tl -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f concat_columns 40 > $O/timeline.txt 2>&1; head -80 $O/timeline.txt
rm -f $f
