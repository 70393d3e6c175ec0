#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c17; rm -rf $O; mkdir -p $O
for mode in recorded segmented; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$mode -o t -- python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 10 --sustain 0 --launch $mode > $O/bench_$mode.json 2>$O/err_$mode.txt
f=$(find $O/tl_$mode -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f concat_columns 50 > $O/timeline_$mode.txt 2>&1
rm -f $f
done
cat $O/timeline_recorded.txt
