#!/bin/bash
# rocprofv3 kernel trace + stats of a microbench selection; prints the per-kernel summary.  Usage: tools/prof_kernels.sh <tag> <microbench args>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- python tools/microbench.py "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(ls gpurun_out/prof_$tag/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_${tag}_kernel_stats.csv && column -s, -t "$f" | cut -c1-200 | head -40
