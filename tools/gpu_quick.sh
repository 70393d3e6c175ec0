#!/bin/bash
# usage: tools/gpu_quick.sh "<pytest args>" ["extra command"]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
timeout 900 python -m pytest $1 -x -q 2>&1 | tail -15
if [ -n "$2" ]; then eval "$2"; fi
