#!/bin/bash
# A/B of MERLIN_HIP_SORT_LEAN on one box (tools/microbench.py embbwd), alternating
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 0 1 0 1 0 1; do echo -n "lean $m: "; MERLIN_HIP_SORT_LEAN=$m timeout 25 python tools/microbench.py embbwd 2>&1 | grep "embedding bwd" | awk '{printf "%s ", $(NF-1)}'; echo; done
