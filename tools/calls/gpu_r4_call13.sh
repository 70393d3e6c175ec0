#!/bin/bash
# look-back second sort pass: correctness (every test that runs the sparse update) then A/B against the classic passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c13; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "embedding or backward or fullsize or models or bag or world2 or route or hygiene" > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
for m in classic lookback classic lookback; do
  echo "== $m" | tee -a $O/ab.txt
  if [ $m = classic ]; then export MERLIN_HIP_SORT=classic; else unset MERLIN_HIP_SORT; fi
  python tools/microbench.py embada emb1m 2>/dev/null | grep "embedding bwd" | tee -a $O/ab.txt
done
