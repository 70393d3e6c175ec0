#!/bin/bash
# knobs of the stream kernel's forward: tile rows and the s_setprio split
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for bn in 64 128; do for prio in 0 1; do
MERLIN_HIP_SCORER_BN_FWD=$bn MERLIN_HIP_SCORER_PRIO=$prio python - <<PY 2>/dev/null
import torch, bench
r = bench.run_scorer_fwd(torch.device("cuda:0"))
print("bn=$bn prio=$prio", round(r["ms"],4), "ms", round(r["tflops"],1), "TF", round(r["frac_of_peak"],3))
PY
done; done
