#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c8; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "tiled_forward or scorer or softmax or contrastive or retrieval" > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
python - <<'PY' | tee $O/scorer_ab.txt
import os, torch, sys
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
for mode in ("stream", "tiled", "stream", "tiled"):
    os.environ["MERLIN_HIP_SCORER_FWD"] = mode
    r = bench.run_scorer_fwd(dev)
    print(mode, round(r["ms"], 4), round(r["frac_of_peak"], 4))
PY
