#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c9; rm -rf $O; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/scorer_lab.hip -Imodels_amd/csrc -o /tmp/scorer_lab && /tmp/scorer_lab | tee $O/lab.txt
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
