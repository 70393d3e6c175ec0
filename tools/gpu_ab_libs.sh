#!/bin/bash
# A/B of prebuilt library variants on one box: gpurun -- 'bash tools/gpu_ab_libs.sh v1 v2 ...'  (v = main: the in-tree library,
# else gpurun_in/libs/lib_<v>.so); two alternating rounds of bench.py (DLRM C2 step) per variant
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in ${REPS:-1 2}; do for v in "$@"; do
  lib=$PWD/gpurun_in/libs/lib_$v.so; [ $v = main ] && lib=$PWD/models_amd/csrc/libmerlin_hip.so
  MERLIN_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-secondary --sustain 1 2>/dev/null | tail -1 > gpurun_out/ab_$v.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
k = d["kernels_ms"]; lp = d["config"]["launch_probe"]
print("$v", "step", round(d["ms_per_step"], 4), "graph", round(lp["hipGraph_replay_ms"], 4), "seg", round(lp.get("segmented_replay_ms", 0), 4),
      "| embbwd", k["embedding_bwd"], "ffwd", k["dlrm_fused_fwd"], "fbwd", k["dlrm_fused_bwd"], "lin", k.get("linear_415x128"), "linbwd", k.get("linear_bwd_415x128"),
      "chains", k.get("mlp_chain_13x128x64"), k.get("mlp_chain_128x64x32x1"), k.get("mlp_chain_bwd_128x64x32x1"), k.get("mlp_chain_bwd_13x128x64"))
PY
done; done
