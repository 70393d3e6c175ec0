#!/bin/bash
# A/B of ONE environment switch in the same library on one box:  gpurun -- 'VAR=MERLIN_HIP_ASTAT VALUES="off 4" bash tools/dbg/ab_env.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in ${REPS:-1 2}; do for v in $VALUES; do
  if [ $v = off ]; then unset $VAR; else export $VAR=$v; fi
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --sustain 1 2>/dev/null | tail -1 > gpurun_out/ab_env_$v.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_env_$v.json").read().strip().splitlines()[-1])
k = d["kernels_ms"]; lp = d["config"]["launch_probe"]
print("$VAR=$v", "step", round(d["ms_per_step"], 4), "graph", round(lp["hipGraph_replay_ms"], 4), "seg", round(lp.get("segmented_replay_ms", 0), 4), "eager", round(lp["eager_side_streams_ms"], 4),
      "| lin", k.get("linear_415x128"), "linbwd", k.get("linear_bwd_415x128"))
PY
done; done
