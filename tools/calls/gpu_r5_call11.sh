#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c11; rm -rf $O; mkdir -p $O
for rows in 1000000 10000000 100000000; do
timeout 600 python tools/dbg/run_secondary.py c4_one_gpu rows=$rows steps=20 > $O/c4_$rows.json 2> $O/c4_$rows.err
python - <<PY
import json
d=json.load(open("$O/c4_$rows.json")); print("rows $rows", "step", round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["kernels_ms"].items() if k in ("dlrm_fused_fwd","dlrm_fused_bwd","embedding_bwd","linear_bwd_442x128","linear_442x128")}, d.get("roofline_dlrm_fused_bwd"))
PY
done
exit 0
