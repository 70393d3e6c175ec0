#!/bin/bash
# the whole GPU suite under the non-default switches (each a code path the default run never takes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c26; rm -rf $O; mkdir -p $O
run() { n=$1; shift
  env "$@" timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 > $O/$n.txt
  echo "== $n: $(tail -1 $O/$n.txt)"
}
run deterministic MERLIN_HIP_DETERMINISTIC=1
run no_side MERLIN_HIP_SIDE_STREAMS=0
run no_chain MERLIN_HIP_MLP_CHAIN=0 MERLIN_HIP_FUSED_DLRM=0
run tail_on MERLIN_HIP_TAIL=1 MERLIN_HIP_SIDE_ALIAS=none
run lookback_tiled MERLIN_HIP_SORT=lookback MERLIN_HIP_SCORER_FWD=tiled MERLIN_HIP_TOPK_FILTER=stream
