#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c30; rm -rf $O; mkdir -p $O
run() { n=$1; shift
  env "$@" timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_world2.py tests/test_gpu_retrieval.py tests/test_gpu_fullsize.py tests/test_gpu_route.py -m gpu -q --tb=short 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -60 > $O/$n.txt
  echo "== $n: $(tail -1 $O/$n.txt)"
}
run default X=1
run deterministic MERLIN_HIP_DETERMINISTIC=1
run no_side MERLIN_HIP_SIDE_STREAMS=0
run no_chain MERLIN_HIP_MLP_CHAIN=0 MERLIN_HIP_FUSED_DLRM=0
run lookback_tiled MERLIN_HIP_SORT=lookback MERLIN_HIP_SCORER_FWD=tiled MERLIN_HIP_TOPK_FILTER=stream
