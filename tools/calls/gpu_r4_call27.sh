#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c27; rm -rf $O; mkdir -p $O
f() { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -45; }
MERLIN_HIP_DETERMINISTIC=1 timeout 300 python -m pytest "tests/test_gpu_fullsize.py::test_c2_embedding_backward_full_batch_properties" -m gpu -q -x 2>&1 | f > $O/deterministic.txt
MERLIN_HIP_SIDE_STREAMS=0 timeout 300 python -m pytest "tests/test_gpu_models.py::test_graph_replayed_train_steps_equal_eager_steps" -m gpu -q 2>&1 | f > $O/no_side.txt
MERLIN_HIP_MLP_CHAIN=0 MERLIN_HIP_FUSED_DLRM=0 timeout 300 python -m pytest "tests/test_gpu_models.py::test_graph_replayed_train_steps_equal_eager_steps" -m gpu -q 2>&1 | f > $O/no_chain.txt
MERLIN_HIP_SCORER_FWD=tiled timeout 300 python -m pytest "tests/test_gpu_retrieval.py::test_inbatch_scorer_matches_oracle" -m gpu -q -x 2>&1 | f > $O/tiled.txt
