#!/bin/bash
# same-box A/B of the top-k call: the shipped library against libraries with another mh_topk.o (MERLIN_HIP_LIB); arguments = library paths
libs="${@:-models_amd/csrc/lab/libmerlin_hip_oldtopk.so models_amd/csrc/libmerlin_hip.so}"
for i in 1 2; do
for lib in $libs; do
  echo "== $lib"
  MERLIN_HIP_LIB=$lib timeout 200 python tools/dbg/run_secondary.py topk | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o['ms_per_step'], o.get('bit_identical_to_f32_pipeline'))"
done; done
