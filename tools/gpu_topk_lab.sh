export MERLIN_HIP_LIB=models_amd/csrc/lab/libmerlin_hip_lab.so
for cfg in "XCD_MAP=0" "XCD_MAP=1" "XCD_MAP=1 GROWTH=2" "XCD_MAP=0 GROWTH=2" "XCD_MAP=1 GROWTH=3"; do
  env_args=""; for kv in $cfg; do env_args="$env_args MERLIN_HIP_TOPK_$kv"; done
  echo "== $cfg"
  env $env_args timeout 200 python tools/dbg/run_secondary.py topk | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o['ms_per_step'], o.get('bit_identical_to_f32_pipeline'))"
done
