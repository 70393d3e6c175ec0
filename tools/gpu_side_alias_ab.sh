for i in 1 2; do
for a in "dw=sort,sparse=sort" "sparse=sort" "none" "dw=sort"; do
  echo "== MERLIN_HIP_SIDE_ALIAS=$a"
  MERLIN_HIP_SIDE_ALIAS=$a timeout 200 python bench.py --no-cpu-baseline --no-secondary --sustain 2 --steps 50 --warmup 10 | tail -1 | python -c "import json,sys; h=json.loads(sys.stdin.read()); print(h['ms_per_step'], h['sustained']['ms_per_step'], h['config']['launch'])"
done; done
