cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
CHAIN_M=65536,262144 timeout 300 python tools/exp/chain_scaling.py 2>&1 | grep -v amdgpu.ids | tee $O/scaling.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 5 > $O/bench_prof.json 2>/dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import csv,glob,json
f=glob.glob("gpurun_out/r2b/train/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f}us {r['Percentage']:>6s}")
d=json.loads([l for l in open("gpurun_out/r2b/bench_prof.json") if l.startswith("{")][-1]); print(d["ms_per_step"], d["step_ms"])
PY
