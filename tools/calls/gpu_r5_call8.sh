#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c8; rm -rf $O; mkdir -p $O
for v in 0 1; do
MERLIN_HIP_DW_LATE=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl$v -o t -- python bench.py --eager --no-secondary --no-cpu-baseline --steps 30 --warmup 10 --sustain 0 > $O/bench$v.json 2>$O/err$v.txt
f=$(find $O/tl$v -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f concat_columns 20 > $O/timeline_dwlate$v.txt 2>&1
rm -rf $O/tl$v
cat $O/timeline_dwlate$v.txt
done
timeout 600 python tools/dbg/run_secondary.py embedding_bag > $O/bag.json 2> $O/bag.err; python - <<PY
import json
d=json.load(open("$O/bag.json")); print({k:(round(v["ms"],3),round(v["frac"],3)) for k,v in d["fwd"].items()}, "cold", round(d["cold"]["ms"],4), round(d["cold"]["frac"],3), "dense", round(d["dense_list_fwd_mean"]["ms"],3))
PY
timeout 600 python tools/dbg/run_secondary.py fit_from_parquet > $O/fit.json 2> $O/fit.err; tail -c 1500 $O/fit.json; tail -3 $O/fit.err
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_loader.py tests/test_gpu_bag_backward.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | cut -c1-300
exit 0
