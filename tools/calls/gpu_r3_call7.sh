#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_bwd.py tests/test_gpu_bag_backward.py tests/test_gpu_edges.py tests/test_gpu_hygiene.py tests/test_gpu_models.py -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest.txt
timeout 200 python tools/microbench.py embbwd embbig 2>&1 | grep " us" | tee $O/microbench.txt
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python tools/microbench.py embada > /dev/null 2>&1
python tools/trace_summary.py $(ls $O/tr/*kernel_trace.csv | head -1) 2 | head -14 | tee $O/trace_embada.txt
rm -rf $O/tr
timeout 300 python bench.py --no-cpu-baseline --no-secondary --sustain 1 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'], 4), d['config']['launch'], d['config']['launch_probe'], 'roofline', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), d['roofline'].get('frac_dedup_aware'))"
