#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c16; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "replayed_train_steps" > $O/pytest_rec.log 2>&1; tail -15 $O/pytest_rec.log
for r in 1 2; do
python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 30 --sustain 2 2>$O/err_$r.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4), d['config']['launch'], d['config']['launch_probe'])
" | tee -a $O/ab.txt
done
tail -3 $O/err_1.txt
