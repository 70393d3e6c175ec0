#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c5; mkdir -p $O
MH_FORCE_DISTRIBUTED=1 timeout 300 python bench.py --steps 50 --warmup 8 --no-cpu-baseline --sustain 1 2>$O/forced_err.txt | tail -1 > $O/forced.json
python -c "
import json; d = json.loads(open('$O/forced.json').read().strip().splitlines()[-1]); print('forced', round(d['ms_per_step'], 4), d['config']['launch'], json.dumps(d['sharded'])[:1500])"
tail -3 $O/forced_err.txt
( time timeout 600 python bench.py 2>$O/default_err.txt | tail -1 > $O/default.json ) 2>&1 | grep real
python -c "
import json; d = json.loads(open('$O/default.json').read().strip().splitlines()[-1]); print('default', round(d['ms_per_step'], 4), d['config']['launch'], d['config']['launch_probe'], d['roofline']['frac'], d['sustained']); s=d['secondary']; print(json.dumps(s.get('hbm_copy_peak'))); print(json.dumps(s.get('c4_one_gpu'))[:1200]); print(s.get('error')); print(d['cpu_baseline']['value'], d.get('max_abs_err_vs_oracle'))"
tail -3 $O/default_err.txt
