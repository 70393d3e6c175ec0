#!/bin/bash
# Round 3, first GPU call:  gpurun --timeout 1500 -- 'bash tools/gpu_r3_call1.sh'
# gpurun_in/libs/lib_<v>.so = main + ONE file group of exp/r3-prep (embbwd, inter, gather, chain, gemm, route) or the whole
# branch (all); built by /tmp/build_variants.sh in the authoring container.  MERLIN_HIP_LIB selects one.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c1; mkdir -p $O
L=$PWD/gpurun_in/libs
echo "== gate: new full-size tests on main" | tee $O/gate_main.txt
timeout 700 python -m pytest tests/test_gpu_fullsize_bwd.py tests/test_golden_vectors.py -m gpu -q --durations=8 2>&1 | tail -60 >> $O/gate_main.txt
tail -3 $O/gate_main.txt
echo "== full suite on the whole branch" | tee $O/suite_all.txt
MERLIN_HIP_LIB=$L/lib_all.so timeout 900 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 >> $O/suite_all.txt
tail -3 $O/suite_all.txt
if grep -q "failed" $O/suite_all.txt; then
  for v in embbwd inter gather chain gemm route; do
    echo "== failed tests of the branch on variant $v" | tee -a $O/suite_variants.txt
    MERLIN_HIP_LIB=$L/lib_$v.so timeout 400 python -m pytest tests -m gpu -q --lf -rf 2>&1 | tail -15 >> $O/suite_variants.txt
  done
fi
ab() { # tag, variant, microbench args
  for rep in 1 2; do
    for v in main $2; do
      lib=$L/lib_$v.so; [ $v = main ] && lib=$PWD/models_amd/csrc/libmerlin_hip.so
      echo "-- $1 lib=$v rep=$rep" >> $O/ab_$1.txt
      MERLIN_HIP_LIB=$lib timeout 200 python tools/microbench.py ${@:3} 2>&1 | grep -v Warning >> $O/ab_$1.txt
    done
  done
}
ab embbwd embbwd embbwd embbig
ab fused inter fused
ab chain chain chain
ab gemm gemm linbwd cross
for v in main all; do
  lib=$L/lib_$v.so; [ $v = main ] && lib=$PWD/models_amd/csrc/libmerlin_hip.so
  MERLIN_HIP_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_$v.json
  python -c "
import sys, json; d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('bench $v', round(d['ms_per_step'], 4), round(d['roofline']['frac'], 3), d.get('kernels_ms'), d['config'].get('launch_probe'))" | tee -a $O/bench_summary.txt
done
MERLIN_HIP_LIB=$L/lib_gemm.so timeout 300 python bench.py --workload dcn --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_dcn_gemm.json
timeout 300 python bench.py --workload dcn --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $O/bench_dcn_main.json
echo "== labs"
timeout 120 gpurun_in/lab/scorer_lab 2>&1 | tee $O/scorer_lab.txt | tail -8
for sh in "top " "dX  "; do timeout 120 gpurun_in/lab/gemm_lab "$sh" 2>&1 | grep -v "NO "; done | tee $O/gemm_lab.txt | tail -30
