#!/bin/bash
# SQ / TCC counters per kernel of any workload (diagnosis, not a judged figure):
#   gpurun -- 'WORK="python tools/microbench.py towers" PASSES="sq1 sq2" bash tools/dbg/pmc_kernels.sh'
# prints, per kernel name (first 60 characters) and pass, the mean counter values over its dispatches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_kernels; rm -rf $O; mkdir -p $O
: ${WORK:="python tools/pmc_workload.py fused"}; : ${PASSES:="sq1 sq2 tcc tcp"}
pass() {  # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- $WORK > $O/$n.log 2>&1
  python - <<PY | tee -a $O/summary.txt
import csv, glob, collections, re
f = glob.glob("$O/$n/**/*counter_collection.csv", recursive=True)
if not f:
    print("$n: no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1].get(next(iter(kv[1])), [0])))):
    n_ = len(next(iter(d.values())))
    if any(t in k for t in ("at::native", "rocprim", "amd_rocclr")):
        continue
    print("$n", k, "x%d" % n_, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
  rm -rf $O/$n
}
want() { case " $PASSES " in *" $1 "*) return 0;; esac; return 1; }
want sq1 && pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
want sq2 && pass sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
want tcc && pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
want tcp && pass tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
true
