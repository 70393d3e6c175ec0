#!/bin/bash
# SQ / TCC counters of the fused gather -> interaction kernels (diagnosis, not a judged figure):  gpurun -- 'bash tools/dbg/pmc_fused.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_fused; rm -rf $O; mkdir -p $O
pass() {  # name counters...
  n=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python tools/pmc_workload.py fused > $O/$n.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$O/$n/**/*counter_collection.csv", recursive=True)
if not f:
    print("$n: no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "dlrm_fused" in k:
        acc["fwd" if "fwd" in k else "bwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("$n", k, {c: round(sum(v[-5:]) / len(v[-5:])) for c, v in d.items()})
PY
  rm -rf $O/$n
}
if [ -n "$1" ]; then PASSES="$*"; else PASSES="sq1 sq2 tcc tcp"; fi
want() { case " $PASSES " in *" $1 "*) return 0;; esac; return 1; }
want sq1 && pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
want sq2 && pass sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
want tcc && pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
want tcp && pass tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
