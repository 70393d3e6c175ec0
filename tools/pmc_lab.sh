#!/bin/bash
# PMC passes over a command (default: the GEMM lab on one shape); prints per-kernel sums of every counter.
#   PMC_CMD="tools/exp/gemm_lab top" bash tools/pmc_lab.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD=${PMC_CMD:-tools/exp/gemm_lab top}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_lab$i -o p -- $CMD > gpurun_out/pmc_lab$i.log 2>&1
done
python - <<'PY'
import csv, collections, glob
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for d in sorted(glob.glob("gpurun_out/pmc_lab[0-9]")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, d)].add(r.get("Dispatch_Id"))
for k, v in tot.items():
    cnt = max(len(s) for (kk, d), s in n.items() if kk == k)
    print(f"== {k}  (dispatches {cnt})")
    print("   " + "  ".join(f"{c}={x / cnt:.4g}" for c, x in sorted(v.items())))
PY
