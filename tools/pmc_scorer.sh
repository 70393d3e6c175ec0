#!/bin/bash
# PMC passes for the scorer kernel (development tool)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/pmc1 -o p -- python tools/microbench.py $1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc2 -o p -- python tools/microbench.py $1 > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
for d in ("gpurun_out/pmc1", "gpurun_out/pmc2"):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        print("no csv in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, v in agg.items():
        if "scorer" in k or "linear_fwd" in k or "gemm" in k or "interaction" in k or "segment" in k:
            print(k, {a: f"{b:.3g}" for a, b in v.items()})
PY
