#!/bin/bash
# SQ busy / MFMA busy / wait counters per kernel for the DLRM train step and the scorer (two PMC passes,
# kernel-trace only).  Prints per-kernel ratios; raw CSVs under gpurun_out/pmc_busy{1,2}.
# PMC_CMD overrides the profiled command, PMC_FILTER (comma-separated substrings) the kernels that are printed.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_busy1 -o p -- ${PMC_CMD:-python tools/microbench.py inter linbwd embbwd scorer} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_busy2 -o p -- ${PMC_CMD:-python tools/microbench.py inter linbwd embbwd scorer} > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob, os
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for d in ("gpurun_out/pmc_busy1", "gpurun_out/pmc_busy2"):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        print("no csv in", d); continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-56:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if d.endswith("1") and key not in seen:
            seen.add(key); n[k] += 1
def ratio(v, a, b):
    return f"{v[a] / v[b]:.2f}" if v.get(b) else "-"
print(f"{'kernel':58s} {'n':>4s} mfma/busy wait_any/wave wait_inst/wave lds_wait/wave lds_conf/lds_act valu/busy vmem/busy")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if not any(s in k for s in os.environ.get("PMC_FILTER", "gemm,linear_fwd,stream_kernel,chain_,interaction,fused,piece,gather_fwd").split(",")):
        continue
    print(f"{k:58s} {n[k]:4d} {ratio(v,'SQ_VALU_MFMA_BUSY_CYCLES','SQ_BUSY_CYCLES'):>9s} {ratio(v,'SQ_WAIT_ANY','SQ_WAVE_CYCLES'):>13s} "
          f"{ratio(v,'SQ_WAIT_INST_ANY','SQ_WAVE_CYCLES'):>14s} {ratio(v,'SQ_WAIT_INST_LDS','SQ_WAVE_CYCLES'):>13s} "
          f"{ratio(v,'SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE'):>16s} {ratio(v,'SQ_ACTIVE_INST_VALU','SQ_BUSY_CYCLES'):>9s} "
          f"{ratio(v,'SQ_ACTIVE_INST_VMEM','SQ_BUSY_CYCLES'):>9s}  insts/wave-launch: valu {v.get('SQ_INSTS_VALU',0)/max(n[k],1):.3g} "
          f"vmem_rd {v.get('SQ_INSTS_VMEM_RD',0)/max(n[k],1):.3g} vmem_wr {v.get('SQ_INSTS_VMEM_WR',0)/max(n[k],1):.3g} "
          f"busy_cycles {v.get('SQ_BUSY_CYCLES',0)/max(n[k],1):.3g} wave_cycles {v.get('SQ_WAVE_CYCLES',0)/max(n[k],1):.3g} gui {v.get('GRBM_GUI_ACTIVE',0)/max(n[k],1):.3g}")
PY
