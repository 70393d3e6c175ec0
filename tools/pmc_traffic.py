#!/usr/bin/env python
"""HBM traffic of the HBM-bound launches from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE in SEPARATE rocprofv3 passes (kernel-trace only), the gfx950 reading of FETCH_SIZE (half the bytes of a wide
coalesced read) checked against a known byte count of our own (the `calib` segment) instead of assumed.
Run on the GPU box from the repo root:  python tools/pmc_traffic.py   ->  gpurun_out/pmc_traffic.json (+ raw CSVs);
copy the json to profiles/pmc_traffic.json (bench.py reads it for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out" / "pmc_traffic"
N = 5
SEGMENTS = {"calib": None, "embbwd": ("radix_", "piece_", "carry_apply", "fill_words"), "gather": ("gather_fwd",), "cold": ("gather_fwd",),
            "fused_fwd": ("dlrm_fused_fwd",), "fused_bwd": ("dlrm_fused_bwd",)}
WORKLOAD = {"fused_fwd": "fused", "fused_bwd": "fused"}  # segment -> pmc_workload.py argument (default: the segment name)


def run_pass(seg, counter):
    d = OUT / f"{seg}_{counter}"
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", str(d), "-o", "p", "--",
                    sys.executable, str(ROOT / "tools" / "pmc_workload.py"), WORKLOAD.get(seg, seg)], cwd="/tmp", env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    f = glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True)
    rows = []
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(rows)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    sys.path.insert(0, str(ROOT))
    from models_amd.build import source_hash

    res = {"source_hash": source_hash(), "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, one counter per pass, one process per segment "
                     "(tools/pmc_workload.py); counter units calibrated on a 1 GiB fill / 1 GiB sum", "launches_per_segment": N}
    raw = {}
    for seg in SEGMENTS:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            raw[(seg, counter)] = run_pass(seg, counter)
    gib = float(1 << 30)
    fills = [v for _, k, v in raw[("calib", "WRITE_SIZE")] if "FillFunctor" in k][-N:]
    sums = [v for _, k, v in raw[("calib", "FETCH_SIZE")] if "reduce_kernel" in k]
    # x.sum() of 2^28 floats = one large reduce_kernel + a tiny second stage: keep the large ones
    sums = sorted(sums)[-N:]
    w_unit = gib / (sum(fills) / len(fills)) if fills else None   # bytes per WRITE_SIZE unit
    f_unit = gib / (sum(sums) / len(sums)) if sums else None      # bytes per FETCH_SIZE unit
    res["calibration"] = {"write_bytes_per_unit": w_unit, "fetch_bytes_per_unit": f_unit,
                          "note": "rocprofv3 documents both counters in KiB; the guide's gfx950 correction (FETCH_SIZE counts "
                                  "half of a wide coalesced read) shows up as fetch_bytes_per_unit ~ 2048"}
    for seg, pats in SEGMENTS.items():
        if pats is None:
            continue
        entry = {}
        for counter, unit in (("FETCH_SIZE", f_unit), ("WRITE_SIZE", w_unit)):
            per = collections.defaultdict(list)
            for _, k, v in raw[(seg, counter)]:
                short = k.replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "")
                if any(p in short for p in pats):
                    per[short].append(v)
            entry[counter] = {k: {"dispatches": len(v), "units_per_launch": sum(v) / N} for k, v in per.items()}
            tot_units = sum(e["units_per_launch"] for e in entry[counter].values())
            entry[counter + "_bytes_per_launch"] = tot_units * unit if unit else None
        if entry.get("FETCH_SIZE_bytes_per_launch") is not None and entry.get("WRITE_SIZE_bytes_per_launch") is not None:
            entry["traffic_bytes"] = entry["FETCH_SIZE_bytes_per_launch"] + entry["WRITE_SIZE_bytes_per_launch"]
        name = {"embbwd": "embedding_bwd", "gather": "embedding_gather", "cold": "gather_cold", "fused_fwd": "dlrm_fused_fwd",
                "fused_bwd": "dlrm_fused_bwd"}[seg]
        res[name] = entry
    json.dump(res, open(ROOT / "gpurun_out" / "pmc_traffic.json", "w"), indent=1)
    for k in ("calibration", "embedding_bwd", "embedding_gather", "gather_cold", "dlrm_fused_fwd", "dlrm_fused_bwd"):
        v = res.get(k, {})
        print(k, {a: b for a, b in v.items() if not isinstance(b, dict) or a == "calibration"} if k != "calibration" else v)


if __name__ == "__main__":
    main()
