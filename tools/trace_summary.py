#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV, keeping launches of one kernel apart by their position inside the
repeating launch sequence (the two sort passes run the same kernels).  Usage: trace_summary.py <kernel_trace.csv> [skip_first_n]"""
import csv
import sys
from collections import OrderedDict, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# occurrence index of a name since the last "first kernel of the sequence" (the most frequent first name after torch's own fills)
# a launch sequence that repeats (timing loops): kernels that run P times per repetition get the suffix #0 .. #P-1
per_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
occ = defaultdict(int)
agg = OrderedDict()
for n, d in zip(names, dur):
    k = f"{n} #{occ[n] % per_rep}" if per_rep > 1 and n.startswith("radix") else n
    occ[n] += 1
    agg.setdefault(k, []).append(d)
print(f"{'kernel':60s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'total ms':>9s}")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:60]:60s} {len(v):6d} {sum(v) / len(v):9.1f} {min(v):9.1f} {sum(v) / 1e3:9.2f}")
