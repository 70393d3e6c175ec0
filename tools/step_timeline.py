#!/usr/bin/env python
"""Timeline of ONE steady-state train step from a rocprofv3 --kernel-trace CSV: every kernel with its queue (stream), start
offset, duration, and the gap to the previous kernel on the same queue; then the critical-path accounting: time covered by the
launch stream's kernels, its idle gaps, and what ran on the other queues meanwhile.

    step_timeline.py <kernel_trace.csv> [anchor kernel substring = concat_columns] [step index from the end = 3]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "concat_columns"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(starts) < back + 1:
    sys.exit(f"anchor {anchor!r} found {len(starts)} times")
a, b = starts[-back - 1], starts[-back]
t0 = int(rows[a]["Start_Timestamp"])
step = rows[a:b]
print(f"step = {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {len(step)} kernels")
last_end = {}
main_q = step[0]["Queue_Id"]
busy_main = 0.0
print(f"{'queue':>5s} {'start':>8s} {'dur':>8s} {'gap':>7s}  kernel")
for r in step:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    if q == main_q:
        busy_main += (e - s) / 1e3
    print(f"{q:>5s} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {'' if q == main_q else '    '}{short(r['Kernel_Name'])}")
print(f"launch-stream kernels: {busy_main:.1f} us busy")
