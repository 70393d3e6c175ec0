#!/usr/bin/env python
"""Static audit of the gfx950 ISA of the kernels under models_amd/csrc (no GPU needed: hipcc cross-compiles).

For every kernel it reports the patterns that turned out to be exposed memory latency in round 2 (profiles/r2_notes.md, "Leads"):

  waited   loads (global / flat / buffer) whose NEXT instruction is `s_waitcnt vmcnt(0)`: a load that is used at once.  In an
           unrolled "all loads in flight" loop this means the loads are sequential round trips (a load behind a per-element
           predicate with its use right behind it; a load that must stay behind a store it may alias; `a || b[i]`).
  flat     flat_load / flat_store / flat_atomic: a pointer the compiler could not prove global (rebuilt from integers, read
           from LDS).  Flat accesses count in lgkmcnt as well, so every LDS wait also waits for them.
  kernarg  global loads addressed as (kernel-argument pointer s[0:1] + VGPR offset): a kernel-argument ARRAY indexed with a
           per-lane value.  The element comes from memory, and whatever address is computed from it hangs behind that load.

usage: python tools/isa_audit.py [file.hip ...]   (default: every models_amd/csrc/*.hip)      -> table on stdout
"""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LOAD = re.compile(r"^(global|flat|buffer)_load")
KERNARG = re.compile(r"^global_load_dword\w*\s+v\[?\d+(:\d+)?\]?, v\d+, s\[0:1\]")
FLAT = re.compile(r"^flat_(load|store|atomic)")


def audit(src: Path):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", str(src), "-o", f"{td}/x.o",
                            "--save-temps=obj"], capture_output=True, text=True, cwd=td)
        asm = list(Path(td).glob("*-hip-amdgcn-amd-amdhsa-gfx950.s"))
        if r.returncode != 0 or not asm:
            print(f"{src.name}: compile failed\n{r.stderr[-2000:]}")
            return []
        rows, cur, prev_load = collections.OrderedDict(), None, False
        for line in asm[0].read_text().splitlines():
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
                rows[cur] = collections.Counter()
                prev_load = False
                continue
            t = line.strip()
            if cur is None or not t or t[0] in ";.":
                continue
            c = rows[cur]
            if LOAD.match(t):
                c["loads"] += 1
                prev_load = True
                if KERNARG.match(t):
                    c["kernarg"] += 1
            else:
                if prev_load and t.startswith("s_waitcnt vmcnt(0)"):
                    c["waited"] += 1
                if not t.startswith("s_nop"):
                    prev_load = False
            if FLAT.match(t):
                c["flat"] += 1
            if t.startswith("v_mfma"):
                c["mfma"] += 1
        names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
        return [(src.name, n.replace("(anonymous namespace)::", "").replace("void ", ""), c) for n, c in zip(names, rows.values())]


def main():
    files = [Path(a) for a in sys.argv[1:]] or sorted((ROOT / "models_amd" / "csrc").glob("*.hip"))
    out = []
    for f in files:
        out += audit(f.resolve())
    out = [(f, n, c) for f, n, c in out if c["waited"] >= 2 or c["flat"] or c["kernarg"]]
    out.sort(key=lambda r: -(r[2]["waited"] + 4 * r[2]["flat"] + 4 * r[2]["kernarg"]))
    print(f"{'file':18s} {'loads':>5s} {'waited':>6s} {'flat':>4s} {'kernarg':>7s} {'mfma':>5s}  kernel")
    for f, n, c in out:
        print(f"{f:18s} {c['loads']:5d} {c['waited']:6d} {c['flat']:4d} {c['kernarg']:7d} {c['mfma']:5d}  {n[:110]}")


if __name__ == "__main__":
    main()
