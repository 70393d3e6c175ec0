"""Compile one HIP source with -Rpass-analysis=kernel-resource-usage and print a compact per-kernel table.
usage: python tools/resusage.py models_amd/csrc/mh_mlp_chain.hip"""
import re, subprocess, sys
src = sys.argv[1]
r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"n": m.group(1)}
        rows.append(cur)
        continue
    for k, pat in [("v", r"    VGPRs: (\d+)"), ("a", r"AGPRs: (\d+)"), ("s", r"ScratchSize \[bytes/lane\]: (\d+)"),
                   ("o", r"Occupancy \[waves/SIMD\]: (\d+)"), ("vs", r"VGPRs Spill: (\d+)"), ("l", r"LDS Size \[bytes/block\]: (\d+)")]:
        m = re.search(pat, line)
        if m and cur is not None:
            cur[k] = m.group(1)
    if "error" in line:
        print(line)
dem = subprocess.run(["c++filt"] + [r_["n"] for r_ in rows], capture_output=True, text=True).stdout.splitlines() if rows else []
for r_, d in zip(rows, dem):
    d = d.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{d[:70]:70s} vgpr {r_.get('v'):>4s} agpr {r_.get('a'):>4s} scratch {r_.get('s'):>4s} occ {r_.get('o')} spill {r_.get('vs')} lds {r_.get('l')}")
