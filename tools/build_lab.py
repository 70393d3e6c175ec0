#!/usr/bin/env python
"""Build the LAB variant of the library (-DMH_LAB: tuning / ablation knobs compiled in) into models_amd/csrc/lab/libmerlin_hip_lab.so.
Never shipped, never loaded by default: MERLIN_HIP_LIB=models_amd/csrc/lab/libmerlin_hip_lab.so selects it for one process."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from models_amd import build as B

out = B.CSRC / "lab"
out.mkdir(exist_ok=True)


def one(src):
    obj = out / (src.stem + ".o")
    deps = [src] + sorted(B.CSRC.glob("*.h"))
    if obj.exists() and obj.stat().st_mtime > max(p.stat().st_mtime for p in deps):
        return obj
    import os
    extra = os.environ.get("MH_LAB_FLAGS", "").split()  # e.g. MH_LAB_FLAGS=-fno-slp-vectorize (delete csrc/lab/*.o first)
    r = subprocess.run([B._hipcc(), *B.FLAGS, *extra, "-DMH_LAB", "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(f"{src.name}:\n{r.stderr}")
    return obj


with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(one, B.sources()))
lib = out / "libmerlin_hip_lab.so"
r = subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *map(str, objs), "-o", str(lib)], capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
print(lib)
