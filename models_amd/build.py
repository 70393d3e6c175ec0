"""Build libmerlin_hip.so (gfx950) with hipcc, in-tree.

The shared object lands next to the sources (``models_amd/csrc/libmerlin_hip.so``): it is
git-ignored but travels to the GPU box with the repository snapshot.  hipcc cross-compiles
without a GPU, so this runs in the authoring container as the "does it build" check.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libmerlin_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: cannot build libmerlin_hip.so")
    return exe


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _deps() -> list[Path]:
    return sources() + sorted(CSRC.glob("*.h")) + [CSRC.parent.parent / "include" / "merlin_hip.h"]


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, the C-ABI header): stamps measurement files (profiles/pmc_traffic.json)
    so that a figure taken with other kernels is recognised as stale."""
    import hashlib

    h = hashlib.sha256()
    for p in _deps():
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in _deps())


def _compile(src: Path, verbose: bool) -> Path:
    obj = src.with_suffix(".o")
    hdr_t = max(p.stat().st_mtime for p in _deps() if p.suffix == ".h")
    if obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_t):
        return obj
    # MH_LAB=1 (with force=True): the tuning / ablation knobs of the labs are compiled in (mh_common.h: MH_LAB_ENV); never shipped
    lab = ["-DMH_LAB"] if os.environ.get("MH_LAB") == "1" else []
    cmd = [_hipcc(), *FLAGS, *lab, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every ``csrc/*.hip`` for gfx950 and link ``libmerlin_hip.so``."""
    if not force and not needs_build():
        return LIB
    # one builder at a time: concurrent ranks (torchrun) finding a stale library would otherwise write the same
    # .o / .so files at once; the losers of the lock find the library fresh and return
    import fcntl

    with open(CSRC / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> Path:
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    tmp = LIB.with_suffix(".so.tmp")
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB)  # atomic: a process loading the library never sees a half-written file
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
