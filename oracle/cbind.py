"""ctypes binding of oracle/liboracle_c.so (TEST INFRASTRUCTURE ONLY; see oracle_c.c)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SO = HERE / "liboracle_c.so"
_lib = None


def build() -> Path:
    src = HERE / "oracle_c.c"
    if not SO.exists() or SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), "liboracle_c.so"], check=True, capture_output=True)
    return SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(SO))
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def gemm_nt_fmaf(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    A, B = _f(A), _f(B)
    M, K = A.shape
    N = B.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    lib().oc_gemm_nt_fmaf(_fp(A), _fp(B), C.c_int64(M), C.c_int64(N), C.c_int64(K), _fp(out))
    return out


def gemm_nn_fmaf(X: np.ndarray, W: np.ndarray) -> np.ndarray:
    X, W = _f(X), _f(W)
    M, K = X.shape
    N = W.shape[1]
    out = np.empty((M, N), dtype=np.float32)
    lib().oc_gemm_nn_fmaf(_fp(X), _fp(W), C.c_int64(M), C.c_int64(N), C.c_int64(K), _fp(out))
    return out


def topk_rows(S: np.ndarray, k: int):
    S = _f(S)
    M, N = S.shape
    vals = np.empty((M, k), dtype=np.float32)
    idx = np.empty((M, k), dtype=np.int32)
    lib().oc_topk_rows(_fp(S), C.c_int64(M), C.c_int64(N), C.c_int32(k), _fp(vals), _fp(idx))
    return vals, idx


def bruteforce_topk(Q: np.ndarray, Cand: np.ndarray, ids, k: int):
    Q, Cand = _f(Q), _f(Cand)
    Bq, E = Q.shape
    N = Cand.shape[0]
    vals = np.empty((Bq, k), dtype=np.float32)
    out_ids = np.empty((Bq, k), dtype=np.int32)
    out_idx = np.empty((Bq, k), dtype=np.int32)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        idp = _fp(ids)
    lib().oc_bruteforce_topk(_fp(Q), _fp(Cand), idp, C.c_int64(Bq), C.c_int64(N), C.c_int64(E),
                             C.c_int32(k), _fp(vals), _fp(out_ids), _fp(out_idx))
    return vals, out_ids, out_idx
