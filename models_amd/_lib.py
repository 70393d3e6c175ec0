"""ctypes binding of libmerlin_hip.so -- the C ABI declared in include/merlin_hip.h.

There is NO CPU fallback: if the shared object is missing and cannot be built, importing the
ops raises.  PyTorch tensors are only buffer carriers (``tensor.data_ptr()``), the stream is
``torch.cuda.current_stream().cuda_stream``.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
from pathlib import Path

from . import build as _build

MH_OK = 0
MH_I32, MH_I64 = 0, 1
ACT = {"linear": 0, None: 0, "none": 0, "relu": 1, "sigmoid": 2}
COMBINER = {"sum": 0, "mean": 1, "sqrtn": 2, "max": 3}
OPT = {"sgd": 0, "adagrad": 1, "adam": 2, "lazy_adam": 2}
MAX_FEATURES = 64

_p = C.c_void_p
_i32, _i64, _f32 = C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every symbol of include/merlin_hip.h
SIGNATURES = {
    "mh_version": (_i32, []),
    "mh_last_error": (C.c_char_p, []),
    "mh_device_info": (_i32, [C.POINTER(_i32), C.POINTER(_i64)]),
    "mh_embedding_gather_fwd": (_i32, [_p, _p, _p, _i32, _i64, _i32, _i32, _p, _i64, _p, _p]),
    "mh_embedding_bag_fwd": (_i32, [_p, _i64, _p, _i64, _p, _i32, _i64, _i32, _i32, _p, _i64, _p]),
    "mh_embedding_dense_list_fwd": (_i32, [_p, _i64, _p, _i32, _i64, _i32, _i32, _i32, _p, _i64, _p]),
    "mh_set_deterministic": (_i32, [_i32]),
    "mh_set_scorer_arith": (_i32, [_i32]),
    "mh_set_gemm_arith": (_i32, [_i32]),
    "mh_cross_layer_split_workspace_bytes": (_i64, [_i64, _i32]),
    "mh_cross_layer_fwd_split": (_i32, [_p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i64, _p]),
    "mh_cross_layer_bwd_split": (_i32, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p, _i32, _p, _p, _p, _p, _i64, _p]),
    "mh_embedding_bwd_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "mh_embedding_bag_expand": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _i32, _p, _i64, _p, _p, _p]),
    "mh_embedding_bag_bwd_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "mh_embedding_bag_bwd": (_i32, [_p, _p, _p, _i64, _p, _i64, _p, _i64, _i32, _i64, _i32, _i32, _p, _i64, _i32, _f32, _f32, _f32, _f32, _p, _p, _i64, _p]),
    "mh_embedding_bag_bwd_multi_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32]),
    "mh_embedding_bag_bwd_multi": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i32, _i32, _i32, _p, _i64, _p, _i32, _f32, _f32, _f32, _f32, _p, _p, _i64, _p]),
    "mh_embedding_gather_bwd": (_i32, [_p, _p, _p, _p, _i32, _i64, _i32, _i32, _p, _i64, _p, _i32, _f32, _f32, _p, _f32, _f32, _p, _p, _i64, _p]),
    "mh_embedding_gather_bwd_prepare": (_i32, [_p, _p, _p, _i32, _i64, _i32, _i32, _p, _i64, _p]),
    "mh_embedding_gather_bwd_apply": (_i32, [_p, _p, _p, _p, _i32, _i64, _i32, _i32, _p, _i64, _p, _i32, _f32, _f32, _p, _f32, _f32, _p, _p, _i64, _p]),
    "mh_l2_batch_reg": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _f32, _p, _p, _p]),
    "mh_linear_bias_act_fwd": (_i32, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _i64, _p]),
    "mh_linear_bwd_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "mh_linear_bias_act_bwd": (_i32, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _i64, _p]),
    "mh_tower_supported": (_i32, [_i64, _i32, _i32]),
    "mh_tower_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "mh_tower_linear_fwd": (_i32, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p]),
    "mh_tower_linear_dx": (_i32, [_p, _i64, _p, _i64, _i32, _i32, _p, _i64, _p, _i64, _p]),
    "mh_linear_split_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "mh_linear_bias_act_fwd_split": (_i32, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p]),
    "mh_linear_bias_act_bwd_split": (_i32, [_p, _i64, _p, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _i64, _p]),
    "mh_mlp_chain_supported": (_i32, [_i32, _p]),
    "mh_mlp_chain_fwd": (_i32, [_p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "mh_mlp_chain_bwd_workspace_bytes": (_i64, [_i64, _i32, _p]),
    "mh_mlp_chain_bwd": (_i32, [_p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _i64, _p, _p, _p, _i64, _p]),
    "mh_dot_interaction_fwd": (_i32, [_p, _i64, _i32, _i32, _p, _i64, _i32, _i32, _p, _i64, _p]),
    "mh_dot_interaction_bwd": (_i32, [_p, _p, _i64, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p]),
    "mh_mlp_chain_bwd_partial": (_i32, [_p, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _i64, _p, _i64, _p]),
    "mh_mlp_chain_bwd_reduce": (_i32, [_i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "mh_rowwise_dot": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _p, _p]),
    "mh_dense_optimizer_step_multi": (_i32, [_p, _p, _p, _p, _i32, _i32, _f32, _f32, _p, _f32, _f32, _p, _p]),
    "mh_eltwise": (_i32, [_i32, _p, _p, _p, _p, _i64, _p]),
    "mh_route_workspace_bytes": (_i64, [_i64, _i32]),
    "mh_route_build": (_i32, [_p, _i32, _i32, _i64, _i32, _p, _i32, _i64, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "mh_route_dedup_workspace_bytes": (_i64, [_i64, _i32]),
    "mh_route_build_dedup": (_i32, [_p, _i32, _i32, _i64, _i32, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "mh_route_local_rows": (_i32, [_p, _i64, _p, _p, _i32, _p, _p]),
    "mh_comm_unique_id": (_i32, [_p]),
    "mh_comm_init": (_i32, [_i32, _i32, _p, C.POINTER(_p)]),
    "mh_comm_destroy": (_i32, [_p]),
    "mh_comm_alltoall": (_i32, [_p, _p, _p, _i64, _p]),
    "mh_allreduce_dense": (_i32, [_p, _p, _i64, _p]),
    "mh_sharded_lookup_workspace_bytes": (_i64, [_i64, _i32, _i32, _i64, _i32]),
    "mh_sharded_lookup_fwd": (_i32, [_p, _p, _i32, _i32, _i64, _i64, _p, _p, _p, _i32, _p, _i64, _p, _p, _p, _i64, _p]),
    "mh_sharded_lookup_bwd": (_i32, [_p, _i32, _i64, _i64, _i32, _p, _p, _p, _p, _i64, _i32, _f32, _f32, _f32, _f32, _p, _p, _i64, _p]),
    "mh_dense_optimizer_step": (_i32, [_p, _p, _p, _i64, _i32, _f32, _f32, _p, _f32, _f32, _p, _p]),
    "mh_adam_tick": (_i32, [_p, _f32, _f32, _f32, _p, _p]),
    "mh_dlrm_interaction_fused_fwd": (_i32, [_p, _p, _p, _i32, _p, _i64, _i64, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "mh_dlrm_interaction_fused_bwd": (_i32, [_p, _p, _p, _i32, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "mh_cross_layer_fwd": (_i32, [_p, _p, _p, _p, _i64, _i32, _p, _p]),
    "mh_cross_layer_fwd_save": (_i32, [_p, _p, _p, _p, _i64, _i32, _p, _p, _p]),
    "mh_cross_layer_lowrank_fwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _p]),
    "mh_cross_layer_bwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _p, _i32, _p, _p, _p, _p, _i64, _p]),
    "mh_l2norm_rows": (_i32, [_p, _i64, _i32, _f32, _p, _p]),
    "mh_l2norm_rows_bwd": (_i32, [_p, _p, _i64, _i32, _f32, _p, _p]),
    "mh_inbatch_softmax_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32]),
    "mh_inbatch_softmax_fwd_dq": (_i32, [_p, _p, _p, _p, _p, _i32, _i64, _i64, _i32, _f32, _f32, _p, _p, _i32, _f32, _p, _p, _p, _p, _p, _i64, _p]),
    "mh_inbatch_softmax_fwd": (_i32, [_p, _p, _p, _p, _p, _i32, _i64, _i64, _i32, _f32, _f32, _p, _p, _i32, _p, _i64, _p, _p, _p, _i64, _p]),
    "mh_inbatch_softmax_bwd": (_i32, [_p, _p, _p, _p, _p, _i32, _i64, _i64, _i32, _f32, _f32, _p, _p, _i32, _p, _f32, _p, _p, _p, _p, _i64, _p]),
    "mh_topk_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "mh_topk_dot": (_i32, [_p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _p]),
    "mh_topk_split": (_i32, [_p, _i64, _i32, _p, _p, _p, _p, _p]),
    "mh_topk_split_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32]),
    "mh_topk_dot_split": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _p]),
    "mh_topk_metrics": (_i32, [_p, _i64, _p, _i64, _i32, _p, _p]),
    "mh_dropout": (_i32, [_p, _p, _i64, _f32, _p, _i32, _p]),
    "mh_batchnorm_workspace_bytes": (_i64, [_i64, _i32]),
    "mh_batchnorm_fwd": (_i32, [_p, _i64, _i64, _i32, _p, _p, _f32, _f32, _i32, _p, _p, _p, _p, _p, _i64, _p, _i64, _p]),
    "mh_batchnorm_bwd": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _i32, _p, _i64, _p, _p, _p, _i64, _p]),
    "mh_copy_many": (_i32, [_p, _p, _p, _i32, _p]),
    "mh_concat_columns": (_i32, [_p, _p, _p, _i32, _i64, _p, _i64, _i32, _p]),
    "mh_stream_copy": (_i32, [_p, _p, _i64, _p]),
    "mh_log_uniform_sample_workspace_bytes": (_i64, [_i64, _i32]),
    "mh_log_uniform_sample": (_i32, [_i64, _i64, _i64, _i32, _p, _p, _p, _i64, _p]),
    "mh_bce_fwd_bwd": (_i32, [_p, _p, _i64, _f32, _p, _p, _p]),
    "mh_bce_mean_fwd_bwd": (_i32, [_p, _p, _i64, _f32, _p, _p, _p, _p]),
    "mh_mean": (_i32, [_p, _i64, _p, _p, _p]),
    "mh_record_begin": (_i32, []),
    "mh_record_end": (_i32, [C.POINTER(_p)]),
    "mh_record_abort": (_i32, []),
    "mh_record_event": (_i32, [_p, C.POINTER(_i64)]),
    "mh_record_wait_event": (_i32, [_p, _i64]),
    "mh_record_replay": (_i32, [_p]),
    "mh_record_info": (_i32, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "mh_record_free": (_i32, [_p]),
    "mh_activation": (_i32, [_i32, _p, _i64, _p, _i64, _p, _i64, _i64, _i32, _p]),
    "mh_fill_columns": (_i32, [_p, _i64, _i64, _i32, _i32, _f32, _p]),
    "mh_bce_mean_partial": (_i32, [_p, _p, _i64, _f32, _p, _p, _p]),
    "mh_bce_mean_finish": (_i32, [_p, _i64, _p, _p]),
}

_LIB = None


class MerlinHipError(RuntimeError):
    pass


def lib_path() -> Path:
    """The in-tree library, or the file named by MERLIN_HIP_LIB (a prebuilt libmerlin_hip.so elsewhere: deployment
    override, also how two builds are compared on one box).  An override is never rebuilt."""
    override = os.environ.get("MERLIN_HIP_LIB")
    return Path(override) if override else _build.LIB


def load() -> C.CDLL:
    """Load (building first if the in-tree .so is absent or stale and hipcc is available)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be resident BEFORE
    # libmerlin_hip.so is dlopen'ed so that both share ONE HIP runtime (streams, allocations);
    # loading ours first would pull in /opt/rocm's copy and torch would then load a second one.
    import torch  # noqa: F401  (buffer carrier + the process-wide HIP runtime)

    path = lib_path()
    if path != _build.LIB:
        if not path.exists():
            raise MerlinHipError(f"MERLIN_HIP_LIB={path} does not exist")
    elif _build.needs_build():
        if shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists():
            _build.build()
        elif not path.exists():
            raise MerlinHipError(
                f"{path} is missing and hipcc is not available: the HIP hot path cannot run "
                "(there is no CPU fallback). Run `python -m models_amd.build` on a ROCm box."
            )
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("MERLIN_HIP_TRACE") == "1":
        lib = _Traced(lib)
    _LIB = lib
    return lib


class _Traced:
    """Debug aid (MERLIN_HIP_TRACE=1): print every C-ABI call before it is issued and synchronise after it, so that a
    GPU memory fault (which aborts the process) is attributed to the call that caused it."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in ("mh_last_error", "mh_version", "mh_mlp_chain_supported") or name.endswith("_workspace_bytes"):
            return fn

        def call(*a):
            import sys

            import torch

            print(f"[mh-trace] {name}", file=sys.stderr, flush=True)
            r = fn(*a)
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
            return r

        return call


def check(status: int, what: str) -> None:
    if status != MH_OK:
        msg = load().mh_last_error().decode("utf-8", "replace")
        raise MerlinHipError(f"{what} failed with status {status}: {msg}")
