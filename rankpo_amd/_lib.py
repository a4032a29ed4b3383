"""ctypes binding of librankpo_hip.so (C ABI in include/rankpo_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librankpo_hip.so")

# enums of include/rankpo_hip.h
RPO_OK = 0
RPO_DT_F32, RPO_DT_BF16, RPO_DT_F16 = 0, 1, 2
RPO_POOL_LAST, RPO_POOL_CLS = 0, 1
RPO_TARGET_INBATCH, RPO_TARGET_FIRST = 0, 1
RPO_LOSS_SIGMOID, RPO_LOSS_HINGE = 0, 1
RPO_BUILD_ONEWAVE64 = 1
RPO_NUM_METRICS = 9
METRIC_KEYS = (
    "rankpo_loss", "sft_loss", "rewards/chosen", "rewards/rejected", "rewards/accuracies",
    "rewards/margins", "scores/chosen", "scores/rejected", "scores/margins",
)


class RankPOParams(C.Structure):
    _fields_ = [
        ("beta", C.c_float), ("temperature", C.c_float), ("gamma_beta_ratio", C.c_float),
        ("label_smoothing", C.c_float), ("rankpo_weight", C.c_float), ("sft_weight", C.c_float),
        ("loss_type", C.c_int32), ("reference_free", C.c_int32),
    ]


class RankPOHipError(RuntimeError):
    pass


_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/rankpo_hip.h one to one
SIGNATURES = {
    "rpo_version": (C.c_int, []),
    "rpo_build_flags": (C.c_int, []),
    "rpo_status_string": (C.c_char_p, [C.c_int]),
    "rpo_last_hip_error": (C.c_char_p, []),
    "rpo_pool_normalize_fwd": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "rpo_pool_normalize_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _vp]),
    "rpo_infonce_workspace_bytes": (_sz, [_i64, _i64, _i64, _i32]),
    "rpo_infonce_debug_poison_tickets": (C.c_int, [C.c_uint64, _vp]),
    "rpo_infonce_fwd": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rpo_infonce_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _f32, _i32, _i64, _i64, _i64, _i64,
                                  _vp, _vp, _vp, _sz, _vp]),
    "rpo_infonce_ds": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _f32, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "rpo_sim_gemm_nt": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "rpo_rankpo_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, C.POINTER(RankPOParams), _vp, _vp, _vp, _vp, _vp, _vp]),
    "rpo_rankpo_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "rpo_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
    "rpo_sumsq_partial": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp]),
    "rpo_swiglu_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "rpo_topk_merge": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _vp]),
    "rpo_topk_merge_split": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "rpo_sim_topk_filter_ok": (C.c_int, [_i64, _i64, _i64]),
    "rpo_sim_topk_filter": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "rpo_sim_scores_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp]),
    "rpo_topk_merge_candidates": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "rpo_swiglu_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "rpo_swiglu_bwd_t": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "rpo_transpose": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "rpo_add_rmsnorm_waves": (C.c_int, [_i64]),
    "rpo_add_rmsnorm_fwd": (C.c_int, [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "rpo_add_rmsnorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "rpo_flash_attn_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp,
                                     _i64, _vp, _i64, _vp, _vp, _i64, _i64, _vp]),
    "rpo_flash_attn_bwd": (C.c_int, [_vp] * 5 + [_i64] * 5 + [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp, _vp,
                                     _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _vp]),
    "rpo_lastq_attn_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _vp]),
    "rpo_lastq_attn_bwd": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _i64, _vp,
                                     _vp, _i64, _vp, _vp, _i64, _i64, _vp]),
    "rpo_rope": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp]),
}

_lib = None


def load():
    """Load (once) and return the library.  Raises ImportError with the build command when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP scoring library is not built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C rankpo_amd/csrc`).  "
            "rankpo_amd has no CPU fallback for its hot path.")
    # PyTorch ships its own libamdhip64.so.7; the process must use ONE HIP runtime, the one that owns torch's
    # streams and allocations.  Importing torch first makes the dynamic loader bind librankpo_hip.so's
    # libamdhip64.so.7 dependency to that already-loaded copy instead of /opt/rocm's.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # stale build
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it (make -C rankpo_amd/csrc)") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != RPO_OK:
        msg = load().rpo_status_string(rc).decode()
        if rc == -4:
            msg += ": " + load().rpo_last_hip_error().decode()
        raise RankPOHipError(f"{what} failed: {msg} (status {rc})")
