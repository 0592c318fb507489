"""ctypes binding of the C-ABI in include/msufsort_hip.h (the drop-in boundary).

The HIP library is the product: if it is missing or cannot be loaded this module raises -
there is no CPU or pure-Python fallback here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSUFSORT_HIP_LIB selects an alternative build of the same C-ABI (profiling / experiment builds); the product library
# in msufsort_amd/lib/ is never overwritten by tooling.
LIB_PATH = os.environ.get("MSUFSORT_HIP_LIB") or os.path.join(_HERE, "lib", "libmsufsort_hip.so")

TEXT_PAD = 64


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("shard", C.c_int32), ("n_shards", C.c_int32),
                ("text_rounds", C.c_int32), ("verbose", C.c_int32), ("force_wide", C.c_int32), ("two_stage", C.c_int32), ("reuse_plan", C.c_int32),
                ("reserved", C.c_int32 * 8)]


class Timings(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("hist16_ms", C.c_double), ("scatter0_ms", C.c_double),
                ("scatter1_ms", C.c_double), ("bucket_sort_ms", C.c_double), ("refine_ms", C.c_double),
                ("other_ms", C.c_double), ("n", C.c_int64), ("m", C.c_int64), ("rounds", C.c_int32),
                ("doubling_rounds", C.c_int32), ("unresolved_after_round0", C.c_int64),
                ("stop_depth", C.c_int64), ("logical_shards", C.c_int64), ("gathered_records", C.c_int64),
                ("ibwt_walk_us", C.c_int64), ("ibwt_total_us", C.c_int64), ("bstar_suffixes", C.c_int64),
                ("induction_launches", C.c_int64), ("b_suffixes", C.c_int64), ("front_ms", C.c_double), ("fallbacks", C.c_int64),
                ("progression_suffixes", C.c_int64), ("bucket_sort_handed_back", C.c_int64), ("hist17_ms", C.c_double), ("radix_bits", C.c_int64),
                ("key1_records", C.c_int64), ("doubling_records", C.c_int64)]


# every symbol include/msufsort_hip.h declares (checked by tests/test_cabi.py)
SYMBOLS = [
    "msufsort_hip_device_count", "msufsort_hip_strerror", "msufsort_hip_last_error", "msufsort_hip_build_id",
    "msufsort_hip_ctx_create", "msufsort_hip_ctx_destroy", "msufsort_hip_ctx_stream", "msufsort_hip_ctx_sync",
    "msufsort_hip_last_timings", "msufsort_hip_make_sa_i32", "msufsort_hip_make_sa_i32_dev",
    "msufsort_hip_make_sa_shard_dev", "msufsort_hip_shard_bounds_dev", "msufsort_hip_plan_cuts",
    "msufsort_hip_make_sa_shard_groups_dev", "msufsort_hip_make_sa_shard_groups_i64_dev", "msufsort_hip_isa_from_slice_dev",
    "msufsort_hip_double_sort_dev", "msufsort_hip_emit_updates_dev", "msufsort_hip_apply_updates_dev", "msufsort_hip_ctx_trim",
    "msufsort_hip_validate_sa_i64_dev", "msufsort_hip_bwt_from_sa_i64_dev", "msufsort_hip_make_sa_multi", "msufsort_hip_release_cached", "msufsort_hip_forward_bwt",
    "msufsort_hip_forward_bwt_dev", "msufsort_hip_bwt_from_sa_dev", "msufsort_hip_inverse_bwt",
    "msufsort_hip_inverse_bwt_dev", "msufsort_hip_lcp_i32", "msufsort_hip_lcp_i32_dev",
    "msufsort_hip_validate_sa_dev", "msufsort_hip_debug_hist16_dev",
    "msufsort_hip_make_sa_i32_ctx", "msufsort_hip_forward_bwt_ctx", "msufsort_hip_inverse_bwt_ctx", "msufsort_hip_lcp_i32_ctx",
    "msufsort_hip_make_sa_i64", "msufsort_hip_make_sa_i64_ctx", "msufsort_hip_make_sa_i64_dev",
    "msufsort_hip_make_sa_two_stage_sharded_dev", "msufsort_hip_bwt_slice_dev", "msufsort_hip_forward_bwt_multi",
    "msufsort_hip_hist_part_dev", "msufsort_hip_hist_plan_dev", "msufsort_hip_hist_install_dev",
]
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32)

_lib = None


class MsufsortHipError(RuntimeError):
    pass


def lib():
    """Loads libmsufsort_hip.so (built by msufsort_amd/csrc/Makefile); raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsufsortHipError(
            f"{LIB_PATH} is missing: build it with `make -C msufsort_amd/csrc` "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process leave the
    # second one without devices.  Load torch's first (when torch is installed) so that this library
    # binds to the same runtime.  torch is plumbing here (device memory, streams, torch.distributed).
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001
        pass
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    L.msufsort_hip_device_count.restype = C.c_int
    L.msufsort_hip_strerror.restype = C.c_char_p
    L.msufsort_hip_strerror.argtypes = [C.c_int]
    L.msufsort_hip_last_error.restype = C.c_char_p
    L.msufsort_hip_build_id.restype = C.c_char_p
    L.msufsort_hip_ctx_create.argtypes = [C.POINTER(vp), i32, i64]
    L.msufsort_hip_ctx_destroy.argtypes = [vp]
    L.msufsort_hip_ctx_destroy.restype = None
    L.msufsort_hip_ctx_stream.argtypes = [vp]
    L.msufsort_hip_ctx_stream.restype = vp
    L.msufsort_hip_ctx_sync.argtypes = [vp]
    L.msufsort_hip_last_timings.argtypes = [vp, C.POINTER(Timings)]
    L.msufsort_hip_make_sa_i32.argtypes = [vp, i64, vp, C.POINTER(Opts)]
    L.msufsort_hip_make_sa_i32_dev.argtypes = [vp, vp, i64, vp, C.POINTER(Opts)]
    L.msufsort_hip_make_sa_shard_dev.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(Opts)]
    L.msufsort_hip_make_sa_shard_groups_dev.argtypes = [vp, vp, i64, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(Opts)]
    L.msufsort_hip_make_sa_shard_groups_i64_dev.argtypes = [vp, vp, i64, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(Opts)]
    L.msufsort_hip_isa_from_slice_dev.argtypes = [vp, vp, vp, i64, i64, vp, i32]
    L.msufsort_hip_double_sort_dev.argtypes = [vp, i64, vp, vp, vp, i64, i64, vp, i64, i32, C.POINTER(Opts), C.POINTER(i64), C.POINTER(i64)]
    L.msufsort_hip_emit_updates_dev.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i64, vp, i64, i32, C.POINTER(i64), C.POINTER(i64)]
    L.msufsort_hip_apply_updates_dev.argtypes = [vp, vp, i64, vp, i32]
    L.msufsort_hip_ctx_trim.argtypes = [vp]
    L.msufsort_hip_release_cached.restype = None
    L.msufsort_hip_make_sa_multi.argtypes = [vp, i32, vp, i64, vp, i32, C.POINTER(Opts), C.POINTER(Timings)]
    L.msufsort_hip_validate_sa_i64_dev.argtypes = [vp, vp, i64, vp, C.POINTER(i64)]
    L.msufsort_hip_bwt_from_sa_i64_dev.argtypes = [vp, vp, i64, vp, vp, C.POINTER(i64)]
    L.msufsort_hip_shard_bounds_dev.argtypes = [vp, vp, i64, i32, C.POINTER(i64)]
    L.msufsort_hip_plan_cuts.argtypes = [vp, i64, i64, i32, vp, vp]
    L.msufsort_hip_hist_part_dev.argtypes = [vp, vp, i64, i32, i32, vp, C.POINTER(i32)]
    L.msufsort_hip_hist_plan_dev.argtypes = [vp, vp, i64, i32, vp, vp, i32, C.POINTER(i64)]
    L.msufsort_hip_hist_install_dev.argtypes = [vp, i32, vp, i32]
    L.msufsort_hip_forward_bwt.argtypes = [vp, i64, C.POINTER(i64), C.POINTER(Opts)]
    L.msufsort_hip_forward_bwt_dev.argtypes = [vp, vp, i64, vp, C.POINTER(i64), C.POINTER(Opts)]
    L.msufsort_hip_bwt_from_sa_dev.argtypes = [vp, vp, i64, vp, vp, C.POINTER(i64)]
    L.msufsort_hip_inverse_bwt.argtypes = [vp, i64, i64, C.POINTER(Opts)]
    L.msufsort_hip_inverse_bwt_dev.argtypes = [vp, vp, i64, i64, vp, C.POINTER(Opts)]
    L.msufsort_hip_lcp_i32.argtypes = [vp, i64, vp, vp, C.POINTER(Opts)]
    L.msufsort_hip_lcp_i32_dev.argtypes = [vp, vp, i64, vp, vp]
    L.msufsort_hip_validate_sa_dev.argtypes = [vp, vp, i64, vp, C.POINTER(i64)]
    L.msufsort_hip_debug_hist16_dev.argtypes = [vp, vp, i64, vp]
    L.msufsort_hip_make_sa_i32_ctx.argtypes = [vp, vp, i64, vp, C.POINTER(Opts)]
    L.msufsort_hip_make_sa_i64.argtypes = [vp, i64, vp, C.POINTER(Opts)]
    L.msufsort_hip_make_sa_i64_ctx.argtypes = [vp, vp, i64, vp, C.POINTER(Opts)]
    L.msufsort_hip_make_sa_i64_dev.argtypes = [vp, vp, i64, vp, C.POINTER(Opts)]
    L.msufsort_hip_make_sa_two_stage_sharded_dev.argtypes = [vp, vp, i64, vp, vp, i64, EXCHANGE_FN, vp, C.POINTER(Opts)]
    L.msufsort_hip_forward_bwt_multi.argtypes = [vp, i32, vp, i64, C.POINTER(i64), C.POINTER(Opts), C.POINTER(Timings)]
    L.msufsort_hip_bwt_slice_dev.argtypes = [vp, vp, i64, vp, i64, i64, i32, vp, C.POINTER(i64)]
    L.msufsort_hip_forward_bwt_ctx.argtypes = [vp, vp, i64, C.POINTER(i64), C.POINTER(Opts)]
    L.msufsort_hip_inverse_bwt_ctx.argtypes = [vp, vp, i64, i64, C.POINTER(Opts)]
    L.msufsort_hip_lcp_i32_ctx.argtypes = [vp, vp, i64, vp, vp]
    _lib = L
    return L


def check(status: int, what: str = "") -> None:
    if status != 0:
        L = lib()
        msg = L.msufsort_hip_strerror(status).decode()
        detail = L.msufsort_hip_last_error().decode()
        raise MsufsortHipError(f"{what}: {msg} ({status}) {detail}")
