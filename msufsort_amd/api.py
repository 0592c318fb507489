"""Host-side mirror of the reference's public interface for the hot path.

Reference surface being mirrored (reference src/library/msufsort/msufsort.h):
  class maniscalco::msufsort(threads)                                        h:42-75
      make_suffix_array(begin, end) -> suffix_array (n+1 int32, [0] = n)    h:57-61
      forward_burrows_wheeler_transform(begin, end) -> sentinel row         h:63-67   (in place)
      static reverse_burrows_wheeler_transform(begin, end, sentinel, thr)   h:69-75   (in place)
  free templates make_suffix_array / forward_... / reverse_...              h:403-476
The demo's LCP (reference src/executable/msufsort/main.cpp:143-159) is exposed as make_lcp_array.

Python buffers are immutable-friendly, so the "in place" transforms return new arrays; argument
meaning and conventions are the reference's.  `threads` is accepted for signature compatibility
and ignored: the work runs on the GPU.  Every function goes through the C-ABI
(include/msufsort_hip.h); nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MsufsortHipError, Opts, Timings  # noqa: F401


def _u8(data) -> np.ndarray:
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    a = np.ascontiguousarray(data)
    if a.dtype != np.uint8:
        if a.dtype == np.int8:
            a = a.view(np.uint8)          # the reference reinterprets int8 as uint8 (h:444)
        else:
            raise TypeError("input must be bytes or a uint8/int8 array")
    return a


def _opts(device=0, verbose=0, text_rounds=0, shard=0, n_shards=1) -> Opts:
    o = Opts()
    o.device, o.verbose, o.text_rounds, o.shard, o.n_shards = device, verbose, text_rounds, shard, n_shards
    return o


def device_count() -> int:
    return int(_lib.lib().msufsort_hip_device_count())


def make_suffix_array(data, threads: int = 1, *, device: int = 0, verbose: int = 0, text_rounds: int = 0) -> np.ndarray:
    """maniscalco::make_suffix_array (h:432-445): n+1 int32 entries, SA[0] = n."""
    t = _u8(data)
    sa = np.empty(t.size + 1, dtype=np.int32)
    o = _opts(device, verbose, text_rounds)
    _lib.check(_lib.lib().msufsort_hip_make_sa_i32(t.ctypes.data, t.size, sa.ctypes.data, C.byref(o)), "make_suffix_array")
    return sa


def make_suffix_array_i64(data, threads: int = 1, *, device: int = 0) -> np.ndarray:
    """The same rows as int64 (msufsort_hip_make_sa_i64; inputs above 2^31 - 2 bytes are rejected in this round)."""
    t = _u8(data)
    sa = np.empty(t.size + 1, dtype=np.int64)
    o = _opts(device)
    _lib.check(_lib.lib().msufsort_hip_make_sa_i64(t.ctypes.data, t.size, sa.ctypes.data, C.byref(o)), "make_suffix_array_i64")
    return sa


def forward_burrows_wheeler_transform(data, threads: int = 1, *, device: int = 0):
    """maniscalco::forward_burrows_wheeler_transform (h:449-462): returns (bwt bytes, sentinel row)."""
    t = _u8(data).copy()
    s = C.c_int64(0)
    o = _opts(device)
    _lib.check(_lib.lib().msufsort_hip_forward_bwt(t.ctypes.data, t.size, C.byref(s), C.byref(o)), "forward_bwt")
    return t, int(s.value)


def reverse_burrows_wheeler_transform(bwt, sentinel_index: int, threads: int = 1, *, device: int = 0) -> np.ndarray:
    """maniscalco::reverse_burrows_wheeler_transform (h:466-476): returns the original text."""
    t = _u8(bwt).copy()
    o = _opts(device)
    _lib.check(_lib.lib().msufsort_hip_inverse_bwt(t.ctypes.data, t.size, int(sentinel_index), C.byref(o)), "inverse_bwt")
    return t


def make_lcp_array(data, sa, threads: int = 1, *, device: int = 0) -> np.ndarray:
    """Demo LCP (main.cpp:143-159): out[i] = lcp(SA[i+1], SA[i+2]); out[n-1] = 0."""
    t = _u8(data)
    sa = np.ascontiguousarray(sa, dtype=np.int32)
    if sa.size != t.size + 1:
        raise ValueError("sa must have n+1 entries")
    out = np.zeros(t.size, dtype=np.int32)
    o = _opts(device)
    _lib.check(_lib.lib().msufsort_hip_lcp_i32(t.ctypes.data, t.size, sa.ctypes.data, out.ctypes.data, C.byref(o)), "lcp")
    return out


class msufsort:
    """Mirror of class maniscalco::msufsort (h:42-75)."""

    def __init__(self, threads: int = 1, device: int = 0):
        self.threads = threads
        self.device = device

    def make_suffix_array(self, data):
        return make_suffix_array(data, self.threads, device=self.device)

    def forward_burrows_wheeler_transform(self, data):
        return forward_burrows_wheeler_transform(data, self.threads, device=self.device)

    @staticmethod
    def reverse_burrows_wheeler_transform(bwt, sentinel_index, threads=1):
        return reverse_burrows_wheeler_transform(bwt, sentinel_index, threads)


class DeviceContext:
    """HBM-resident path: one stream + workspace (msufsort_hip_ctx).  Takes torch CUDA tensors
    (or raw device pointers) so bench.py can time with the input already in HBM."""

    def __init__(self, device: int = 0, max_n: int = 0):
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.msufsort_hip_ctx_create(C.byref(h), device, max_n), "ctx_create")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            self._L.msufsort_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return int(self._L.msufsort_hip_ctx_stream(self._h) or 0)

    def timings(self) -> Timings:
        t = Timings()
        _lib.check(self._L.msufsort_hip_last_timings(self._h, C.byref(t)), "timings")
        return t

    @staticmethod
    def _ptr(x) -> int:
        """Raw device pointer of a torch tensor (or an int).  The engine runs on its own HIP stream, so pending
        work on torch's current stream (the upload that filled the tensor) is drained first."""
        if hasattr(x, "data_ptr"):
            if getattr(x, "is_cuda", False):
                import torch
                torch.cuda.current_stream(x.device).synchronize()
            return int(x.data_ptr())
        return int(x)

    def make_sa(self, d_text, n: int, d_sa, *, verbose=0, text_rounds=0):
        """d_text: >= n+64 bytes in HBM; d_sa: n+1 int32 in HBM."""
        o = _opts(self.device, verbose, text_rounds)
        _lib.check(self._L.msufsort_hip_make_sa_i32_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa), C.byref(o)), "make_sa_dev")

    def make_sa_i64(self, d_text, n: int, d_sa64, *, verbose=0, text_rounds=0):
        """d_text: >= n+64 bytes in HBM; d_sa64: n+1 int64 in HBM (the int32 rows, widened on the device)."""
        o = _opts(self.device, verbose, text_rounds)
        _lib.check(self._L.msufsort_hip_make_sa_i64_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa64), C.byref(o)), "make_sa_i64_dev")

    def shard_bounds(self, d_text, n: int, n_shards: int):
        b = (C.c_int64 * (n_shards + 1))()
        _lib.check(self._L.msufsort_hip_shard_bounds_dev(self._h, self._ptr(d_text), n, n_shards, b), "shard_bounds")
        return [int(x) for x in b]

    def make_sa_shard(self, d_text, n: int, d_slice, capacity: int, shard: int, n_shards: int, *, verbose=0, text_rounds=0):
        o = _opts(self.device, verbose, text_rounds, shard, n_shards)
        lo, hi = C.c_int64(0), C.c_int64(0)
        _lib.check(self._L.msufsort_hip_make_sa_shard_dev(self._h, self._ptr(d_text), n, self._ptr(d_slice), capacity,
                                                          C.byref(lo), C.byref(hi), C.byref(o)), "make_sa_shard")
        return int(lo.value), int(hi.value)

    def make_sa_shard_groups(self, d_text, n: int, d_slice, d_grp_slice, capacity: int, shard: int, n_shards: int, *, verbose=0, text_rounds=0):
        """Like make_sa_shard, plus the tie-group heads of the slice rows.  Returns (lo, hi, unresolved, depth)."""
        o = _opts(self.device, verbose, text_rounds, shard, n_shards)
        lo, hi, depth = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        r = self._L.msufsort_hip_make_sa_shard_groups_dev(self._h, self._ptr(d_text), n, self._ptr(d_slice), self._ptr(d_grp_slice), capacity,
                                                          C.byref(lo), C.byref(hi), C.byref(depth), C.byref(o))
        if r not in (0, 1):
            _lib.check(r, "make_sa_shard_groups")
        return int(lo.value), int(hi.value), r == 1, int(depth.value)

    def finish_sa(self, d_text, n: int, d_sa_full, d_grp_full, depth: int, *, verbose=0):
        o = _opts(self.device, verbose)
        _lib.check(self._L.msufsort_hip_finish_sa_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa_full), self._ptr(d_grp_full), depth, C.byref(o)), "finish_sa")

    def bwt_from_sa(self, d_text, n: int, d_sa, d_bwt) -> int:
        s = C.c_int64(0)
        _lib.check(self._L.msufsort_hip_bwt_from_sa_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa), self._ptr(d_bwt), C.byref(s)), "bwt_from_sa")
        return int(s.value)

    def forward_bwt(self, d_text, n: int, d_bwt) -> int:
        s = C.c_int64(0)
        o = _opts(self.device)
        _lib.check(self._L.msufsort_hip_forward_bwt_dev(self._h, self._ptr(d_text), n, self._ptr(d_bwt), C.byref(s), C.byref(o)), "forward_bwt_dev")
        return int(s.value)

    def inverse_bwt(self, d_bwt, n: int, sentinel: int, d_out):
        o = _opts(self.device)
        _lib.check(self._L.msufsort_hip_inverse_bwt_dev(self._h, self._ptr(d_bwt), n, sentinel, self._ptr(d_out), C.byref(o)), "inverse_bwt_dev")

    def lcp(self, d_text, n: int, d_sa, d_lcp):
        _lib.check(self._L.msufsort_hip_lcp_i32_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa), self._ptr(d_lcp)), "lcp_dev")

    def validate_sa(self, d_text, n: int, d_sa) -> int:
        e = C.c_int64(0)
        _lib.check(self._L.msufsort_hip_validate_sa_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa), C.byref(e)), "validate")
        return int(e.value)

    def debug_hist16(self, d_text, n: int, d_hist):
        _lib.check(self._L.msufsort_hip_debug_hist16_dev(self._h, self._ptr(d_text), n, self._ptr(d_hist)), "hist16")
