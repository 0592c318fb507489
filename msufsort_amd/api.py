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


def _opts(device=0, verbose=0, text_rounds=0, shard=0, n_shards=1, force_wide=0, two_stage=0, reuse_plan=0) -> Opts:
    o = Opts()
    o.device, o.verbose, o.text_rounds, o.shard, o.n_shards, o.force_wide = device, verbose, text_rounds, shard, n_shards, int(force_wide)
    o.two_stage = int(two_stage)
    o.reuse_plan = int(reuse_plan)
    return o


def device_count() -> int:
    return int(_lib.lib().msufsort_hip_device_count())


def make_suffix_array(data, threads: int = 1, *, device: int = 0, verbose: int = 0, text_rounds: int = 0, two_stage: int = 0) -> np.ndarray:
    """maniscalco::make_suffix_array (h:432-445): n+1 int32 entries, SA[0] = n."""
    t = _u8(data)
    sa = np.empty(t.size + 1, dtype=np.int32)
    o = _opts(device, verbose, text_rounds, two_stage=two_stage)
    _lib.check(_lib.lib().msufsort_hip_make_sa_i32(t.ctypes.data, t.size, sa.ctypes.data, C.byref(o)), "make_suffix_array")
    return sa


def make_suffix_array_i64(data, threads: int = 1, *, device: int = 0, force_wide: bool = False, n_shards: int = 1,
                          text_rounds: int = 0, verbose: int = 0) -> np.ndarray:
    """The same rows as int64 (msufsort_hip_make_sa_i64).  Inputs above 2^31 - 2 bytes - and any input with
    force_wide - run the wide engine (40-bit indices, logical shards, distributed prefix doubling)."""
    t = _u8(data)
    sa = np.empty(t.size + 1, dtype=np.int64)
    o = _opts(device, verbose, text_rounds, 0, n_shards, force_wide)
    _lib.check(_lib.lib().msufsort_hip_make_sa_i64(t.ctypes.data, t.size, sa.ctypes.data, C.byref(o)), "make_suffix_array_i64")
    return sa


def make_suffix_array_multi(data, devices=None, *, index_bytes: int = 4, n_shards: int = 0, text_rounds: int = 0, force_wide: bool = False,
                            verbose: int = 0, timings: bool = False, two_stage: int = 0):
    """msufsort_hip_make_sa_multi: one process, several GPUs (default: MSUFSORT_DEVICES, else all visible), host text in,
    host suffix array out; finished slices stream to the host while the remaining key ranges are sorted."""
    t = _u8(data)
    sa = np.empty(t.size + 1, dtype=np.int64 if index_bytes == 8 else np.int32)
    o = _opts(0, verbose, text_rounds, 0, n_shards, force_wide, two_stage=two_stage)
    tm = Timings()
    dv = (C.c_int32 * len(devices))(*devices) if devices else None
    _lib.check(_lib.lib().msufsort_hip_make_sa_multi(dv, len(devices) if devices else 0, t.ctypes.data, t.size, sa.ctypes.data, index_bytes,
                                                    C.byref(o), C.byref(tm)), "make_suffix_array_multi")
    return (sa, tm) if timings else sa


def forward_burrows_wheeler_transform(data, threads: int = 1, *, device: int = 0, two_stage: int = 0):
    """maniscalco::forward_burrows_wheeler_transform (h:449-462): returns (bwt bytes, sentinel row)."""
    t = _u8(data).copy()
    s = C.c_int64(0)
    o = _opts(device, two_stage=two_stage)
    _lib.check(_lib.lib().msufsort_hip_forward_bwt(t.ctypes.data, t.size, C.byref(s), C.byref(o)), "forward_bwt")
    return t, int(s.value)


def forward_burrows_wheeler_transform_multi(data, devices=None, *, n_shards: int = 0, force_wide: bool = False, text_rounds: int = 0, timings: bool = False):
    """msufsort_hip_forward_bwt_multi: one process, the listed GPUs; the BWT bytes of finished key-range slices stream to the host
    while the remaining shards are sorted.  Returns (bwt bytes, sentinel row[, timings])."""
    t = _u8(data).copy()
    s = C.c_int64(0)
    o = _opts(0, 0, text_rounds, 0, n_shards, force_wide)
    tm = Timings()
    dv = (C.c_int32 * len(devices))(*devices) if devices else None
    _lib.check(_lib.lib().msufsort_hip_forward_bwt_multi(dv, len(devices) if devices else 0, t.ctypes.data, t.size, C.byref(s), C.byref(o), C.byref(tm)), "forward_bwt_multi")
    return (t, int(s.value), tm) if timings else (t, int(s.value))


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

    def make_sa(self, d_text, n: int, d_sa, *, verbose=0, text_rounds=0, logical_shards=0, two_stage=0):
        """d_text: >= n+64 bytes in HBM; d_sa: n+1 int32 in HBM.  logical_shards > 1: the shards of a multi-GPU build,
        one after the other on this GPU (same code path as the distributed build, bounded workspace).  two_stage: 0 = B* sort +
        induction when the input looks like text, 1 = whenever possible, -1 = never (sort all suffixes)."""
        o = _opts(self.device, verbose, text_rounds, -1 if logical_shards > 1 else 0, max(logical_shards, 1), two_stage=two_stage)
        _lib.check(self._L.msufsort_hip_make_sa_i32_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa), C.byref(o)), "make_sa_dev")

    def make_sa_i64(self, d_text, n: int, d_sa64, *, verbose=0, text_rounds=0, force_wide=False, n_shards=1):
        """d_text: >= n+64 bytes in HBM; d_sa64: n+1 int64 in HBM.  n > 2^31 - 2 (or force_wide): the wide engine with at
        least n_shards logical shards."""
        o = _opts(self.device, verbose, text_rounds, 0, n_shards, force_wide)
        _lib.check(self._L.msufsort_hip_make_sa_i64_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa64), C.byref(o)), "make_sa_i64_dev")

    def make_sa_two_stage_sharded(self, d_text, n: int, d_sa, d_bstar, shard: int, n_shards: int, exchange=None, *, two_stage=0, verbose=0):
        """msufsort_hip_make_sa_two_stage_sharded_dev: B* suffixes sorted by key-range shards (shard = -1: all of them here), the rest
        induced from the complete sorted-B* array on every rank.  exchange(bounds: list, my_status: int) -> agreed status is the
        caller's collective.  Returns 0 (d_sa complete), 1 (declined on every rank), 2 (failed on this rank after the exchange)."""
        o = _opts(self.device, verbose, 0, shard, n_shards, two_stage=two_stage)

        def _cb(user, bounds, ns, status):
            try:
                return int(exchange([int(bounds[i]) for i in range(ns + 1)], int(status)))
            except Exception:  # noqa: BLE001  (an exception must not unwind through the C caller)
                import traceback
                traceback.print_exc()
                return -1
        cb = _lib.EXCHANGE_FN(_cb) if exchange is not None else C.cast(None, _lib.EXCHANGE_FN)
        r = self._L.msufsort_hip_make_sa_two_stage_sharded_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa), self._ptr(d_bstar), int(d_bstar.numel()),
                                                               cb, None, C.byref(o))
        if r not in (0, 1, 2):
            _lib.check(r, "make_sa_two_stage_sharded")
        return r

    def trim(self):
        _lib.check(self._L.msufsort_hip_ctx_trim(self._h), "ctx_trim")

    def shard_bounds(self, d_text, n: int, n_shards: int):
        b = (C.c_int64 * (n_shards + 1))()
        _lib.check(self._L.msufsort_hip_shard_bounds_dev(self._h, self._ptr(d_text), n, n_shards, b), "shard_bounds")
        return [int(x) for x in b]

    # ---- the 16-bit histogram computed sharded (include/msufsort_hip.h; driver: dist.plan_sharded) ----
    def hist_part(self, d_text, n: int, part: int, parts: int, d_hist):
        """Counts my stripes of the text into d_hist (65,536 x int64, device).  Returns (stripes in all, my first, my end)."""
        assert d_hist.numel() == 65536 and d_hist.element_size() == 8
        st = (C.c_int32 * 3)()
        _lib.check(self._L.msufsort_hip_hist_part_dev(self._h, self._ptr(d_text), n, part, parts, self._ptr(d_hist), st), "hist_part")
        return int(st[0]), int(st[1]), int(st[2])

    def hist_plan(self, d_text, n: int, n_shards: int, d_hist_sum, d_sums):
        """d_sums: [n_shards][stripes_per_part][256] int32 (device).  Returns the slice bounds, or None when the plan needs a
        replicated histogram after all (a shard boundary inside a heavy two-byte key: DNA, text)."""
        assert d_sums.dim() == 3 and d_sums.shape[0] == n_shards and d_sums.shape[2] == 256 and d_sums.element_size() == 4 and d_sums.is_contiguous()
        b = (C.c_int64 * (n_shards + 1))()
        r = self._L.msufsort_hip_hist_plan_dev(self._h, self._ptr(d_text), n, n_shards, self._ptr(d_hist_sum), self._ptr(d_sums), int(d_sums.shape[1]), b)
        if r == 3:
            return None
        _lib.check(r, "hist_plan")
        return [int(x) for x in b]

    def hist_install(self, shard: int, d_stripe_sums):
        assert d_stripe_sums.dim() == 2 and d_stripe_sums.shape[1] == 256 and d_stripe_sums.element_size() == 4 and d_stripe_sums.is_contiguous()
        _lib.check(self._L.msufsort_hip_hist_install_dev(self._h, shard, self._ptr(d_stripe_sums), int(d_stripe_sums.shape[0])), "hist_install")

    def make_sa_shard(self, d_text, n: int, d_slice, capacity: int, shard: int, n_shards: int, *, verbose=0, text_rounds=0, reuse_plan=False):
        """reuse_plan: this is ANOTHER shard of the text the previous shard call on this context planned (same n_shards, contents
        unchanged): histogram and cuts are taken from that call (msufsort_hip_opts.reuse_plan)."""
        o = _opts(self.device, verbose, text_rounds, shard, n_shards, reuse_plan=reuse_plan)
        lo, hi = C.c_int64(0), C.c_int64(0)
        _lib.check(self._L.msufsort_hip_make_sa_shard_dev(self._h, self._ptr(d_text), n, self._ptr(d_slice), capacity,
                                                          C.byref(lo), C.byref(hi), C.byref(o)), "make_sa_shard")
        return int(lo.value), int(hi.value)

    def make_sa_shard_groups(self, d_text, n: int, d_slice, d_grp_slice, capacity: int, shard: int, n_shards: int, *, verbose=0, text_rounds=0,
                             index_bytes=4, reuse_plan=False):
        """Like make_sa_shard, plus the tie-group heads of the slice rows (uint32, relative to the slice).  index_bytes = 8:
        the wide engine (int64 rows).  Returns (lo, hi, unresolved, depth)."""
        o = _opts(self.device, verbose, text_rounds, shard, n_shards, reuse_plan=reuse_plan)
        lo, hi, depth = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        f = self._L.msufsort_hip_make_sa_shard_groups_i64_dev if index_bytes == 8 else self._L.msufsort_hip_make_sa_shard_groups_dev
        r = f(self._h, self._ptr(d_text), n, self._ptr(d_slice), self._ptr(d_grp_slice), capacity, C.byref(lo), C.byref(hi), C.byref(depth), C.byref(o))
        if r not in (0, 1):
            _lib.check(r, "make_sa_shard_groups")
        return int(lo.value), int(hi.value), r == 1, int(depth.value)

    # ---- distributed prefix doubling (include/msufsort_hip.h) ----
    def isa_from_slice(self, d_sa_slice, d_grp_slice, lo: int, hi: int, d_isa, index_bytes=4):
        _lib.check(self._L.msufsort_hip_isa_from_slice_dev(self._h, self._ptr(d_sa_slice), self._ptr(d_grp_slice), lo, hi, self._ptr(d_isa), index_bytes), "isa_from_slice")

    def double_sort(self, n: int, d_sa_slice, d_grp_slice, d_grp_prev_slice, lo: int, hi: int, d_isa, h: int, index_bytes=4, verbose=0):
        """One doubling step's sort work for this slice; returns (groups tied when it began - 0: slice final,
        work items the emit pass has to scan)."""
        o = _opts(self.device, verbose)
        t, items = C.c_int64(0), C.c_int64(0)
        _lib.check(self._L.msufsort_hip_double_sort_dev(self._h, n, self._ptr(d_sa_slice), self._ptr(d_grp_slice), self._ptr(d_grp_prev_slice), lo, hi,
                                                        self._ptr(d_isa), h, index_bytes, C.byref(o), C.byref(t), C.byref(items)), "double_sort")
        return int(t.value), int(items.value)

    def emit_updates(self, d_sa_slice, d_grp_slice, d_grp_prev_slice, lo: int, hi: int, i0: int, i1: int, items_total: int, d_updates, capacity: int,
                     index_bytes=4):
        """Rank updates of work items [i0, i1) -> d_updates; returns (count, tied_rows).  i1 == items_total closes the step."""
        cnt, tied = C.c_int64(0), C.c_int64(0)
        _lib.check(self._L.msufsort_hip_emit_updates_dev(self._h, self._ptr(d_sa_slice), self._ptr(d_grp_slice), self._ptr(d_grp_prev_slice), lo, hi, i0, i1,
                                                         items_total, self._ptr(d_updates), capacity, index_bytes, C.byref(cnt), C.byref(tied)), "emit_updates")
        return int(cnt.value), int(tied.value)

    def apply_updates(self, d_updates, count: int, d_isa, index_bytes=4):
        _lib.check(self._L.msufsort_hip_apply_updates_dev(self._h, self._ptr(d_updates), count, self._ptr(d_isa), index_bytes), "apply_updates")

    def bwt_from_sa(self, d_text, n: int, d_sa, d_bwt, index_bytes=4) -> int:
        s = C.c_int64(0)
        f = self._L.msufsort_hip_bwt_from_sa_i64_dev if index_bytes == 8 else self._L.msufsort_hip_bwt_from_sa_dev
        _lib.check(f(self._h, self._ptr(d_text), n, self._ptr(d_sa), self._ptr(d_bwt), C.byref(s)), "bwt_from_sa")
        return int(s.value)

    def bwt_slice(self, d_text, n: int, d_sa_slice, lo: int, hi: int, d_row_bytes, index_bytes=4) -> int:
        """One byte per row of the finished slice rows [lo, hi): the byte in front of the suffix; returns the sentinel row if it
        lies in the slice, else -1 (msufsort_hip_bwt_slice_dev; the sharded forward transform of msufsort_amd/dist.py)."""
        s = C.c_int64(0)
        _lib.check(self._L.msufsort_hip_bwt_slice_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa_slice), lo, hi, index_bytes, self._ptr(d_row_bytes), C.byref(s)), "bwt_slice")
        return int(s.value)

    def forward_bwt(self, d_text, n: int, d_bwt, *, two_stage=0) -> int:
        s = C.c_int64(0)
        o = _opts(self.device, two_stage=two_stage)
        _lib.check(self._L.msufsort_hip_forward_bwt_dev(self._h, self._ptr(d_text), n, self._ptr(d_bwt), C.byref(s), C.byref(o)), "forward_bwt_dev")
        return int(s.value)

    def inverse_bwt(self, d_bwt, n: int, sentinel: int, d_out):
        o = _opts(self.device)
        _lib.check(self._L.msufsort_hip_inverse_bwt_dev(self._h, self._ptr(d_bwt), n, sentinel, self._ptr(d_out), C.byref(o)), "inverse_bwt_dev")

    def lcp(self, d_text, n: int, d_sa, d_lcp):
        _lib.check(self._L.msufsort_hip_lcp_i32_dev(self._h, self._ptr(d_text), n, self._ptr(d_sa), self._ptr(d_lcp)), "lcp_dev")

    def validate_sa(self, d_text, n: int, d_sa, index_bytes=4) -> int:
        e = C.c_int64(0)
        f = self._L.msufsort_hip_validate_sa_i64_dev if index_bytes == 8 else self._L.msufsort_hip_validate_sa_dev
        _lib.check(f(self._h, self._ptr(d_text), n, self._ptr(d_sa), C.byref(e)), "validate")
        return int(e.value)

    def debug_hist16(self, d_text, n: int, d_hist):
        _lib.check(self._L.msufsort_hip_debug_hist16_dev(self._h, self._ptr(d_text), n, self._ptr(d_hist)), "hist16")
