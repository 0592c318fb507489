"""ctypes bindings for the two CPU checkers.

* ``port``      - liboracle.so, our plain-C restatement (oracle/msufsort_oracle.c)
* ``reference`` - _ref/libmsufsort_ref.so, the unmodified reference compiled from
                  /root/reference by oracle/Makefile (present only where it was built)

TEST INFRASTRUCTURE ONLY: the product path never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libmsufsort_ref.so")

__all__ = ["build", "have_reference", "make_suffix_array", "forward_bwt", "reverse_bwt", "lcp",
           "validate_sa", "fnv1a64", "ref_make_suffix_array", "ref_forward_bwt", "ref_reverse_bwt",
           "ref_lcp"]

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


def build(quiet: bool = True) -> None:
    """Compile liboracle.so (always) and _ref (only where /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _load(path):
    return C.CDLL(path)


_port = None
_ref = None


def _p():
    global _port
    if _port is None:
        if not os.path.exists(_PORT):
            build()
        _port = _load(_PORT)
        _port.oracle_make_suffix_array.argtypes = [_u8p, C.c_int64, _i32p]
        _port.oracle_forward_bwt.argtypes = [_u8p, C.c_int64]
        _port.oracle_forward_bwt.restype = C.c_int32
        _port.oracle_reverse_bwt.argtypes = [_u8p, C.c_int64, C.c_int32]
        _port.oracle_lcp.argtypes = [_u8p, C.c_int64, _i32p, _i32p]
        _port.oracle_validate_sa.argtypes = [_u8p, C.c_int64, _i32p]
        _port.oracle_validate_sa.restype = C.c_int64
        _port.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_int64]
        _port.oracle_fnv1a64.restype = C.c_uint64
    return _port


def have_reference() -> bool:
    return os.path.exists(_REF)


def _r():
    global _ref
    if _ref is None:
        _ref = _load(_REF)
        _ref.ref_make_suffix_array.argtypes = [_u8p, C.c_int64, _i32p, C.c_int32]
        _ref.ref_forward_bwt.argtypes = [_u8p, C.c_int64, C.c_int32]
        _ref.ref_forward_bwt.restype = C.c_int32
        _ref.ref_reverse_bwt.argtypes = [_u8p, C.c_int64, C.c_int32, C.c_int32]
        _ref.ref_lcp.argtypes = [_u8p, C.c_int64, _i32p, _i32p, C.c_int32]
    return _ref


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                             dtype=np.uint8)
    return a


def _ptr8(a):
    return a.ctypes.data_as(_u8p)


def _ptr32(a):
    return a.ctypes.data_as(_i32p)


# ------------------------------------------------------------------ port
def make_suffix_array(text) -> np.ndarray:
    t = _u8(text)
    sa = np.empty(t.size + 1, dtype=np.int32)
    if _p().oracle_make_suffix_array(_ptr8(t), t.size, _ptr32(sa)) != 0:
        raise RuntimeError("oracle_make_suffix_array failed")
    return sa


def forward_bwt(text):
    t = _u8(text).copy()
    s = _p().oracle_forward_bwt(_ptr8(t), t.size)
    if s < 0:
        raise RuntimeError("oracle_forward_bwt failed")
    return t, int(s)


def reverse_bwt(bwt, sentinel: int) -> np.ndarray:
    t = _u8(bwt).copy()
    if _p().oracle_reverse_bwt(_ptr8(t), t.size, int(sentinel)) != 0:
        raise RuntimeError("oracle_reverse_bwt failed")
    return t


def lcp(text, sa) -> np.ndarray:
    t = _u8(text)
    sa = np.ascontiguousarray(sa, dtype=np.int32)
    out = np.zeros(t.size, dtype=np.int32)
    _p().oracle_lcp(_ptr8(t), t.size, _ptr32(sa), _ptr32(out))
    return out


def validate_sa(text, sa) -> int:
    t = _u8(text)
    sa = np.ascontiguousarray(sa, dtype=np.int32)
    assert sa.size == t.size + 1
    return int(_p().oracle_validate_sa(_ptr8(t), t.size, _ptr32(sa)))


def fnv1a64(a) -> int:
    a = np.ascontiguousarray(a)
    return int(_p().oracle_fnv1a64(a.ctypes.data, a.nbytes))


# ------------------------------------------------------------------ reference
def ref_make_suffix_array(text, threads: int = 1) -> np.ndarray:
    t = _u8(text)
    sa = np.empty(t.size + 1, dtype=np.int32)
    if _r().ref_make_suffix_array(_ptr8(t), t.size, _ptr32(sa), threads) != 0:
        raise RuntimeError("ref_make_suffix_array refused (n out of [1, 2^30-1])")
    return sa


def ref_forward_bwt(text, threads: int = 1):
    t = _u8(text).copy()
    s = _r().ref_forward_bwt(_ptr8(t), t.size, threads)
    if s < 0:
        raise RuntimeError("ref_forward_bwt refused")
    return t, int(s)


def ref_reverse_bwt(bwt, sentinel: int, threads: int = 1) -> np.ndarray:
    t = _u8(bwt).copy()
    if _r().ref_reverse_bwt(_ptr8(t), t.size, int(sentinel), threads) != 0:
        raise RuntimeError("ref_reverse_bwt refused")
    return t


def ref_lcp(text, sa, threads: int = 1) -> np.ndarray:
    t = _u8(text)
    sa = np.ascontiguousarray(sa, dtype=np.int32)
    out = np.zeros(t.size, dtype=np.int32)
    if _r().ref_lcp(_ptr8(t), t.size, _ptr32(sa), _ptr32(out), threads) != 0:
        raise RuntimeError("ref_lcp refused")
    return out
