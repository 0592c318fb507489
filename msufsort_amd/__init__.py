"""msufsort_amd - MI355X-native suffix-array / BWT engine behind the msufsort API.

Product path = hand-written HIP (msufsort_amd/csrc) reached through the C-ABI in
include/msufsort_hip.h.  Importing the package does not load the library; the first call
does, and fails loudly if the library has not been built (no CPU fallback).
"""
from .api import (DeviceContext, MsufsortHipError, device_count, forward_burrows_wheeler_transform,  # noqa: F401
                  forward_burrows_wheeler_transform_multi,
                  make_lcp_array, make_suffix_array, make_suffix_array_i64, make_suffix_array_multi, msufsort,
                  reverse_burrows_wheeler_transform)

__all__ = ["make_suffix_array", "make_suffix_array_i64", "make_suffix_array_multi", "forward_burrows_wheeler_transform", "forward_burrows_wheeler_transform_multi", "reverse_burrows_wheeler_transform",
           "make_lcp_array", "msufsort", "DeviceContext", "device_count", "MsufsortHipError"]
