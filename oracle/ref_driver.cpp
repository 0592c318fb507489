// TEST INFRASTRUCTURE ONLY - never linked into the product path.
//
// Thin extern "C" driver around the UNMODIFIED reference library, compiled from
// the sources where they lie under /root/reference (see oracle/Makefile, target
// `_ref`).  Output goes to oracle/_ref/libmsufsort_ref.so (git-ignored).  It is
// used (1) to pin oracle/msufsort_oracle.c and the golden fixtures, (2) as the
// checker in tests, (3) as the `cpu_baseline` leg of bench.py ("kind":
// "reference").  Nothing from the reference is copied into this repository:
// this file only *calls* the public API declared in
// /root/reference/src/library/msufsort/msufsort.h:403-476.
#include <library/msufsort.h>
#include <cstdint>
#include <cstring>

extern "C" {

// maniscalco::make_suffix_array (msufsort.h:432-445). sa_out has n+1 entries.
int ref_make_suffix_array(const uint8_t* text, int64_t n, int32_t* sa_out, int32_t threads)
{
    if (n <= 0 || n > 0x3fffffff) return -1;       // reference limit: SURVEY section 0
    auto sa = maniscalco::make_suffix_array(text, text + n, threads);
    std::memcpy(sa_out, sa.data(), sizeof(int32_t) * (size_t)(n + 1));
    return 0;
}

// maniscalco::forward_burrows_wheeler_transform (msufsort.h:449-462); in place.
int32_t ref_forward_bwt(uint8_t* inout, int64_t n, int32_t threads)
{
    if (n <= 0 || n > 0x3fffffff) return -1;
    return maniscalco::forward_burrows_wheeler_transform(inout, inout + n, threads);
}

// maniscalco::reverse_burrows_wheeler_transform (msufsort.h:466-476); in place.
int ref_reverse_bwt(uint8_t* inout, int64_t n, int32_t sentinel, int32_t threads)
{
    if (n <= 0 || threads <= 0) return -1;
    maniscalco::reverse_burrows_wheeler_transform(inout, inout + n, sentinel, threads);
    return 0;
}

}
