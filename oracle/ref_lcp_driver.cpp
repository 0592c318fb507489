// TEST INFRASTRUCTURE ONLY - never linked into the product path.
//
// The reference keeps its only LCP code inside the demo executable
// (/root/reference/src/executable/msufsort/main.cpp:16-159, anonymous
// namespace).  To run THAT code as the LCP oracle we compile the demo's source
// file where it lies (its `main` renamed) and call `lcp_multithreaded`
// (main.cpp:66-101) from the same translation unit.  Nothing is copied.
#define main msufsort_ref_demo_main
#include <executable/msufsort/main.cpp>
#undef main
#include <vector>

extern "C" {

// LCP convention of the demo: out[i] = lcp(suffix(SA[i+1]),
// suffix(SA[i+2])) for i in [0, n-2]; the demo reads one element past its buffer
// for entry n-1 (main.cpp:85), so that entry is undefined there: we give the demo
// a padded buffer and define out[n-1] = 0.
int ref_lcp(const uint8_t* text, int64_t n, const int32_t* sa /* n+1 */, int32_t* out /* n */, int32_t threads)
{
    if (n <= 0 || threads <= 0) return -1;
    std::vector<int32_t> work(sa + 1, sa + n + 1);
    work.push_back(0);                               // padding for the OOB read at main.cpp:85
    lcp_multithreaded((int8_t const*)text, (int8_t const*)text + n, work.data(), (int32_t)n, threads);
    for (int64_t i = 0; i + 1 < n; ++i) out[i] = work[i];
    out[n - 1] = 0;
    return 0;
}

}
