// Drop-in replacement for the public surface of michaelmaniscalco/msufsort, backed by the
// MI355X engine through the C-ABI in include/msufsort_hip.h (link with -lmsufsort_hip).
//
// Mirrors, name for name and argument for argument:
//   class maniscalco::msufsort                         reference src/library/msufsort/msufsort.h:42-75
//   maniscalco::make_suffix_array<It>                  reference msufsort.h:403-445
//   maniscalco::forward_burrows_wheeler_transform<It>  reference msufsort.h:449-462
//   maniscalco::reverse_burrows_wheeler_transform<It>  reference msufsort.h:466-476
// Conventions are the reference's: SA has n+1 entries with SA[0] = n; the BWT is written in place
// and the sentinel row is returned; the inverse transforms in place.
// Differences, all deliberate: `threads` is accepted and ignored (the work runs on the GPU);
// n = 0 is defined (SA = {0}, BWT no-op, sentinel 0) where the reference is UB (cpp:1588);
// n may be up to 2^31-2 (the reference corrupts its output at n >= 2^30);
// a failure of the HIP path (no device, out of memory) throws std::runtime_error - there is
// no CPU fallback behind this header.
#pragma once

// The reference header pulls these in (msufsort.h:30-36) and its one consumer relies on that: the demo uses std::thread
// without including <thread> itself (main.cpp:76).  A drop-in has to keep the transitive includes too
// (tests/test_cabi.py::test_reference_consumer_compiles_unchanged builds that file against this header).
#include <vector>
#include <stdint.h>
#include <atomic>
#include <thread>
#include <memory>
#include <array>
#include <functional>
// what this header needs itself
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../msufsort_hip.h"

namespace maniscalco
{

    namespace msufsort_detail
    {
        // The reference value-initialises its result (`suffix_array sa(n + 1)` + a zero-fill, msufsort.cpp:1754-1758): one thread
        // writing 4 GiB of fresh memory costs 0.7 s on the GPU box's host - seven times the whole GPU build including the
        // PCIe copies.  Every entry is overwritten by the device-to-host copy, so the vector is grown WITHOUT touching its
        // storage: reserve(), then the end pointer is moved (the idea of folly's UninitializedMemoryHacks.h) - with libstdc++, the one
        // standard library this header could be compiled and tested against (the image has no libc++ headers: a branch for it was
        // written in round 4 and removed in round 6, never having been compiled); other libraries, debug and sanitizer builds fall
        // back to the value-initialising resize.
#if defined(__GLIBCXX__) && !defined(_GLIBCXX_DEBUG) && !defined(__SANITIZE_ADDRESS__)
#define MSUFSORT_UNINITIALIZED_RESIZE 1
        // libstdc++: the storage pointers live in the protected base's _M_impl; a class derived from the vector may form the
        // pointer to that member and apply it to any vector
        struct vec_access : std::vector<std::int32_t>
        {
            static void set_size(std::vector<std::int32_t> & v, std::size_t n)
            {
                auto & impl = v.*(&vec_access::_M_impl);
                impl._M_finish = impl._M_start + n;
            }
        };
        inline void grow_uninitialized(std::vector<std::int32_t> & v, std::size_t n)
        {
            v.reserve(n);
            vec_access::set_size(v, n);
        }
#else
        inline void grow_uninitialized(std::vector<std::int32_t> & v, std::size_t n) { v.resize(n); }
#endif
    }

    class msufsort
    {
    public:

        static auto constexpr max_radix_size = (1 << 16);
        using suffix_index = std::int32_t;
        using suffix_array = std::vector<suffix_index>;

        // The reference constructor spawns its worker pool (msufsort.h:315-341); this one owns a GPU context
        // (stream + workspace), created on first use and reused by every call on the instance.
        msufsort(std::int32_t numThreads = 1) : numThreads_(numThreads) { ++instances(); }
        ~msufsort()
        {
            if (ctx_) ::msufsort_hip_ctx_destroy(ctx_);
            // the streaming entry point pools its contexts (stream + workspace + a text copy per device) for the life of the
            // process; the last instance gives them back, as the reference's destructor joins its worker pool (msufsort.h:343-349)
            if (--instances() == 0 && pooled()) { ::msufsort_hip_release_cached(); pooled() = false; }
        }
        msufsort(msufsort const &) = delete;
        msufsort & operator = (msufsort const &) = delete;

        suffix_array make_suffix_array(std::uint8_t const * inputBegin, std::uint8_t const * inputEnd)
        {
            auto n = static_cast<std::int64_t>(inputEnd - inputBegin);
            suffix_array sa;
            msufsort_detail::grow_uninitialized(sa, static_cast<std::size_t>(n) + 1);       // (filled by the library; see above)
            // large inputs: the streaming entry point - finished slices leave for the host while the rest is sorted - on this
            // instance's device; several GPUs only when the caller asks for them (MSUFSORT_DEVICES="0,1,..."): a library that
            // allocates on every visible GPU of a shared node by default is a bad neighbour.  Small inputs: this instance's
            // context (lowest latency)
            if (n >= (std::int64_t(32) << 20))
            {
                std::int32_t const own = 0;
                char const * const env = std::getenv("MSUFSORT_DEVICES");
                bool const listed = env != nullptr && *env != '\0';      // (empty counts as unset: device 0)
                pooled() = true;
                check(::msufsort_hip_make_sa_multi(listed ? nullptr : &own, listed ? 0 : 1, inputBegin, n, sa.data(), 4, nullptr, nullptr), "make_suffix_array");
            }
            else
                check(::msufsort_hip_make_sa_i32_ctx(ctx(), inputBegin, n, sa.data(), nullptr), "make_suffix_array");
            return sa;
        }

        std::int32_t forward_burrows_wheeler_transform(std::uint8_t * inputBegin, std::uint8_t * inputEnd)
        {
            std::int64_t sentinel = 0;
            auto n = static_cast<std::int64_t>(inputEnd - inputBegin);
            // large inputs: the streaming entry point - the BWT bytes of finished key ranges leave for the host while the rest is
            // sorted (n bytes over PCIe, none of the rows) - on this instance's device, or on the devices MSUFSORT_DEVICES lists;
            // text-like inputs are recognised there and take one two-stage build.  Small inputs: this instance's context
            if (n >= (std::int64_t(32) << 20))
            {
                std::int32_t const own = 0;
                char const * const env = std::getenv("MSUFSORT_DEVICES");
                bool const listed = env != nullptr && *env != '\0';
                pooled() = true;
                check(::msufsort_hip_forward_bwt_multi(listed ? nullptr : &own, listed ? 0 : 1, inputBegin, n, &sentinel, nullptr, nullptr), "forward_burrows_wheeler_transform");
            }
            else
                check(::msufsort_hip_forward_bwt_ctx(ctx(), inputBegin, n, &sentinel, nullptr), "forward_burrows_wheeler_transform");
            return static_cast<std::int32_t>(sentinel);
        }

        static void reverse_burrows_wheeler_transform(std::uint8_t * inputBegin, std::uint8_t * inputEnd,
                                                      std::int32_t sentinelIndex, std::int32_t /*numThreads*/)
        {
            check(::msufsort_hip_inverse_bwt(inputBegin, static_cast<std::int64_t>(inputEnd - inputBegin), sentinelIndex, nullptr),
                  "reverse_burrows_wheeler_transform");
        }

        // Extension (the reference keeps its LCP in the demo executable, main.cpp:143-159):
        // out[i] = lcp(suffix SA[i+1], suffix SA[i+2]) for i in [0, n-2], out[n-1] = 0.
        std::vector<std::int32_t> make_lcp_array(std::uint8_t const * inputBegin, std::uint8_t const * inputEnd, suffix_array const & sa)
        {
            auto n = static_cast<std::int64_t>(inputEnd - inputBegin);
            std::vector<std::int32_t> lcp;
            msufsort_detail::grow_uninitialized(lcp, static_cast<std::size_t>(n));
            check(::msufsort_hip_lcp_i32_ctx(ctx(), inputBegin, n, sa.data(), lcp.data()), "make_lcp_array");
            return lcp;
        }

    private:

        static void check(int status, char const * what)
        {
            if (status != MSUFSORT_HIP_OK)
                throw std::runtime_error(std::string("maniscalco::msufsort::") + what + ": " +
                                         ::msufsort_hip_strerror(status) + " - " + ::msufsort_hip_last_error());
        }

        ::msufsort_hip_ctx * ctx()
        {
            if (!ctx_) check(::msufsort_hip_ctx_create(&ctx_, 0, 0), "context");
            return ctx_;
        }

        static std::atomic<int> & instances() { static std::atomic<int> n{0}; return n; }
        static std::atomic<bool> & pooled() { static std::atomic<bool> p{false}; return p; }      // (the pool only frees idle contexts)

        std::int32_t numThreads_;
        ::msufsort_hip_ctx * ctx_ = nullptr;
    };


    template <typename input_iter>
    msufsort::suffix_array make_suffix_array(input_iter begin, input_iter end, std::int32_t numThreads = 1)
    {
        return msufsort(numThreads).make_suffix_array((std::uint8_t const *)&*begin, (std::uint8_t const *)&*begin + (end - begin));
    }

    template <typename input_iter>
    std::int32_t forward_burrows_wheeler_transform(input_iter begin, input_iter end, std::int32_t numThreads = 1)
    {
        return msufsort(numThreads).forward_burrows_wheeler_transform((std::uint8_t *)&*begin, (std::uint8_t *)&*begin + (end - begin));
    }

    template <typename input_iter>
    void reverse_burrows_wheeler_transform(input_iter begin, input_iter end, std::int32_t sentinelIndex, std::int32_t numThreads = 1)
    {
        msufsort::reverse_burrows_wheeler_transform((std::uint8_t *)&*begin, (std::uint8_t *)&*begin + (end - begin), sentinelIndex, numThreads);
    }

} // namespace maniscalco
