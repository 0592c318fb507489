// msufsort_demo - command-line parity with the reference demo (reference
// src/executable/msufsort/main.cpp:290-502): same modes, same argument order, same checks, but the
// work runs on the MI355X through include/library/msufsort.h.
//
//   msufsort_demo [b|s|l|t] inputFile [numThreads]
//     b  forward BWT, inverse BWT, verify the round trip          (main.cpp:466-488)
//     s  suffix array + validate_suffix_array                     (main.cpp:437-453, 236-270)
//     l  suffix array + LCP array + validate_lcp                  (main.cpp:455-464, 106-159)
//     t  self test sweep: alphabet x size grid, SA and BWT        (main.cpp:389-435)  [inputFile ignored]
// numThreads is accepted for compatibility and ignored.  This file is written from the demo's
// behaviour, not from its code: the validators below are this repository's own.
#include <library/msufsort.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

namespace
{
    using clock_type = std::chrono::steady_clock;

    double ms_since(clock_type::time_point t0)
    {
        return std::chrono::duration<double, std::milli>(clock_type::now() - t0).count();
    }

    // unsigned bytes, shorter suffix first (semantics of compare, main.cpp:210-232)
    bool suffix_less(std::vector<std::uint8_t> const & t, std::int32_t a, std::int32_t b)
    {
        auto n = static_cast<std::int32_t>(t.size());
        while (a < n && b < n) { if (t[a] != t[b]) return t[a] < t[b]; ++a; ++b; }
        return a >= n && b < n;
    }

    std::int64_t validate_suffix_array(std::vector<std::uint8_t> const & t, std::vector<std::int32_t> const & sa)
    {
        std::int64_t errors = (sa.size() != t.size() + 1) || (sa[0] != static_cast<std::int32_t>(t.size()));
        std::vector<bool> seen(t.size(), false);
        for (std::size_t i = 1; i < sa.size(); ++i)
        {
            if (sa[i] < 0 || static_cast<std::size_t>(sa[i]) >= t.size() || seen[sa[i]]) { ++errors; continue; }
            seen[sa[i]] = true;
            if (i >= 2 && !suffix_less(t, sa[i - 1], sa[i])) ++errors;
        }
        return errors;
    }

    std::int64_t validate_lcp(std::vector<std::uint8_t> const & t, std::vector<std::int32_t> const & sa, std::vector<std::int32_t> const & lcp)
    {
        std::int64_t errors = 0;
        auto n = static_cast<std::int64_t>(t.size());
        for (std::int64_t i = 0; i + 1 < n; ++i)
        {
            std::int64_t a = sa[i + 1], b = sa[i + 2], m = 0;
            while (a + m < n && b + m < n && t[a + m] == t[b + m]) ++m;
            errors += (m != lcp[i]);
        }
        return errors;
    }

    std::vector<std::uint8_t> load(char const * path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) { std::cout << "failed to open file: " << path << std::endl; std::exit(1); }
        return std::vector<std::uint8_t>(std::istreambuf_iterator<char>(f), {});
    }

    // deterministic stand-in for the demo's srand/rand inputs (main.cpp:274-286)
    std::vector<std::uint8_t> sweep_input(std::uint32_t alphabet, std::uint32_t size)
    {
        std::vector<std::uint8_t> v(size);
        std::uint64_t s = 0x9E3779B97F4A7C15ull * (alphabet * 100003ull + size);
        for (auto & e : v)
        {
            s += 0x9E3779B97F4A7C15ull;
            std::uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            e = static_cast<std::uint8_t>(z % alphabet);
        }
        return v;
    }

    int self_test(bool full)
    {
        maniscalco::msufsort engine(1);
        std::int64_t cases = 0, errors = 0;
        std::uint32_t const step = full ? 1 : 7;
        for (std::uint32_t a = 1; a < 256 && !errors; a += (a < 8 ? 1 : step))
            for (std::uint32_t n = 1; n < 1024 && !errors; n += (n < 40 ? 1 : step * 5))
            {
                auto t = sweep_input(a, n);
                auto sa = engine.make_suffix_array(t.data(), t.data() + t.size());
                errors += validate_suffix_array(t, sa);
                auto copy = t;
                auto sentinel = engine.forward_burrows_wheeler_transform(copy.data(), copy.data() + copy.size());
                maniscalco::msufsort::reverse_burrows_wheeler_transform(copy.data(), copy.data() + copy.size(), sentinel, 1);
                errors += (copy != t);
                ++cases;
                if (errors) std::cout << "**** ERROR: alphabet " << a << " size " << n << std::endl;
            }
        std::cout << "self test: " << cases << " cases, " << errors << " errors" << std::endl;
        return errors ? 1 : 0;
    }
}

int main(int argc, char ** argv)
{
    if (argc < 2 || std::string(argv[1]).size() != 1 || (argc < 3 && argv[1][0] != 't' && argv[1][0] != 'T'))
    {
        std::cout << "msufsort_demo (MI355X engine) - usage: msufsort_demo [b|s|l|t] inputFile [numThreads]" << std::endl;
        return 0;
    }
    char const mode = argv[1][0];
    try
    {
        if (mode == 't' || mode == 'T') return self_test(mode == 'T');
        auto input = load(argv[2]);
        std::int32_t threads = argc > 3 ? std::atoi(argv[3]) : 1;
        maniscalco::msufsort engine(threads);
        if (mode == 's' || mode == 'l')
        {
            auto t0 = clock_type::now();
            auto sa = engine.make_suffix_array(input.data(), input.data() + input.size());
            std::cout << "suffix array completed - total elapsed time: " << static_cast<long>(ms_since(t0)) << " ms" << std::endl;
            auto e = validate_suffix_array(input, sa);
            std::cout << (e ? "suffix array error count = " + std::to_string(e) : std::string("suffix array validated")) << std::endl;
            if (mode == 'l')
            {
                t0 = clock_type::now();
                auto lcp = engine.make_lcp_array(input.data(), input.data() + input.size(), sa);
                std::cout << "lcp array completed - total elapsed time: " << static_cast<long>(ms_since(t0)) << " ms" << std::endl;
                auto le = validate_lcp(input, sa, lcp);
                std::cout << (le ? "lcp array error count = " + std::to_string(le) : std::string("lcp array validated")) << std::endl;
                e += le;
            }
            return e ? 1 : 0;
        }
        if (mode == 'b')
        {
            auto copy = input;
            auto t0 = clock_type::now();
            auto sentinel = engine.forward_burrows_wheeler_transform(copy.data(), copy.data() + copy.size());
            std::cout << "forward BWT completed - total elapsed time: " << static_cast<long>(ms_since(t0)) << " ms, sentinel index " << sentinel << std::endl;
            t0 = clock_type::now();
            maniscalco::msufsort::reverse_burrows_wheeler_transform(copy.data(), copy.data() + copy.size(), sentinel, threads);
            std::cout << "inverse BWT completed - total elapsed time: " << static_cast<long>(ms_since(t0)) << " ms" << std::endl;
            bool ok = (copy == input);
            std::cout << (ok ? "BWT validated" : "**** BWT ERROR DETECTED ****") << std::endl;
            return ok ? 0 : 1;
        }
        std::cout << "unknown mode '" << mode << "'" << std::endl;
        return 1;
    }
    catch (std::exception const & e)
    {
        std::cout << "caught exception: " << e.what() << std::endl;
        return 2;
    }
}
