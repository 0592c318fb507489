// Minimal C++ consumer written against the reference's API (compare reference
// src/executable/msufsort/main.cpp:437-486): builds with either the reference or this repo's
// include/ + -lmsufsort_hip.  Usage: demo [file]   (no file: built-in sample)
#include <library/msufsort.h>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

int main(int argc, char** argv)
{
    std::vector<std::int8_t> input;
    if (argc > 1) { std::ifstream f(argv[1], std::ios::binary); input.assign(std::istreambuf_iterator<char>(f), {}); }
    else { const char* s = "mississippi"; input.assign(s, s + 11); }
    auto sa = maniscalco::make_suffix_array(input.begin(), input.end(), 4);
    std::printf("n = %zu, SA[0] = %d, SA[1] = %d\n", input.size(), sa[0], sa.size() > 1 ? sa[1] : -1);
    auto copy = input;
    auto sentinel = maniscalco::forward_burrows_wheeler_transform(copy.begin(), copy.end(), 4);
    maniscalco::reverse_burrows_wheeler_transform(copy.begin(), copy.end(), sentinel, 4);
    std::printf("sentinel row = %d, round trip %s\n", sentinel, copy == input ? "ok" : "FAILED");
    return copy == input ? 0 : 1;
}
