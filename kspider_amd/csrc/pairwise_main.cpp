// `pairwise PREFIX THREADS` — same command line as the reference's exe
// (/root/reference/pairwise.cpp:3-5, CMake target `pairwise`).
#include <cstdio>
#include <exception>
#include <string>

#include "../../include/kSpider.hpp"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s INDEX_PREFIX THREADS\n", argv[0]);
        return 2;
    }
    try {
        kSpider::pairwise(argv[1], std::stoi(argv[2]));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "pairwise: %s\n", e.what());
        return 1;
    }
    return 0;
}
