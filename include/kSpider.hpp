// Drop-in declaration of the one reference entry point this library implements.
//
// Reference: /root/reference/include/kSpider.hpp:11
//     namespace kSpider { void pairwise(string index_prefix, int user_threads); }
// defined in /root/reference/src/pairwise.cpp:123.  Same name, argument meaning, files
// read and written, and stdout phase lines; the accumulate region (src/pairwise.cpp:
// 194-237) runs on an MI355X through libkspider_amd.so.  Errors: the reference asserts or
// reads garbage on a missing/short file; this implementation throws std::runtime_error
// and never leaves a partial TSV behind.
//
// The reference header also pulls in argh.h, colored_kDataFrame.hpp and phmap.h
// (include/kSpider.hpp:3-7): none of them is needed by callers of pairwise() and they are
// deliberately not reproduced.  The other eleven functions of the reference header
// (index_*, *_to_kDataFrame, sourmash_sigs_indexing, bins_indexing) are outside the scope
// of this library (out of scope / "next" rows).
#ifndef KSPIDER_AMD_KSPIDER_HPP
#define KSPIDER_AMD_KSPIDER_HPP
#include <string>

namespace kSpider {

// Device = $KSPIDER_DEVICE (default 0), or the devices of $KSPIDER_DEVICES ("0,1,2,...": every device builds the
// block lists of 1 / n of the colours and joins 1 / n of the tiles; the edges meet on the first device over xGMI).
// An index of 2^30 colour memberships or more is cut into slices by itself (the reference has no size limit,
// src/pairwise.cpp:95-111).  user_threads: host threads used for formatting the TSV (the reference uses it for its
// OpenMP accumulate loop).
void pairwise(std::string index_prefix, int user_threads);

}  // namespace kSpider
#endif
