// Driver of the sanitizer build (make asan): runs one host entry point on one input and prints its return code.
//   host_asan_check index PREFIX | info PREFIX | sigs DIR KSIZE OUTPREFIX | bins DIR OUTPREFIX
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/kspider_amd.h"

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    int rc = -1;
    if (!std::strcmp(argv[1], "index")) rc = kspider_pairwise(argv[2], 2);
    else if (!std::strcmp(argv[1], "info")) { uint64_t out[6]; rc = ksp_index_info(argv[2], out); if (!rc) std::printf("info %llu %llu %llu %llu %llu %llu\n", (unsigned long long)out[0], (unsigned long long)out[1], (unsigned long long)out[2], (unsigned long long)out[3], (unsigned long long)out[4], (unsigned long long)out[5]); }
    else if (!std::strcmp(argv[1], "sigs") && argc >= 5) rc = kspider_pairwise_sigs(argv[2], std::atoi(argv[3]), argv[4], 2);
    else if (!std::strcmp(argv[1], "bins") && argc >= 4) rc = kspider_pairwise_bins(argv[2], argv[3], 2);
    std::printf("rc %d %s\n", rc, rc ? ksp_last_error() : "");
    return 0;
}
