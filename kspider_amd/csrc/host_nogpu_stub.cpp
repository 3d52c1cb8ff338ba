// Sanitizer build of the HOST side only (make asan): the file parsers and TSV writers are compiled with
// -fsanitize=address,undefined and linked against this stub instead of the HIP engine, so that a CPU test can
// feed them malformed files (tests/test_host_hardening_cpu.py).  Every compute entry fails with KSP_E_HIP —
// there is no CPU implementation of the hot path, here or anywhere else in the product.
#include <cstdlib>
#include <string>

#include "../../include/kspider_amd.h"
#include "engine_internal.h"

namespace ksp {
static thread_local std::string g_error;
void set_error(const std::string& s) { g_error = s; }
}  // namespace ksp

namespace ksp {
int pairwise_postings_multi_cc(const uint64_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, const int*, int, ksp_edge**,
                               uint64_t*, ksp_stats*, CcRequest*) {
    set_error("host-only sanitizer build: no HIP engine");
    return KSP_E_HIP;
}
void cc_critical(double, float* vcrit, int* mode) { *vcrit = 0; *mode = 0; }
void read_names_map(const std::string&, std::vector<std::string>& name_of) { name_of.clear(); }
void write_cluster_file(const std::string&, double, const std::vector<uint32_t>&, const std::vector<std::string>&) {}
}  // namespace ksp

extern "C" {
const char* ksp_last_error(void) { return ksp::g_error.c_str(); }
void ksp_free(void* p) { std::free(p); }
int ksp_pairwise_host_multi(const uint64_t*, const uint32_t*, const uint64_t*, uint32_t, const int*, int, ksp_edge**,
                            uint64_t*, ksp_stats*) {
    ksp::set_error("host-only sanitizer build: no HIP engine");
    return KSP_E_HIP;
}
int ksp_pairwise_postings_host_multi(const uint64_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, const int*, int,
                                     ksp_edge**, uint64_t*, ksp_stats*) {
    ksp::set_error("host-only sanitizer build: no HIP engine");
    return KSP_E_HIP;
}
}
