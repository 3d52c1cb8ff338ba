// Experiment (round 1): rocPRIM radix_sort_pairs<u64 keys, u16 values> on the bits the engine sorts, default
// configuration against hand-picked onesweep configurations (block size, items per thread, rank algorithm).
// Result on MI355X, 5.0e7 pairs, 32 key bits: default 1.75 ms; best custom (1024 x 8, match) 1.84 ms; "basic"
// rank algorithms 6-8 ms; more than 8 radix bits per pass do not fit the LDS.  The engine keeps the default.
// hipcc --offload-arch=gfx950 -O3 -DRS_RB=8 -DRS_BS=1024 -DRS_IPT=8 -DRS_ALG=rocprim::block_radix_rank_algorithm::match
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#ifndef RS_RB
#define RS_RB 11
#endif
#ifndef RS_BS
#define RS_BS 512
#endif
#ifndef RS_IPT
#define RS_IPT 12
#endif
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<RS_BS, RS_IPT>, RS_RB, RS_ALG>>;
int main(int argc, char** argv) {
    size_t n = 50190311;
    std::vector<uint64_t> k(n); std::vector<uint16_t> v(n);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; k[i] = x >> 9; v[i] = (uint16_t)i; }
    uint64_t *dk, *dk2; uint16_t *dv, *dv2;
    hipMalloc(&dk, n * 8); hipMalloc(&dk2, n * 8); hipMalloc(&dv, n * 2); hipMalloc(&dv2, n * 2);
    hipMemcpy(dk, k.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 2, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        size_t tb = 0; void* tmp = nullptr;
        if (variant == 0) rocprim::radix_sort_pairs(nullptr, tb, dk, dk2, dv, dv2, n, 23, 55, 0);
        else rocprim::radix_sort_pairs<cfg>(nullptr, tb, dk, dk2, dv, dv2, n, 23, 55, 0);
        hipMalloc(&tmp, tb);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(a, 0);
            hipError_t e = variant == 0 ? rocprim::radix_sort_pairs(tmp, tb, dk, dk2, dv, dv2, n, 23, 55, 0)
                                        : rocprim::radix_sort_pairs<cfg>(tmp, tb, dk, dk2, dv, dv2, n, 23, 55, 0);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); best = std::min(best, ms);
            if (e != hipSuccess) { printf("error %d\n", (int)e); return 1; }
        }
        std::vector<uint64_t> out(n); hipMemcpy(out.data(), dk2, n * 8, hipMemcpyDeviceToHost);
        bool ok = true; for (size_t i = 1; i < n; ++i) if ((out[i - 1] >> 23) > (out[i] >> 23)) { ok = false; break; }
        printf("%s: %.3f ms, temp %.1f MB, sorted %d\n", variant ? "custom" : "default", best, tb / 1e6, (int)ok);
        hipFree(tmp);
    }
    return 0;
}
