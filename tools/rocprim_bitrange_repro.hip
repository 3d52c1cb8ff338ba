// Stand-alone sweep that found the rocPRIM 4.2 (ROCm 7.2) issue worked around in engine.hip:
// radix_sort_pairs on 64-bit keys with a bit range [b > 0, 64) returns mis-ordered keys for inputs
// below ~1-2 M items on gfx950; ranges that end below bit 64 and the full range [0, 64) are fine.
// build: hipcc --offload-arch=gfx950 -O2 -o repro tools/rocprim_bitrange_repro.hip ; prints the failing cases.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
int main() {
  for (size_t n : {5000ul, 271428ul, 900000ul, 2000000ul}) {
   for (int b1 : {40, 48, 55, 56, 60, 63, 64}) for (int b0 : {0, 1, 8, 16, 23, 24, 30, 31, 32, 33, 40}) { if (b0 >= b1) continue;
    
    std::mt19937_64 rng(1);
    std::vector<uint64_t> k(n); std::vector<uint32_t> v(n);
    for (size_t i = 0; i < n; ++i) { k[i] = rng(); if (b1 < 64) k[i] &= ((1ull << b1) - 1); v[i] = (uint32_t)i; }
    // duplicates
    for (size_t i = 0; i + 7 < n; i += 7) k[i + 3] = k[i];
    uint64_t *dk, *dko; uint32_t *dv, *dvo;
    hipMalloc(&dk, n * 8); hipMalloc(&dko, n * 8); hipMalloc(&dv, n * 4); hipMalloc(&dvo, n * 4);
    hipMemcpy(dk, k.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    size_t tb = 0; void* tmp = nullptr;
    rocprim::radix_sort_pairs(nullptr, tb, dk, dko, dv, dvo, n, b0, b1, 0);
    hipMalloc(&tmp, tb);
    rocprim::radix_sort_pairs(tmp, tb, dk, dko, dv, dvo, n, b0, b1, 0);
    hipDeviceSynchronize();
    std::vector<uint64_t> o(n); hipMemcpy(o.data(), dko, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 1; i < n; ++i) { uint64_t a = (o[i-1] >> b0), b = (o[i] >> b0); if (b1 < 64) { a &= (1ull << (b1-b0)) - 1; b &= (1ull << (b1-b0)) - 1; } if (a > b) ++bad; }
    if (bad) printf("n=%zu bits[%d,%d) inversions=%zu\n", n, b0, b1, bad);
    hipFree(dk); hipFree(dko); hipFree(dv); hipFree(dvo); hipFree(tmp);
   }
  }
  printf("sweep done\n");
  return 0;
}
