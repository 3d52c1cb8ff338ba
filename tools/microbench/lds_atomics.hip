// LDS atomic throughput on gfx950: what one compute unit sustains for the instruction mix of k_bucket_group's inserts.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
// Every workgroup (512 threads) hammers a 4 096-slot table in LDS with pseudo-random slots; the grid is 3 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
constexpr int ITERS = 2048;
template <int MODE>
__global__ __launch_bounds__(512) void k(u64* out, u32 seed) {
    __shared__ u64 t64[4097];
    __shared__ u32 t32[4097];
    for (u32 i = threadIdx.x; i <= 4096; i += 512) { t64[i] = ~0ull; t32[i] = ~0u; }
    __syncthreads();
    u32 x = seed ^ (blockIdx.x * 512 + threadIdx.x) * 0x9E3779B1u;
    u64 acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        x = x * 1664525u + 1013904223u;
        const u32 h = (x >> 12) & 4095u;
        if (MODE == 0) acc += atomicCAS(&t64[h], ~0ull, (u64)x | ((u64)h << 32));        // ds_cmpst_rtn_b64
        if (MODE == 1) acc += atomicCAS(&t32[h], ~0u, x);                                  // ds_cmpst_rtn_b32
        if (MODE == 2) atomicAdd(&t32[h], 1u);                                             // ds_add_u32 (no return)
        if (MODE == 3) acc += atomicAdd(&t32[h], 1u);                                      // ds_add_rtn_u32
        if (MODE == 4) acc += t32[h];                                                      // ds_read_b32 (gather)
        if (MODE == 5) acc += t64[h];                                                      // ds_read_b64 (gather)
        if (MODE == 6) t64[h] = x;                                                         // ds_write_b64 (scatter)
        if (MODE == 7) { atomicMax(&t64[h], (u64)x); }                                     // ds_max_u64 (no return)
    }
    if (acc == 0x1234567ull) out[0] = acc;
    if (threadIdx.x == 0 && MODE == 2) out[1 + blockIdx.x] = t32[5];
}
template <int MODE>
static void run(const char* name, u64* d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 3;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, d, 1u);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, d, 2u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // per CU: 3 workgroups x 8 waves x ITERS wave-instructions
    const double wave_instr = 3.0 * 8 * ITERS;
    printf("%-28s %8.3f ms  -> %6.1f ns per wave instruction per CU (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / wave_instr, ms * 1e6 / wave_instr * 2.4);
}
int main() {
    u64* d;
    hipMalloc(&d, 8 * 4096);
    run<0>("cmpst_rtn_b64 (CAS 64)", d);
    run<1>("cmpst_rtn_b32 (CAS 32)", d);
    run<2>("add_u32 (no return)", d);
    run<3>("add_rtn_u32", d);
    run<4>("read_b32 gather", d);
    run<5>("read_b64 gather", d);
    run<6>("write_b64 scatter", d);
    run<7>("max_u64 (no return)", d);
    return 0;
}
