// kspider_amd engine — hand-written HIP for gfx950 (MI355X, CDNA4).
//
// What it replaces: the accumulate region of kSpider::pairwise()
// (/root/reference/src/pairwise.cpp:194-237): for every colour/k-mer, all C(m,2)
// source pairs are pushed through a 4096-shard mutex-protected hash map
// (PAIRS_COUNTER, :22-27).  Here the per-source hash sets live in HBM as sorted
// uint64 runs and the N x N shared-k-mer matrix is produced tile by tile:
//
//   stage 1 (build_blocks)  the sorted runs of each block of TB = 128 sources are
//            merged into ONE sorted list of distinct keys with postings (which of
//            the 128 sources hold the key) — so a key is compared once per block
//            pair instead of once per source pair (128x fewer comparisons).
//   stage 2 (k_join)        one workgroup per block pair (I, J): a 128 x 128 tile of
//            uint32 pair counters lives in LDS (64 KB); the two block lists stream
//            through the CU once, coalesced; each wave merge-intersects 64-key
//            chunks (A chunk one key per lane in VGPRs, B chunk broadcast from LDS,
//            64 x 64 compares, ballot/popcount bookkeeping); matches add the key's
//            weight to S[i][j] with LDS atomics; finally non-zero counters are
//            compacted (ballot + popcount prefix + one global atomic per wave) into
//            (source_1, source_2, shared) edges.
//
// Integer set intersection: no MFMA.  The dominant kernel k_join is bound by the
// HBM/L2 stream of the block lists (12 B per key), see DESIGN.md.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/kspider_amd.h"
#include "engine_internal.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

namespace ksp {

thread_local std::string g_error;
void set_error(const std::string& s) { g_error = s; }

#define KSP_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t err__ = (call);                                                                 \
        if (err__ != hipSuccess) {                                                                 \
            ksp::set_error(std::string(#call) + ": " + hipGetErrorString(err__));                  \
            return KSP_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

constexpr int TB = 128;       // sources per block (tile edge)
constexpr int NP = 64;        // value-range parts per block (intra-tile work items)
constexpr int JW = 8;         // waves per join workgroup
constexpr int CH = 64;        // keys per chunk = wavefront width
constexpr u32 MULTI = 0x80000000u;

// ------------------------------------------------------------------------------------
// stage 1 kernels
// ------------------------------------------------------------------------------------

// One workgroup per source: tag each entry with (block << 8 | local id) [and weight].
template <bool W>
__global__ void k_tag(const u64* __restrict__ off, const u32* __restrict__ wts, u32* __restrict__ val32,
                      u64* __restrict__ val64) {
    const u32 s = blockIdx.x;
    const u64 b = off[s], e = off[s + 1];
    const u32 tag = ((s / TB) << 8) | (s % TB);
    for (u64 i = b + threadIdx.x; i < e; i += blockDim.x) {
        if (W) val64[i] = ((u64)wts[i] << 32) | tag;
        else val32[i] = tag;
    }
}

template <class V> __device__ inline u32 tag_of(V v) { return (u32)v; }

// flag[e] = 1 when entry e opens a new (block, key) group.
template <class V>
__global__ void k_heads(const u64* __restrict__ keys, const V* __restrict__ vals, u32* __restrict__ flag, u64 n) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    u32 f = 1;
    if (e > 0) f = ((tag_of(vals[e]) >> 8) != (tag_of(vals[e - 1]) >> 8)) || (keys[e] != keys[e - 1]);
    flag[e] = f;
}

// distinct keys + first entry of each group; estart[Ktot] = n.
__global__ void k_emit_keys(const u64* __restrict__ keys, const u32* __restrict__ flag, const u32* __restrict__ didx,
                            u64* __restrict__ bkeys, u32* __restrict__ estart, u32* __restrict__ ktot, u64 n) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (flag[e]) {
        bkeys[didx[e]] = keys[e];
        estart[didx[e]] = (u32)e;
    }
    if (e == n - 1) {
        u32 k = didx[e] + flag[e];
        estart[k] = (u32)n;
        *ktot = k;
    }
}

// first distinct-key index of each block (entries of a block are contiguous).
__global__ void k_blk_off(const u64* __restrict__ off, const u32* __restrict__ didx, const u32* __restrict__ ktot,
                          u32* __restrict__ blk_off, u32 nb, u32 n_sources, u64 n) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    u64 s = (u64)b * TB;
    u64 es = off[s < n_sources ? s : n_sources];
    blk_off[b] = (es < n) ? didx[es] : *ktot;
}

__global__ void k_mmsize(const u32* __restrict__ estart, const u32* __restrict__ ktot, u32* __restrict__ mmsz,
                         u64 cap) {
    u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= cap) return;
    u32 v = 0;
    if (d < *ktot) {
        u32 c = estart[d + 1] - estart[d];
        v = c >= 2 ? c + 1 : 0;
    }
    mmsz[d] = v;
}

// info word per distinct key: singleton -> local id; otherwise MULTI | offset into mm,
// where mm[off] = count-1 and mm[off+1 ..] = the local ids in ascending order.
template <class V, bool W>
__global__ void k_emit_info(const u32* __restrict__ estart, const u32* __restrict__ mmoff,
                            const u32* __restrict__ ktot, const V* __restrict__ vals, u32* __restrict__ info,
                            u8* __restrict__ mm, u32* __restrict__ bw) {
    u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= *ktot) return;
    u32 b = estart[d], c = estart[d + 1] - b;
    V v0 = vals[b];
    if (W) bw[d] = (u32)((u64)v0 >> 32);
    if (c == 1) {
        info[d] = tag_of(v0) & 0xFF;
    } else {
        u32 o = mmoff[d];
        info[d] = MULTI | o;
        mm[o] = (u8)(c - 1);
        for (u32 i = 0; i < c; ++i) mm[o + 1 + i] = (u8)(tag_of(vals[b + i]) & 0xFF);
    }
}

// part[b][p] = first key of block b that is >= p * step  (p = 0..NP; part[b][NP] = end).
__global__ void k_part(const u64* __restrict__ bkeys, const u32* __restrict__ blk_off, u32* __restrict__ part,
                       u32 nb, u64 step) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * (NP + 1)) return;
    u32 b = i / (NP + 1), p = i % (NP + 1);
    u32 lo = blk_off[b], hi = blk_off[b + 1];
    if (p == NP) { part[i] = hi; return; }
    u64 v = (u64)p * step;   // NP * step > max key, p * step never overflows
    while (lo < hi) {
        u32 mid = lo + ((hi - lo) >> 1);
        if (bkeys[mid] < v) lo = mid + 1; else hi = mid;
    }
    part[i] = lo;
}

// ------------------------------------------------------------------------------------
// stage 2: the join kernel
// ------------------------------------------------------------------------------------
struct JoinArgs {
    const u64* bkeys;
    const u32* info;
    const u32* bw;      // NULL -> weight 1
    const u8* mm;
    const u32* blk_off; // nb + 1
    const u32* part;    // nb * (NP + 1)
    u32 nb;
    u32 n_sources;
    u64 tile_begin;
    ksp_edge* out;
    u64 cap;
    unsigned long long* out_count;
};

__host__ __device__ inline u64 tile_row_start(u64 r, u64 nb) { return r * nb - r * (r - 1) / 2; }

__host__ __device__ inline void tile_decode(u64 t, u32 nb, u32& I, u32& J) {
    double b = 2.0 * (double)nb + 1.0;
    double disc = b * b - 8.0 * (double)t;
    long long i = (long long)floor((b - sqrt(disc > 0 ? disc : 0.0)) * 0.5);
    if (i < 0) i = 0;
    if (i >= (long long)nb) i = (long long)nb - 1;
    while (i > 0 && tile_row_start((u64)i, nb) > t) --i;
    while (i + 1 < (long long)nb && tile_row_start((u64)i + 1, nb) <= t) ++i;
    I = (u32)i;
    J = (u32)(i + (long long)(t - tile_row_start((u64)i, nb)));
}

// members of a posting: singleton id or list in mm.
struct Members {
    const u8* p;
    u32 n;
    u32 single;
    __device__ inline u32 get(u32 i) const { return p ? (u32)p[i] : single; }
};
__device__ inline Members members_of(u32 info, const u8* __restrict__ mm) {
    Members m;
    if (info & MULTI) {
        const u8* q = mm + (info & ~MULTI);
        m.n = (u32)q[0] + 1;
        m.p = q + 1;
        m.single = 0;
    } else {
        m.n = 1;
        m.p = nullptr;
        m.single = info;
    }
    return m;
}

template <bool W>
__global__ __launch_bounds__(JW * 64) void k_join(JoinArgs a) {
    __shared__ u32 S[TB * TB];
    __shared__ u64 sBk[JW][CH];
    __shared__ u32 sBi[JW][CH];
    __shared__ int s_next;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u32 I, J;
    tile_decode(a.tile_begin + blockIdx.x, a.nb, I, J);

    for (int i = tid; i < TB * TB; i += JW * 64) S[i] = 0;
    if (tid == 0) s_next = 0;
    __syncthreads();

    if (I == J) {
        // self tile: every key of the block matches itself; only multi-member keys
        // produce pairs (i < j because postings are ascending).
        const u32 kb = a.blk_off[I], ke = a.blk_off[I + 1];
        for (u32 k = kb + tid; k < ke; k += JW * 64) {
            u32 inf = a.info[k];
            if (inf & MULTI) {
                const u32 w = W ? a.bw[k] : 1u;
                Members m = members_of(inf, a.mm);
                for (u32 x = 0; x + 1 < m.n; ++x) {
                    u32 mx = m.p[x];
                    for (u32 y = x + 1; y < m.n; ++y) atomicAdd(&S[mx * TB + m.p[y]], w);
                }
            }
        }
    } else {
        const u32* partI = a.part + (size_t)I * (NP + 1);
        const u32* partJ = a.part + (size_t)J * (NP + 1);
        u64* myBk = sBk[wv];
        u32* myBi = sBi[wv];
        while (true) {
            int p = 0;
            if (lane == 0) p = atomicAdd(&s_next, 1);
            p = __builtin_amdgcn_readfirstlane(p);
            if (p >= NP) break;
            u32 pa = partI[p], ea = partI[p + 1], pb = partJ[p], eb = partJ[p + 1];
            if (pa >= ea || pb >= eb) continue;

            // current chunks (A: one key per lane; B: staged in LDS, padded with its last key)
            u32 na = min((u32)CH, ea - pa), nbk = min((u32)CH, eb - pb);
            u64 ka = 0; u32 ia = 0, wa = 1;
            if (lane < (int)na) { ka = a.bkeys[pa + lane]; ia = a.info[pa + lane]; if (W) wa = a.bw[pa + lane]; }
            {
                u32 l = min((u32)lane, nbk - 1);
                myBk[lane] = a.bkeys[pb + l];
                myBi[lane] = a.info[pb + l];
            }
            // prefetched next chunks
            u64 ka_n = 0, kb_n = 0; u32 ia_n = 0, wa_n = 1, ib_n = 0;
            u32 na_n = 0, nb_n = 0;
            {
                u32 qa = pa + na;
                na_n = qa < ea ? min((u32)CH, ea - qa) : 0;
                if (lane < (int)na_n) { ka_n = a.bkeys[qa + lane]; ia_n = a.info[qa + lane]; if (W) wa_n = a.bw[qa + lane]; }
                u32 qb = pb + nbk;
                nb_n = qb < eb ? min((u32)CH, eb - qb) : 0;
                if (nb_n) { u32 l = min((u32)lane, nb_n - 1); kb_n = a.bkeys[qb + l]; ib_n = a.info[qb + l]; }
            }
            while (true) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // 64 x 64 compare: every lane's A key against the broadcast B chunk
                int hit = -1;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    u64 b = myBk[k];
                    if (ka == b) hit = k;
                }
                if (lane < (int)na && hit >= 0) {
                    hit = min(hit, (int)nbk - 1);   // padded tail duplicates the last key
                    u32 ib = myBi[hit];
                    if (!((ia | ib) & MULTI)) {
                        atomicAdd(&S[ia * TB + ib], wa);
                    } else {
                        Members mA = members_of(ia, a.mm), mB = members_of(ib, a.mm);
                        for (u32 x = 0; x < mA.n; ++x) {
                            u32 row = mA.get(x) * TB;
                            for (u32 y = 0; y < mB.n; ++y) atomicAdd(&S[row + mB.get(y)], wa);
                        }
                    }
                }
                // advance the chunk(s) with the smaller last key
                u64 aLast = __shfl(ka, (int)na - 1);
                u64 bLast = myBk[nbk - 1];
                bool advA = aLast <= bLast, advB = bLast <= aLast;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (advA) {
                    pa += na;
                    if (pa >= ea) break;
                    na = na_n; ka = ka_n; ia = ia_n; wa = wa_n;
                    u32 qa = pa + na;
                    na_n = qa < ea ? min((u32)CH, ea - qa) : 0;
                    ka_n = 0; ia_n = 0; wa_n = 1;
                    if (lane < (int)na_n) { ka_n = a.bkeys[qa + lane]; ia_n = a.info[qa + lane]; if (W) wa_n = a.bw[qa + lane]; }
                }
                if (advB) {
                    pb += nbk;
                    if (pb >= eb) break;
                    nbk = nb_n;
                    myBk[lane] = kb_n;
                    myBi[lane] = ib_n;
                    u32 qb = pb + nbk;
                    nb_n = qb < eb ? min((u32)CH, eb - qb) : 0;
                    if (nb_n) { u32 l = min((u32)lane, nb_n - 1); kb_n = a.bkeys[qb + l]; ib_n = a.info[qb + l]; }
                }
            }
        }
    }
    __syncthreads();

    // flush: compact the non-zero counters of the tile into edges
    const u32 gi0 = I * TB, gj0 = J * TB;
    for (int base = 0; base < TB * TB; base += JW * 64) {
        int idx = base + tid;
        u32 v = S[idx];
        bool nz = v != 0;
        unsigned long long mask = __ballot(nz);
        if (mask == 0) continue;
        unsigned long long wbase = 0;
        if (lane == 0) wbase = atomicAdd(a.out_count, (unsigned long long)__popcll(mask));
        wbase = __shfl(wbase, 0);
        if (nz) {
            u64 pos = wbase + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < a.cap) {
                ksp_edge e;
                e.source_1 = gi0 + (u32)(idx / TB);
                e.source_2 = gj0 + (u32)(idx % TB);
                e.shared = v;
                a.out[pos] = e;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return KSP_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 8 + 256;
        KSP_HIP(hipMalloc(&p, want));
        bytes = want;
        return KSP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

}  // namespace ksp

struct ksp_engine {
    int device = 0;
    // inputs / geometry
    u32 n_sources = 0, nb = 0;
    u64 n_entries = 0;
    bool weighted = false;
    bool built = false;
    int key_bits = 64;
    std::vector<u64> h_off;
    std::vector<u32> h_blk_off;   // distinct-key offsets of the block lists (host copy)
    // workspace
    ksp::Buf d_off, KA, KB, VA, VB, tmp, bkeys, info, bw, mm, blk_off, part, scalars, count;
    unsigned long long* h_count = nullptr;   // pinned
    u64* h_scal = nullptr;                   // pinned: [0] max key, [1] ktot
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    ksp_stats st{};
};

namespace ksp {

static inline unsigned grid_for(u64 n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

template <bool W>
static int build_impl(ksp_engine* e, const u64* d_keys, const u32* d_w, hipStream_t st) {
    typedef typename std::conditional<W, u64, u32>::type V;
    const u64 n = e->n_entries;
    const u32 N = e->n_sources, nb = e->nb;
    int rc;
    if ((rc = e->KA.ensure((n + 4) * 8))) return rc;
    if ((rc = e->KB.ensure((n + 4) * 8))) return rc;
    if ((rc = e->VA.ensure((n + 4) * sizeof(V)))) return rc;
    if ((rc = e->VB.ensure((n + 4) * sizeof(V)))) return rc;
    if ((rc = e->bkeys.ensure((n + 4) * 8))) return rc;
    if ((rc = e->info.ensure((n + 4) * 4))) return rc;
    if (W && (rc = e->bw.ensure((n + 4) * 4))) return rc;
    if ((rc = e->mm.ensure(n + n / 2 + 64))) return rc;
    if ((rc = e->blk_off.ensure(((size_t)nb + 2) * 4))) return rc;
    if ((rc = e->part.ensure(((size_t)nb + 1) * (NP + 1) * 4))) return rc;
    if ((rc = e->scalars.ensure(64))) return rc;

    u64* KA = e->KA.as<u64>();
    u64* KB = e->KB.as<u64>();
    V* VA = e->VA.as<V>();
    V* VB = e->VB.as<V>();
    u64* d_off = e->d_off.as<u64>();
    u64* d_max = e->scalars.as<u64>();
    u32* d_ktot = (u32*)(e->scalars.as<u64>() + 1);

    // key range (one 8-byte D2H, unless the caller passed key_bits)
    size_t tb = 0;
    if (e->key_bits <= 0) {
        KSP_HIP(rocprim::reduce(nullptr, tb, d_keys, d_max, (u64)0, n, rocprim::maximum<u64>(), st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::reduce(e->tmp.p, tb, d_keys, d_max, (u64)0, n, rocprim::maximum<u64>(), st));
        KSP_HIP(hipMemcpyAsync(e->h_scal, d_max, 8, hipMemcpyDeviceToHost, st));
        KSP_HIP(hipStreamSynchronize(st));
        u64 mx = e->h_scal[0];
        int bits = 1;
        while (bits < 64 && (mx >> bits)) ++bits;
        e->key_bits = bits;
    }
    const int kbits = e->key_bits;
    int bbits = 1;
    while ((1u << bbits) < nb) ++bbits;

    hipLaunchKernelGGL((k_tag<W>), dim3(N), dim3(256), 0, st, d_off, d_w, W ? nullptr : (u32*)VA,
                       W ? (u64*)VA : nullptr);
    // sort 1: by key (payload = tag [+weight])
    tb = 0;
    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, d_keys, KA, VA, VB, n, 0, kbits, st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, d_keys, KA, VA, VB, n, 0, kbits, st));
    // sort 2: stable by block id (bits [8, 8+bbits) of the tag), payload = key
    tb = 0;
    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, VB, VA, KA, KB, n, 8, 8 + bbits, st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, VB, VA, KA, KB, n, 8, 8 + bbits, st));
    // now: KB = keys sorted by (block, key); VA = tags in the same order.  KA, VB are free.
    u32* flag = (u32*)VB;
    u32* didx = (u32*)KA;                 // n u32
    u32* estart = (u32*)KA + (n + 2);     // up to n+1 u32
    const unsigned bs = 256;
    hipLaunchKernelGGL((k_heads<V>), dim3(grid_for(n, bs)), dim3(bs), 0, st, KB, VA, flag, n);
    tb = 0;
    KSP_HIP(rocprim::exclusive_scan(nullptr, tb, flag, didx, (u32)0, n, rocprim::plus<u32>(), st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb, flag, didx, (u32)0, n, rocprim::plus<u32>(), st));
    hipLaunchKernelGGL(k_emit_keys, dim3(grid_for(n, bs)), dim3(bs), 0, st, KB, flag, didx, e->bkeys.as<u64>(),
                       estart, d_ktot, n);
    hipLaunchKernelGGL(k_blk_off, dim3(grid_for((u64)nb + 1, bs)), dim3(bs), 0, st, d_off, didx, d_ktot,
                       e->blk_off.as<u32>(), nb, N, n);
    u32* mmsz = flag;     // flags are dead now
    u32* mmoff = didx;    // didx is dead after k_blk_off
    hipLaunchKernelGGL(k_mmsize, dim3(grid_for(n, bs)), dim3(bs), 0, st, estart, d_ktot, mmsz, n);
    tb = 0;
    KSP_HIP(rocprim::exclusive_scan(nullptr, tb, mmsz, mmoff, (u32)0, n, rocprim::plus<u32>(), st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb, mmsz, mmoff, (u32)0, n, rocprim::plus<u32>(), st));
    hipLaunchKernelGGL((k_emit_info<V, W>), dim3(grid_for(n, bs)), dim3(bs), 0, st, estart, mmoff, d_ktot, VA,
                       e->info.as<u32>(), e->mm.as<u8>(), W ? e->bw.as<u32>() : nullptr);
    // value-range partition of every block list
    u64 step = (kbits >= 64 ? (~0ull >> 6) : (((1ull << kbits) - 1) >> 6)) + 1;   // NP = 64 = 2^6
    hipLaunchKernelGGL(k_part, dim3(grid_for((u64)nb * (NP + 1), bs)), dim3(bs), 0, st, e->bkeys.as<u64>(),
                       e->blk_off.as<u32>(), e->part.as<u32>(), nb, step);
    KSP_HIP(hipGetLastError());
    return KSP_OK;
}

}  // namespace ksp

using namespace ksp;

extern "C" {

const char* ksp_last_error(void) { return ksp::g_error.c_str(); }

int ksp_device_count(int* count) {
    if (!count) return KSP_E_ARG;
    KSP_HIP(hipGetDeviceCount(count));
    return KSP_OK;
}

int ksp_engine_create(int device, ksp_engine** out) {
    if (!out) { set_error("ksp_engine_create: out is NULL"); return KSP_E_ARG; }
    int n = 0;
    KSP_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { set_error("ksp_engine_create: no such device"); return KSP_E_HIP; }
    KSP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    KSP_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error(std::string("kspider_amd is built for gfx950 only; device is ") + prop.gcnArchName);
        return KSP_E_HIP;
    }
    ksp_engine* e = new ksp_engine();
    e->device = device;
    KSP_HIP(hipHostMalloc((void**)&e->h_count, 64));
    KSP_HIP(hipHostMalloc((void**)&e->h_scal, 64));
    for (int i = 0; i < 4; ++i) KSP_HIP(hipEventCreate(&e->ev[i]));
    *out = e;
    return KSP_OK;
}

void ksp_engine_destroy(ksp_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    ksp::Buf* bufs[] = {&e->d_off, &e->KA, &e->KB, &e->VA, &e->VB, &e->tmp, &e->bkeys, &e->info,
                        &e->bw, &e->mm, &e->blk_off, &e->part, &e->scalars, &e->count};
    for (auto* b : bufs) b->release();
    if (e->h_count) (void)hipHostFree(e->h_count);
    if (e->h_scal) (void)hipHostFree(e->h_scal);
    for (int i = 0; i < 4; ++i) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    delete e;
}

int ksp_engine_build_blocks(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights,
                            const uint64_t* h_offsets, uint32_t n_sources, int key_bits, void* stream) {
    if (!e || !h_offsets) { set_error("build_blocks: NULL argument"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    e->built = false;
    for (u32 s = 0; s < n_sources; ++s)
        if (h_offsets[s + 1] < h_offsets[s]) { set_error("build_blocks: offsets not monotone"); return KSP_E_ARG; }
    const u64 n = n_sources ? h_offsets[n_sources] - h_offsets[0] : 0;
    if (n_sources && h_offsets[0] != 0) { set_error("build_blocks: offsets[0] must be 0"); return KSP_E_ARG; }
    if (n >= (1ull << 30)) { set_error("build_blocks: more than 2^30 key entries per call"); return KSP_E_LIMIT; }
    if (n && !d_keys) { set_error("build_blocks: d_keys is NULL"); return KSP_E_ARG; }
    e->n_sources = n_sources;
    e->n_entries = n;
    e->nb = (n_sources + TB - 1) / TB;
    e->weighted = d_weights != nullptr;
    e->key_bits = key_bits;
    e->h_off.assign(h_offsets, h_offsets + n_sources + 1);
    e->st = ksp_stats{};
    e->st.n_sources = n_sources;
    e->st.n_entries = n;
    e->st.n_blocks = e->nb;
    e->st.n_tiles = (u64)e->nb * (e->nb + 1) / 2;
    e->st.weighted = e->weighted;
    if (n == 0 || e->nb == 0) {   // nothing can intersect
        e->built = true;
        e->st.key_bits = 0;
        return KSP_OK;
    }
    int rc;
    if ((rc = e->d_off.ensure(((size_t)n_sources + 1) * 8))) return rc;
    KSP_HIP(hipEventRecord(e->ev[0], st));
    KSP_HIP(hipMemcpyAsync(e->d_off.p, h_offsets, ((size_t)n_sources + 1) * 8, hipMemcpyHostToDevice, st));
    rc = e->weighted ? build_impl<true>(e, d_keys, d_weights, st) : build_impl<false>(e, d_keys, d_weights, st);
    if (rc) return rc;
    KSP_HIP(hipEventRecord(e->ev[1], st));
    KSP_HIP(hipMemcpyAsync(e->h_scal + 1, e->scalars.as<u64>() + 1, 8, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipStreamSynchronize(st));
    KSP_HIP(hipEventElapsedTime(&e->st.ms_build, e->ev[0], e->ev[1]));
    e->st.n_block_keys = (u32)e->h_scal[1];
    e->h_blk_off.resize((size_t)e->nb + 1);
    KSP_HIP(hipMemcpy(e->h_blk_off.data(), e->blk_off.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToHost));
    e->st.key_bits = e->key_bits;
    e->built = true;
    return KSP_OK;
}

uint64_t ksp_engine_num_tiles(const ksp_engine* e) { return e ? (u64)e->nb * (e->nb + 1) / 2 : 0; }

uint64_t ksp_engine_tile_pairs(const ksp_engine* e, uint64_t t0, uint64_t t1) {
    if (!e || !e->nb) return 0;
    u64 T = ksp_engine_num_tiles(e);
    if (t1 > T) t1 = T;
    u64 pairs = 0;
    const u64 last = e->n_sources - (u64)(e->nb - 1) * TB;   // sources in the last block
    u64 t = t0;
    while (t < t1) {
        u32 I, J;
        tile_decode(t, e->nb, I, J);
        // rest of row I inside [t, t1)
        u64 row_end = tile_row_start((u64)I + 1, e->nb);
        u64 stop = std::min(row_end, t1);
        u64 nI = (I == e->nb - 1) ? last : TB;
        for (u64 tt = t; tt < stop; ++tt) {
            u32 JJ = J + (u32)(tt - t);
            u64 nJ = (JJ == e->nb - 1) ? last : TB;
            pairs += (JJ == I) ? nI * (nI - 1) / 2 : nI * nJ;
        }
        t = stop;
    }
    return pairs;
}

int ksp_engine_join(ksp_engine* e, uint64_t tile_begin, uint64_t tile_end, ksp_edge* d_edges, uint64_t capacity,
                    uint64_t* h_count, void* stream) {
    if (!e || !h_count) { set_error("join: NULL argument"); return KSP_E_ARG; }
    if (!e->built) { set_error("join: build_blocks has not been run"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    const u64 T = ksp_engine_num_tiles(e);
    if (tile_end > T) tile_end = T;
    *h_count = 0;
    e->st.last_tiles = 0; e->st.last_pairs = 0; e->st.last_edges = 0; e->st.last_stream_bytes = 0; e->st.ms_join = 0;
    if (tile_begin >= tile_end || e->n_entries == 0) return KSP_OK;
    if (capacity && !d_edges) { set_error("join: d_edges is NULL"); return KSP_E_ARG; }
    if (tile_end - tile_begin > 0x7FFFFFFFull) { set_error("join: more than 2^31 tiles in one launch"); return KSP_E_LIMIT; }
    int rc;
    if ((rc = e->count.ensure(64))) return rc;
    JoinArgs a;
    a.bkeys = e->bkeys.as<u64>();
    a.info = e->info.as<u32>();
    a.bw = e->weighted ? e->bw.as<u32>() : nullptr;
    a.mm = e->mm.as<u8>();
    a.blk_off = e->blk_off.as<u32>();
    a.part = e->part.as<u32>();
    a.nb = e->nb;
    a.n_sources = e->n_sources;
    a.tile_begin = tile_begin;
    a.out = d_edges;
    a.cap = capacity;
    a.out_count = e->count.as<unsigned long long>();
    KSP_HIP(hipMemsetAsync(a.out_count, 0, 8, st));
    KSP_HIP(hipEventRecord(e->ev[2], st));
    dim3 grid((unsigned)(tile_end - tile_begin)), block(JW * 64);
    if (e->weighted) hipLaunchKernelGGL((k_join<true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_join<false>), grid, block, 0, st, a);
    KSP_HIP(hipGetLastError());
    KSP_HIP(hipEventRecord(e->ev[3], st));
    KSP_HIP(hipMemcpyAsync(e->h_count, a.out_count, 8, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipStreamSynchronize(st));
    KSP_HIP(hipEventElapsedTime(&e->st.ms_join, e->ev[2], e->ev[3]));
    *h_count = *e->h_count;
    e->st.last_tiles = tile_end - tile_begin;
    e->st.last_pairs = ksp_engine_tile_pairs(e, tile_begin, tile_end);
    e->st.last_edges = *h_count;
    {   // bytes the kernel streams: keys (8 B) + info (4 B) [+ weight 4 B] of both lists; self tiles read info only
        const u64 per = e->weighted ? 16 : 12;
        u64 bytes = 0;
        for (u64 t = tile_begin; t < tile_end;) {
            u32 I, J;
            tile_decode(t, e->nb, I, J);
            u64 stop = std::min(tile_row_start((u64)I + 1, e->nb), tile_end);
            u64 kI = e->h_blk_off[I + 1] - e->h_blk_off[I];
            for (u64 tt = t; tt < stop; ++tt) {
                u32 JJ = J + (u32)(tt - t);
                u64 kJ = e->h_blk_off[JJ + 1] - e->h_blk_off[JJ];
                bytes += (JJ == I) ? kI * (e->weighted ? 8 : 4) : (kI + kJ) * per;
            }
            t = stop;
        }
        e->st.last_stream_bytes = bytes;
    }
    if (*h_count > capacity) {
        set_error("join: edge buffer too small (" + std::to_string(*h_count) + " > " + std::to_string(capacity) + ")");
        return KSP_E_OVERFLOW;
    }
    return KSP_OK;
}

int ksp_engine_get_stats(const ksp_engine* e, ksp_stats* out) {
    if (!e || !out) return KSP_E_ARG;
    *out = e->st;
    return KSP_OK;
}

int ksp_engine_block_key_counts(const ksp_engine* e, uint32_t* h_blk_off /* nb+1 */) {
    if (!e || !h_blk_off) return KSP_E_ARG;
    if (!e->nb || !e->n_entries) return KSP_OK;
    KSP_HIP(hipMemcpy(h_blk_off, e->blk_off.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToHost));
    return KSP_OK;
}

int ksp_device_malloc(int device, uint64_t bytes, void** d_ptr) {
    if (!d_ptr) return KSP_E_ARG;
    KSP_HIP(hipSetDevice(device));
    KSP_HIP(hipMalloc(d_ptr, bytes ? bytes : 8));
    return KSP_OK;
}
int ksp_device_free(void* d_ptr) {
    if (d_ptr) KSP_HIP(hipFree(d_ptr));
    return KSP_OK;
}
int ksp_memcpy_h2d(void* d, const void* h, uint64_t bytes) {
    if (bytes) KSP_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return KSP_OK;
}
int ksp_memcpy_d2h(void* h, const void* d, uint64_t bytes) {
    if (bytes) KSP_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return KSP_OK;
}

void ksp_free(void* p) { std::free(p); }

int ksp_pairwise_host(const uint64_t* keys, const uint32_t* weights, const uint64_t* offsets, uint32_t n_sources,
                      int device, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats) {
    if (!offsets || !out_edges || !n_edges) { set_error("pairwise_host: NULL argument"); return KSP_E_ARG; }
    *out_edges = nullptr;
    *n_edges = 0;
    ksp_engine* e = nullptr;
    int rc = ksp_engine_create(device, &e);
    if (rc) return rc;
    const u64 n = n_sources ? offsets[n_sources] : 0;
    void *d_keys = nullptr, *d_w = nullptr, *d_edges = nullptr;
    std::vector<ksp_edge> all;
    auto cleanup = [&]() {
        if (d_keys) (void)hipFree(d_keys);
        if (d_w) (void)hipFree(d_w);
        if (d_edges) (void)hipFree(d_edges);
        ksp_engine_destroy(e);
    };
    do {
        if (n) {
            if ((rc = ksp_device_malloc(device, n * 8, &d_keys))) break;
            if ((rc = ksp_memcpy_h2d(d_keys, keys, n * 8))) break;
            if (weights) {
                if ((rc = ksp_device_malloc(device, n * 4, &d_w))) break;
                if ((rc = ksp_memcpy_h2d(d_w, weights, n * 4))) break;
            }
        }
        if ((rc = ksp_engine_build_blocks(e, (const u64*)d_keys, (const u32*)d_w, offsets, n_sources, 0, nullptr))) break;
        // batches of tile rows whose worst-case edge count fits the buffer
        const u64 T = ksp_engine_num_tiles(e);
        u64 cap = 1ull << 26;   // 64 Mi edges = 1 GiB
        if ((rc = ksp_device_malloc(device, cap * sizeof(ksp_edge), &d_edges))) break;
        u64 t = 0;
        u64 step = std::max<u64>(1, cap / ((u64)TB * TB));
        ksp_stats acc{};
        while (t < T && !rc) {
            u64 t1 = std::min(T, t + step);
            u64 cnt = 0;
            rc = ksp_engine_join(e, t, t1, (ksp_edge*)d_edges, cap, &cnt, nullptr);
            if (rc) break;
            size_t old = all.size();
            all.resize(old + cnt);
            if (cnt) rc = ksp_memcpy_d2h(all.data() + old, d_edges, cnt * sizeof(ksp_edge));
            acc.ms_join += e->st.ms_join;
            acc.last_tiles += e->st.last_tiles;
            acc.last_pairs += e->st.last_pairs;
            t = t1;
        }
        if (rc) break;
        std::sort(all.begin(), all.end(), [](const ksp_edge& x, const ksp_edge& y) {
            return x.source_1 != y.source_1 ? x.source_1 < y.source_1 : x.source_2 < y.source_2;
        });
        if (stats) {
            ksp_engine_get_stats(e, stats);
            stats->ms_join = acc.ms_join;
            stats->last_tiles = acc.last_tiles;
            stats->last_pairs = acc.last_pairs;
            stats->last_edges = all.size();
        }
        ksp_edge* out = (ksp_edge*)std::malloc(std::max<size_t>(1, all.size()) * sizeof(ksp_edge));
        if (!out) { set_error("pairwise_host: out of host memory"); rc = KSP_E_ARG; break; }
        if (!all.empty()) std::memcpy(out, all.data(), all.size() * sizeof(ksp_edge));
        *out_edges = out;
        *n_edges = all.size();
    } while (0);
    cleanup();
    return rc;
}

}  // extern "C"
