// kspider_amd engine — hand-written HIP for gfx950 (MI355X, CDNA4).
//
// What it replaces: the accumulate region of kSpider::pairwise()
// (/root/reference/src/pairwise.cpp:194-237): for every colour/k-mer, all C(m,2)
// source pairs are pushed through a 4096-shard mutex-protected hash map
// (PAIRS_COUNTER, :22-27).  Here the per-source hash sets live in HBM as sorted
// uint64 runs and the N x N shared-k-mer matrix is produced tile by tile:
//
//   stage 1 (build_blocks)  every 64-bit hash is replaced by its dense rank among all
//            distinct hashes (exact, order preserving, 32 bit), and the sorted runs of
//            each block of TB = 128 sources are merged into ONE sorted list of
//            distinct ranks with postings (which of the 128 sources hold the key) —
//            so a key is compared once per block pair instead of once per source
//            pair (128x fewer comparisons, half the bytes).
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
// HBM/L2 stream of the block lists (8 B per key: rank + posting word), see DESIGN.md.
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
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

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
constexpr int CELL_TARGET = 216;  // mean keys of the longer list per cell (253 always fit a window)
constexpr int WIN = 256;      // keys per chunk / LDS window: 4 per lane of a wavefront
constexpr u32 BIG = 0xE0000000u;      // posting word: more than INLINE_MAX sources, mask index in the low bits
constexpr u32 INLINE_MAX = 4;         // sources whose 7-bit ids fit into the posting word

// ------------------------------------------------------------------------------------
// stage 1 kernels
// ------------------------------------------------------------------------------------

// One workgroup per source: tag each entry with (block << 8 | local id) [and weight].
// Weighted mode also records the source's weight sum (the bound of any pair counter that source
// takes part in; unweighted: k_src_size).
template <bool W>
__global__ void k_tag(const u64* __restrict__ off, const u32* __restrict__ wts, u32* __restrict__ val32,
                      u64* __restrict__ val64, u32* __restrict__ src_bound) {
    __shared__ unsigned long long acc;
    const u32 s = blockIdx.x;
    const u64 b = off[s], e = off[s + 1];
    const u32 tag = ((s / TB) << 8) | (s % TB);
    if (W) { if (threadIdx.x == 0) acc = 0; __syncthreads(); }
    unsigned long long part = 0;
    for (u64 i = b + threadIdx.x; i < e; i += blockDim.x) {
        if (W) { const u32 w = wts[i]; part += w; val64[i] = ((u64)w << 32) | tag; }
        else val32[i] = tag;
    }
    if (W) {
        if (part) atomicAdd(&acc, part);
        __syncthreads();
        if (threadIdx.x == 0) src_bound[s] = (u32)(acc > 0xFFFFFFFFull ? 0xFFFFFFFFull : acc);
    }
}
__global__ void k_src_size(const u64* __restrict__ off, u32* __restrict__ src_bound, u32 n_sources) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_sources) { const u64 c = off[s + 1] - off[s]; src_bound[s] = (u32)(c > 0xFFFFFFFFull ? 0xFFFFFFFFull : c); }
}

// ---- source reordering ---------------------------------------------------------------------
// Sources that share keys are moved next to each other before they are cut into blocks: a source's
// label is the smallest source id among the holders of any of its shared keys (one round of
// min-label propagation over the key groups), and the sources are ordered by (label, id).  Related
// sources then meet inside a block — their common keys collapse into one list word with a
// multi-source posting — and most block pairs share no key at all, which the join skips.  The
// engine works on the new indices; k_join maps them back when it emits an edge.
__global__ void k_iota(u32* __restrict__ p, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
__device__ inline u32 src_of_tag(u32 t) { return (t >> 8) * TB + (t & 0xFFu); }
template <class V>
__global__ void k_label(const u32* __restrict__ rk, const V* __restrict__ vals, const u32* __restrict__ first,
                        u32* __restrict__ label, const u32 skip, u64 n) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || (rk[e] & skip)) return;
    const u32 s = src_of_tag((u32)vals[e]);
    const u32 f = src_of_tag((u32)vals[first[rk[e]]]);   // entries of a key are in ascending source order
    if (f < label[s]) atomicMin(&label[s], f);
}
// Postings input (an inverted index: per key its holders, e.g. the reference's colour -> sources map):
// the state stage 1 reaches after sorting and pruning, written directly — entry tags, the key index as
// rank, and per source the bound of its pair counters (k-mer count / weight sum).
template <class V, bool W>
__global__ void k_post_expand(const u32* __restrict__ koff, const u32* __restrict__ src, const u32* __restrict__ kw,
                              V* __restrict__ vals, u32* __restrict__ rk, u32* __restrict__ src_bound, u32 n_keys,
                              u32 n_sources, u32* __restrict__ bad) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_keys) return;
    const u32 w = W ? kw[r] : 1u;
    for (u32 e = koff[r]; e < koff[r + 1]; ++e) {
        u32 s = src[e];
        if (s >= n_sources) { *bad = 1; s = 0; }   // reported as KSP_E_ARG by the caller; keep the stores in bounds
        const u32 tag = ((s / TB) << 8) | (s % TB);
        vals[e] = W ? (V)(((u64)w << 32) | tag) : (V)tag;
        rk[e] = r;
        if (w) atomicAdd(&src_bound[s], w);   // (the caller guarantees sums below 2^32)
    }
}
// order[i] = i-th source in (label, id) order  ->  newidx[order[i]] = i; order itself is the inverse map
__global__ void k_perm(const u32* __restrict__ order, u32* __restrict__ newidx, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) newidx[order[i]] = i;
}
template <class V>
__global__ void k_retag(V* __restrict__ vals, const u32* __restrict__ newidx, u64 n) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const V v = vals[e];
    const u32 ni = newidx[src_of_tag((u32)v)];
    vals[e] = (V)((v & ~(V)0xFFFFFFFFu) | (V)(((ni / TB) << 8) | (ni % TB)));
}
// per block (of the new order): the largest per-source bound
__global__ void k_blk_bound(const u32* __restrict__ src_bound, const u32* __restrict__ newidx, u32* __restrict__ blk_max,
                            u32 n_sources) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_sources) atomicMax(&blk_max[newidx[s] / TB], src_bound[s]);
}

template <class V> __device__ inline u32 tag_of(V v) { return (u32)v; }

// The global sort only looks at the top 32 significant bits of the keys (4 radix passes
// instead of up to 8).  Entries whose keys agree in those bits are adjacent afterwards;
// almost always they are copies of ONE key (the same hash in several sources).  Where two or
// more distinct keys share the bits, this kernel orders that short run by the full key so
// that equal keys become adjacent (ranks only have to be consistent, not numerically
// ordered).  Runs longer than MAX_FIX raise *overflow and the caller falls back to a
// full-width sort.
constexpr u32 MAX_FIX = 2048;
// pass 1 (streaming): positions where two neighbours share the sorted prefix but differ as
// full keys go to a work list (rare: ~D^2 / 2^33 of D distinct keys).
__global__ __launch_bounds__(1024) void k_find_mixed(const u64* __restrict__ keys, u64 n, int shift,
                                                     u32* __restrict__ list, u32* __restrict__ count, u32 cap,
                                                     u32* __restrict__ overflow) {
    // 4096 entries per workgroup; hits are compacted in LDS so that the global counter sees
    // one atomic per workgroup (a single word saturates at ~88 atomics/us on this chip)
    __shared__ u32 local[4096];
    __shared__ u32 nlocal, base;
    if (threadIdx.x == 0) nlocal = 0;
    __syncthreads();
    const u64 e0 = (u64)blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const u64 e = e0 + (u64)r * 1024;
        if (e > 0 && e < n) {
            const u64 a = keys[e - 1], b = keys[e];
            if (a != b && (a >> shift) == (b >> shift)) local[atomicAdd(&nlocal, 1u)] = (u32)e;
        }
    }
    __syncthreads();
    const u32 m = nlocal;
    if (m == 0) return;
    if (threadIdx.x == 0) base = atomicAdd(count, m);
    __syncthreads();
    for (u32 i = threadIdx.x; i < m; i += 1024) {
        if (base + i < cap) list[base + i] = local[i];
        else *overflow = 1;
    }
}
// pass 2 (work list, read-only): keep only the first listed position of every run — the one
// with no differing neighbour pair between the run's start and itself; list[i] |= DROP otherwise.
constexpr u32 DROP = 0x80000000u;
__global__ void k_mark_first(const u64* __restrict__ keys, int shift, u32* __restrict__ list,
                             const u32* __restrict__ count, u32 cap, u32* __restrict__ overflow) {
    const u32 m = min(*count, cap);
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const u64 p = list[i];
        const u64 h0 = keys[p] >> shift;
        bool first = true;
        for (u64 s = p - 1; s > 0 && (keys[s - 1] >> shift) == h0; --s) {
            if (keys[s - 1] != keys[s]) { first = false; break; }
            if (p - s > MAX_FIX) { *overflow = 1; first = false; break; }
        }
        if (!first) list[i] = (u32)p | DROP;
    }
}
// pass 3: one wavefront per kept position orders its run by the full key (runs are disjoint).
// A run (two or three keys, each held by up to a few hundred sources) is ranked through LDS: every
// entry counts the entries that must precede it (smaller key, or equal key and earlier position:
// stable, so sources stay ascending inside a key) and is scattered to that place.  Runs longer than
// FIX_WAVE entries fall back to a serial insertion sort by lane 0.
constexpr u32 FIX_WAVE = 512;
template <class V>
__global__ __launch_bounds__(256) void k_fix_runs(u64* __restrict__ keys, V* __restrict__ vals, u64 n, int shift,
                                                  const u32* __restrict__ list, const u32* __restrict__ count, u32 cap,
                                                  u32* __restrict__ overflow) {
    __shared__ u64 sk[4][FIX_WAVE];
    __shared__ V sv[4][FIX_WAVE];
    const u32 m = min(*count, cap);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (u32 i = wave; i < m; i += nwaves) {
        const u32 li = list[i];
        if (li & DROP) continue;
        const u64 p = li;
        const u64 h0 = keys[p] >> shift;
        // run start: walk back 64 entries at a time
        u64 s = p;
        bool open = true;
        while (open && p - s < FIX_WAVE) {
            const bool in = s >= (u64)(64 - lane) && (keys[s - 64 + lane] >> shift) == h0;   // entry s - 64 + lane
            const unsigned long long mb = __ballot(in);
            const int back = mb == ~0ull ? 64 : __builtin_clzll(~mb);   // run entries right before s
            s -= (u64)back;
            open = back == 64;
        }
        u64 end = p + 1;
        open = true;
        while (open && end - s <= FIX_WAVE) {
            const u64 q = end + (u64)lane;
            const bool in = q < n && (keys[q] >> shift) == h0;
            const unsigned long long mf = __ballot(in);
            const int fwd = mf == ~0ull ? 64 : __builtin_ctzll(~mf);
            end += (u64)fwd;
            open = fwd == 64;
        }
        const u32 len = (u32)(end - s);
        if (len <= FIX_WAVE) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (u32 j = lane; j < len; j += 64) { sk[wv][j] = keys[s + j]; sv[wv][j] = vals[s + j]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (u32 j = lane; j < len; j += 64) {
                const u64 k = sk[wv][j];
                u32 before = 0;
                for (u32 t = 0; t < len; ++t) {
                    const u64 kt = sk[wv][t];
                    before += (kt < k || (kt == k && t < j)) ? 1u : 0u;
                }
                keys[s + before] = k;
                vals[s + before] = sv[wv][j];
            }
            continue;
        }
        if (lane != 0) continue;
        // very long run: serial (rare)
        u64 s0 = p;
        while (s0 > 0 && (keys[s0 - 1] >> shift) == h0) --s0;   // <= MAX_FIX steps (checked by k_mark_first)
        u64 e1 = p + 1;
        while (e1 < n && (keys[e1] >> shift) == h0 && e1 - s0 <= MAX_FIX) ++e1;
        if (e1 - s0 > MAX_FIX) { *overflow = 1; continue; }
        for (u64 a = s0 + 1; a < e1; ++a) {
            const u64 k = keys[a];
            const V v = vals[a];
            u64 j = a;
            while (j > s0 && keys[j - 1] > k) { keys[j] = keys[j - 1]; vals[j] = vals[j - 1]; --j; }
            keys[j] = k;
            vals[j] = v;
        }
    }
}

// Singleton pruning + dense ranks, after the global sort by key.  A key held by exactly one
// source (a run of length 1) cannot contribute to any pair: its entry is dropped, which
// shortens every block list (less to stream and search in the join) and all later passes.
// The kept keys get dense ranks 0..U-1 — an exact, order-preserving 32-bit stand-in for the
// 64-bit hash.  One scan over packed counters: low word = kept entries, high word = kept keys.
struct PruneFn {
    const u64* keys;
    u64 n;
    __device__ u64 operator()(u64 e) const {
        const u64 k = keys[e];
        const bool head = e == 0 || keys[e - 1] != k;
        const bool last = e + 1 == n || keys[e + 1] != k;
        const bool single = head && last;
        return (single ? 0ull : 1ull) | ((u64)(head && !single) << 32);
    }
};
// Output "iterator" of the prune scan: instead of storing the packed prefix sums (and reading them back
// in a second pass), the scan's store of element e moves entry e to its place among the kept entries.
template <class V>
struct PruneScatterIt {
    using iterator_category = std::random_access_iterator_tag;
    using value_type = u64;
    using difference_type = std::ptrdiff_t;
    using pointer = void;
    struct Ctx {
        const u64* keys;
        const V* vals;
        V* vals2;
        u32* rank2;
        u32* first;
        u64* scal;
        u64 n;
    };
    struct Ref {
        Ctx c;
        u64 e;
        __device__ const Ref& operator=(const u64 cur) const {
            const u64 f = PruneFn{c.keys, c.n}(e);
            const u32 lo = (u32)cur, hi = (u32)(cur >> 32);
            if (f & 1ull) {   // kept entry
                c.vals2[lo - 1] = c.vals[e];
                c.rank2[lo - 1] = hi - 1u;
                if (f >> 32) c.first[hi - 1u] = lo - 1;   // first kept entry of its key
            }
            if (e == c.n - 1) {
                c.scal[6] = lo;   // kept entries
                c.scal[2] = hi;   // kept distinct keys (U)
            }
            return *this;
        }
    };
    using reference = Ref;
    Ctx c;
    u64 base;
    __host__ __device__ PruneScatterIt operator+(const std::ptrdiff_t d) const { return PruneScatterIt{c, base + (u64)d}; }
    __host__ __device__ PruneScatterIt& operator+=(const std::ptrdiff_t d) { base += (u64)d; return *this; }
    __device__ Ref operator[](const std::ptrdiff_t i) const { return Ref{c, base + (u64)i}; }
    __device__ Ref operator*() const { return Ref{c, base}; }
};

// 1 when entry e opens a new (block, rank) group; evaluated on the fly by the scan and the passes after it
// (two neighbouring loads of two arrays) instead of being written out by a pass of its own
template <class V>
struct HeadFn {
    const u32* rk;
    const V* vals;
    __device__ u32 operator()(u64 e) const {
        if (e == 0) return 1u;
        return ((tag_of(vals[e]) >> 8) != (tag_of(vals[e - 1]) >> 8)) || (rk[e] != rk[e - 1]) ? 1u : 0u;
    }
};

// scal[1] = Ktot (distinct (block, key) groups), estart[Ktot] = n.
template <class V>
__global__ void k_ktot(const HeadFn<V> head, const u32* __restrict__ didx, u32* __restrict__ estart,
                       u64* __restrict__ scal, u64 n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u32 k = didx[n - 1] + head(n - 1);
        estart[k] = (u32)n;
        scal[1] = k;
    }
}

// raw (unpadded) first distinct-key index of each block: entries are sorted by block, so the
// first entry of block b is found by bisection over the tags.
template <class V>
__global__ void k_blk_raw(const V* __restrict__ vals, const u32* __restrict__ didx, const u64* __restrict__ scal,
                          u32* __restrict__ blk_raw, u32 nb, u64 n) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    u64 lo = 0, hi = n;
    while (lo < hi) {
        u64 mid = lo + ((hi - lo) >> 1);
        if ((tag_of(vals[mid]) >> 8) < b) lo = mid + 1; else hi = mid;
    }
    blk_raw[b] = (lo < n) ? didx[lo] : (u32)scal[1];
}

// Padded layout of the block lists: every list starts at a multiple of 4 entries and is
// followed by >= WIN pad entries (rank PAD = +inf), so that any 16-byte-aligned window of
// WIN entries that starts inside a list is sorted and never runs into the next list.
__global__ void k_blk_pos(const u32* __restrict__ blk_raw, u32* __restrict__ blk_pos, u64* __restrict__ scal, u32 nb) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u32 pos = 0;
        for (u32 b = 0; b < nb; ++b) {
            blk_pos[b] = pos;
            u32 cnt = blk_raw[b + 1] - blk_raw[b];
            pos = ((pos + cnt + 3u) & ~3u) + WIN;
        }
        blk_pos[nb] = pos;
        scal[3] = pos;
    }
}

// tail pads of every block list (+inf ranks): from the end of list b to the start of list b + 1, and
// 4 windows of slack behind the last list
__global__ void k_pad(const u32* __restrict__ blk_raw, const u32* __restrict__ blk_pos, u32* __restrict__ brk, u32 nb, u32 padv) {
    const u32 b = blockIdx.x;
    const u32 lo = b < nb ? blk_pos[b] + (blk_raw[b + 1] - blk_raw[b]) : blk_pos[nb];
    const u32 hi = b < nb ? blk_pos[b + 1] : blk_pos[nb] + 4u * WIN;
    for (u32 i = lo + threadIdx.x; i < hi; i += blockDim.x) brk[i] = padv;
}
__global__ void k_fill(u32* __restrict__ p, u32 v, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// distinct ranks of every block (padded layout) + first entry of each group.
template <class V>
__global__ void k_emit_keys(const u32* __restrict__ rk, const V* __restrict__ vals, const HeadFn<V> head,
                            const u32* __restrict__ didx, const u32* __restrict__ blk_raw,
                            const u32* __restrict__ blk_pos, u32* __restrict__ brk, u32* __restrict__ estart, u64 n) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (head(e)) {
        u32 d = didx[e], b = tag_of(vals[e]) >> 8;
        brk[blk_pos[b] + (d - blk_raw[b])] = rk[e];
        estart[d] = (u32)e;
    }
}

__global__ void k_bigflag(const u32* __restrict__ estart, const u64* __restrict__ scal, u32* __restrict__ big,
                          u64 cap) {
    u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= cap) return;
    u32 v = 0;
    if (d < scal[1]) v = (estart[d + 1] - estart[d]) > INLINE_MAX ? 1u : 0u;
    big[d] = v;
}

// Posting word of a distinct key (which of the block's 128 sources hold it):
//   bits 31..29 = c-1 for c <= 4 sources, whose 7-bit local ids sit in bits 0..27
//   (ascending, 7 bits each);  bits 31..29 = 7 -> more than 4 sources: bits 0..28 index a
//   128-bit membership mask in `bigmask`.
template <class V, bool W>
__global__ void k_emit_info(const u32* __restrict__ estart, const u32* __restrict__ bigoff,
                            const u64* __restrict__ scal, const V* __restrict__ vals,
                            const u32* __restrict__ blk_raw, const u32* __restrict__ blk_pos, u32* __restrict__ info,
                            uint4* __restrict__ bigmask, u32* __restrict__ bw) {
    u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= scal[1]) return;
    u32 b = estart[d], c = estart[d + 1] - b;
    V v0 = vals[b];
    u32 blk = tag_of(v0) >> 8;
    u32 dst = blk_pos[blk] + ((u32)d - blk_raw[blk]);
    if (W) bw[dst] = (u32)((u64)v0 >> 32);
    if (c <= INLINE_MAX) {
        u32 inf = (c - 1) << 29;
        for (u32 i = 0; i < c; ++i) inf |= (tag_of(vals[b + i]) & 0x7F) << (7 * i);
        info[dst] = inf;
    } else {
        u32 m[4] = {0, 0, 0, 0};
        for (u32 i = 0; i < c; ++i) {
            u32 id = tag_of(vals[b + i]) & 0x7F;
            m[id >> 5] |= 1u << (id & 31);
        }
        u32 o = bigoff[d];
        info[dst] = BIG | o;
        bigmask[o] = make_uint4(m[0], m[1], m[2], m[3]);
    }
}

// Fine cell index: cidx[b][f] = position (padded layout) of the first key of block b whose rank
// is >= f * ceil(U / ncell)  (f = 0..ncell).  Ranks are dense, so equal rank ranges are equal
// shares of the distinct keys whatever the distribution of the hash values.
__global__ void k_cidx(const u32* __restrict__ brk, const u32* __restrict__ blk_raw, const u32* __restrict__ blk_pos,
                       const u64* __restrict__ scal, u32* __restrict__ cidx, u32 nb, u32 ncell) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u64)nb * (ncell + 1)) return;
    u32 b = (u32)(i / (ncell + 1)), f = (u32)(i % (ncell + 1));
    u32 lo = blk_pos[b], hi = lo + (blk_raw[b + 1] - blk_raw[b]);
    if (f == ncell) { cidx[i] = hi; return; }
    u64 step = (scal[2] + ncell - 1) / ncell;
    u64 v = (u64)f * step;
    while (lo < hi) {
        u32 mid = lo + ((hi - lo) >> 1);
        if ((u64)brk[mid] < v) lo = mid + 1; else hi = mid;
    }
    cidx[i] = lo;
}

__device__ inline u64 tile_row_start_dev(u64 r, u64 nb) { return r * nb - r * (r - 1) / 2; }

// ---- which block pairs share a key, and how much work a diagonal tile is ------------------------
// (rank, block) of every list word; sorted by rank, the words of one key are adjacent and the
// block pairs among them are exactly the tiles that have something to count.
__global__ void k_list_pairs(const u32* __restrict__ brk, const u32* __restrict__ info, const uint4* __restrict__ bigmask,
                             const u32* __restrict__ blk_raw, const u32* __restrict__ blk_pos, u32* __restrict__ pr,
                             u32* __restrict__ pb, unsigned long long* __restrict__ work) {
    // grid (block, share): also sums the pair updates of the block's diagonal tile, C(holders, 2) per key
    const u32 b = blockIdx.x;
    const u32 cnt = blk_raw[b + 1] - blk_raw[b], src = blk_pos[b], dst = blk_raw[b];
    const u32 i0 = (u32)(((u64)cnt * blockIdx.y) / gridDim.y), i1 = (u32)(((u64)cnt * (blockIdx.y + 1)) / gridDim.y);
    unsigned long long acc = 0, holders = 0;
    for (u32 i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        pr[dst + i] = brk[src + i];
        pb[dst + i] = b;
        const u32 inf = info[src + i];
        u32 c;
        if (inf >= BIG) { const uint4 m = bigmask[inf & ~BIG]; c = __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w); }
        else c = (inf >> 29) + 1;
        acc += (unsigned long long)c * (c - 1) / 2;
        holders += c;
    }
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_down(acc, o); holders += __shfl_down(holders, o); }
    if ((threadIdx.x & 63) == 0) {
        if (acc) atomicAdd(&work[b], acc);
        if (holders) atomicAdd(&work[gridDim.x], holders);   // slot nb: holders summed over all list words
    }
}
// one byte per tile: plain idempotent stores (a few hundred active tiles take millions of hits —
// atomics on the same words would serialise in L2)
__global__ void k_tile_flags(const u32* __restrict__ pr, const u32* __restrict__ pb, u64 n, u32 nb,
                             unsigned char* __restrict__ flags) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 r = pr[i], I = pb[i];
    for (u64 j = i + 1; j < n && pr[j] == r; ++j) {   // (stable sort: blocks ascend inside a key)
        const u64 t = tile_row_start_dev(I, nb) + (pb[j] - I);
        if (!flags[t]) flags[t] = 1;
    }
}
__global__ void k_pack_flags(const unsigned char* __restrict__ flags, u64 n, u32* __restrict__ bits) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;   // blockDim is a multiple of 64
    const unsigned long long m = __ballot(t < n && flags[t] != 0);
    if ((threadIdx.x & 63) == 0) { bits[t >> 5] = (u32)m; bits[(t >> 5) + 1] = (u32)(m >> 32); }
}

// ---- key-range slices (multi-GPU build) -----------------------------------------------
// Rank p of G builds the block lists of the keys in its 1/G share of the hash range only
// (filter -> same pipeline on n/G entries); the slices are exchanged (all-gather) and every
// rank assembles the full lists: slice p's ranks are shifted by the number of distinct keys
// of the slices before it, so concatenating the slices of a block in part order is sorted.
// Every source's run is sorted, so its keys inside [lo, hi] are one contiguous sub-run: two
// bisections per source instead of a pass over all entries.
__global__ void k_range_bounds(const u64* __restrict__ keys, const u64* __restrict__ off, u64 lo, u64 hi,
                               u32* __restrict__ first, u32* __restrict__ cnt, u32 n_sources) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sources) return;
    const u64 b = off[s], e = off[s + 1];
    u64 l = b, r = e;
    while (l < r) { u64 m = l + ((r - l) >> 1); if (keys[m] < lo) l = m + 1; else r = m; }
    const u64 a = l;
    r = e;
    while (l < r) { u64 m = l + ((r - l) >> 1); if (keys[m] <= hi) l = m + 1; else r = m; }
    first[s] = (u32)(a - b);
    cnt[s] = (u32)(l - a);
}
// one workgroup per source: copy its sub-run and tag it (block << 8 | local id [| weight << 32])
template <class V, bool W>
__global__ void k_range_copy(const u64* __restrict__ keys, const u32* __restrict__ wts, const u64* __restrict__ off,
                             const u32* __restrict__ first, const u32* __restrict__ cnt, const u32* __restrict__ fpos,
                             u64* __restrict__ fkeys, V* __restrict__ ftags) {
    const u32 s = blockIdx.x;
    const u64 src = off[s] + first[s];
    const u32 c = cnt[s], dst = fpos[s];
    const u32 tag = ((s / TB) << 8) | (s % TB);
    for (u32 i = threadIdx.x; i < c; i += blockDim.x) {
        fkeys[dst + i] = keys[src + i];
        if (W) ftags[dst + i] = (V)(((u64)wts[src + i] << 32) | tag);
        else ftags[dst + i] = (V)tag;
    }
}
__global__ void k_range_total(const u32* __restrict__ fpos, const u32* __restrict__ cnt, u64* __restrict__ scal, u32 n_sources) {
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[8] = (u64)fpos[n_sources - 1] + cnt[n_sources - 1];
}
// largest key = largest last element of the sorted runs
__global__ void k_max_last(const u64* __restrict__ keys, const u64* __restrict__ off, unsigned long long* __restrict__ out,
                           u32 n_sources) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (s < n_sources && off[s + 1] > off[s]) v = keys[off[s + 1] - 1];
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned long long)__shfl_down(v, o));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}

__global__ void k_nbig(const u32* __restrict__ bigflag, const u32* __restrict__ bigoff, u64* __restrict__ scal) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const u64 k = scal[1];
        scal[7] = k ? (u64)bigoff[k - 1] + bigflag[k - 1] : 0;
    }
}
// global block counts from the parts' counts (serial: nb x parts is small)
__global__ void k_asm_counts(const u32* __restrict__ raw_all, u32 stride, u32 nparts, u32 nb, u32* __restrict__ blk_raw) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        u32 acc = 0;
        for (u32 b = 0; b < nb; ++b) {
            blk_raw[b] = acc;
            for (u32 p = 0; p < nparts; ++p) acc += raw_all[(size_t)p * stride + b + 1] - raw_all[(size_t)p * stride + b];
        }
        blk_raw[nb] = acc;
    }
}
// one workgroup per (block, part): copy the part's slice of the block into the full list
template <bool W>
__global__ void k_asm_copy(const u32* __restrict__ brk_all, const u32* __restrict__ info_all,
                           const u32* __restrict__ bw_all, size_t lstride, const u32* __restrict__ raw_all,
                           const u32* __restrict__ pos_all, u32 bstride, const u32* __restrict__ rank_off,
                           const u32* __restrict__ big_off, const u32* __restrict__ blk_pos, u32* __restrict__ brk,
                           u32* __restrict__ info, u32* __restrict__ bw) {
    const u32 b = blockIdx.x, p = blockIdx.y;
    u32 before = 0;
    for (u32 q = 0; q < p; ++q) before += raw_all[(size_t)q * bstride + b + 1] - raw_all[(size_t)q * bstride + b];
    const u32 cnt = raw_all[(size_t)p * bstride + b + 1] - raw_all[(size_t)p * bstride + b];
    const size_t src = (size_t)p * lstride + pos_all[(size_t)p * bstride + b];
    const u32 dst = blk_pos[b] + before;
    const u32 ro = rank_off[p], bo = big_off[p];
    for (u32 i = threadIdx.x; i < cnt; i += blockDim.x) {
        brk[dst + i] = brk_all[src + i] + ro;
        u32 inf = info_all[src + i];
        if (inf >= BIG) inf = BIG | ((inf & ~BIG) + bo);
        info[dst + i] = inf;
        if (W) bw[dst + i] = bw_all[src + i];
    }
}

// ------------------------------------------------------------------------------------
// stage 2: the join kernel
// ------------------------------------------------------------------------------------
struct JoinArgs {
    const u32* brk;     // block lists: distinct key ranks, ascending inside a block
    const u32* info;
    const u32* bw;      // NULL -> weight 1
    const uint4* bigmask; // 128-bit membership masks of the postings with > 4 sources
    const u32* blk_raw; // nb + 1: unpadded distinct-key offsets (counts)
    const u32* blk_pos; // nb + 1: start of every block list in the padded layout
    const u32* cidx;    // nb * (ncell + 1): fine cell index, positions in the padded layout
    u32 ncell;          // fine cells per block (power of two, >= NP)
    const u32* blk_max; // nb: largest per-source k-mer count (weight sum) in the block
    const u32* inv;     // engine source index -> caller's source id
    u32 collect;        // unweighted off-diagonal tiles: 1 = collect + bit-sliced accumulation, 0 = LDS counters
    // work-list mode (sched != NULL): only the block pairs that share a key are visited, and a
    // tile is cut into as many shares (rank ranges / key ranges) as its estimated work asks for
    const u32* sched;   // per workgroup of the whole work list: index of its active tile
    const u32* act;     // per active tile: I, J, first workgroup, index among the split tiles (4 x u32)
    u32 wg0;            // first workgroup of this launch in the work list
    u32 split0;         // index of the first split tile of this join call (tail buffer slot 0)
    u32 nb;
    u32 n_sources;
    u64 tile_begin;
    ksp_edge* out;
    u64 cap;
    unsigned long long* out_count;
    u32 n_normal;       // blocks [0, n_normal) own one tile each; the rest split the tail tiles
    u32 tail_sp;        // workgroups per tail tile (each takes 1/tail_sp of the rank range)
    u32* tailbuf;       // n_tail x TB*TB 32-bit counters the tail workgroups add into
    u32* tail_done;     // work-list mode: per split tile, the shares that have added their counters
    u32 dbg;            // timing-only ablation switches (-DKSP_ABLATE builds + KSP_DEBUG_ABLATE; results are wrong when set)
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

// Ranks are < 2^30, so these never equal a real rank; as signed ints they are positive,
// which keeps the branch-free "b < a" test ((int)(b - a) >> 31) exact.
constexpr u32 PAD = 0x7FFFFFFFu;     // +inf: tail padding of every block list
constexpr u32 INF_A = 0x7FFFFFFEu;   // masked A keys / "last key" of a final A chunk
constexpr u32 INF_B = 0x7FFFFFFFu;   // "last key" of a final B window

// One 256-key chunk of a block list, 4 consecutive ranks per lane (one 16-byte load).
// Only ranks are streamed; posting words / weights are gathered for matching keys only.
__device__ inline uint4 load_a(const JoinArgs& a, u32 cbase, u32 pa, u32 ea, int lane) {
    uint4 k = make_uint4(INF_A, INF_A, INF_A, INF_A);
    if (cbase < ea) {   // wave-uniform
        k = reinterpret_cast<const uint4*>(a.brk)[(cbase >> 2) + lane];
        if (cbase < pa || cbase + WIN > ea) {   // wave-uniform: only the first / last chunk of a part
            const u32 p0 = cbase + 4u * lane;
            // keys outside [pa, ea) belong to a neighbouring part: mask them
            k.x = (p0 >= pa && p0 < ea) ? k.x : INF_A;
            k.y = (p0 + 1 >= pa && p0 + 1 < ea) ? k.y : INF_A;
            k.z = (p0 + 2 >= pa && p0 + 2 < ea) ? k.z : INF_A;
            k.w = (p0 + 3 >= pa && p0 + 3 < ea) ? k.w : INF_A;
        }
    }
    return k;
}
__device__ inline uint4 load_b(const JoinArgs& a, u32 wbase, u32 eb, int lane) {
    uint4 k = make_uint4(PAD, PAD, PAD, PAD);
    // a window that starts inside the part stays sorted: it may run into the next part
    // (larger keys, harmless because A is masked) and into the +inf tail pads
    if (wbase < eb) k = reinterpret_cast<const uint4*>(a.brk)[(wbase >> 2) + lane];
    return k;
}

// (s < a) as 0/1 without touching VCC (both < 2^31).
__device__ inline u32 lt(u32 s, u32 a) { return (s - a) >> 31; }

// B window of one wave in LDS: the 256 sorted ranks (leaf level: 64 nodes of 4) plus two
// inner levels of an implicit 4-ary search tree over the leaves' last keys.  The root
// (3 separators) lives in SGPRs.  Every level is one conflict-free ds_read_b128.
struct Window {
    uint4 leaf[64];   // leaf[l] = ranks 4l .. 4l+3
    uint4 l2[16];     // l2[m]   = last rank of leaves 4m .. 4m+3
    uint4 l1[4];      // l1[q]   = last rank of leaves 16q+3, 16q+7, 16q+11, 16q+15
};

// position (0..255) of the first window entry >= key, and whether it equals key
__device__ inline u32 window_find(const Window& w, u32 s0, u32 s1, u32 s2, u32 key, bool& hit) {
    const u32 c0 = lt(s0, key) + lt(s1, key) + lt(s2, key);
    const uint4 n1 = w.l1[c0];
    const u32 m = 4u * c0 + lt(n1.x, key) + lt(n1.y, key) + lt(n1.z, key);
    const uint4 n2 = w.l2[m];
    const u32 lb = 4u * m + lt(n2.x, key) + lt(n2.y, key) + lt(n2.z, key);
    const uint4 lf = w.leaf[lb];
    const u32 c3 = lt(lf.x, key) + lt(lf.y, key) + lt(lf.z, key);
    hit = (lf.x == key) | (lf.y == key) | (lf.z == key) | (lf.w == key);
    return 4u * lb + c3;
}

// ---- applying a match to the LDS tile of pair counters -------------------------------
// C16: two 16-bit counters per LDS word (row-major, even column in the low half).  Exact
// whenever every counter of the tile stays < 2^16, which the kernel guarantees by only
// taking tiles where one of the two blocks holds no source with >= 65536 k-mers
// (shared <= min(n_a, n_b)).  Halves the tile to 32 KB -> 3 workgroups per CU.
template <bool C16>
__device__ inline void s_add(u32* S, u32 idx, u32 w) {
    if (C16) atomicAdd(&S[idx >> 1], w << ((idx & 1u) * 16u));
    else atomicAdd(&S[idx], w);
}

// Small postings (<= 4 sources, inline in the posting word): the lane adds its own
// cross product.  No memory traffic besides the LDS atomics.
template <bool C16>
__device__ inline void add_inline(u32* S, u32 ia, u32 ib, u32 w) {
    const u32 nA = (ia >> 29) + 1, nB = (ib >> 29) + 1;
    for (u32 x = 0; x < nA; ++x) {
        const u32 row = ((ia >> (7 * x)) & 127u) * TB;
        for (u32 y = 0; y < nB; ++y) s_add<C16>(S, row + ((ib >> (7 * y)) & 127u), w);
    }
}

// 128-bit membership mask of a posting word (wave-uniform arguments).
__device__ inline uint4 posting_mask(u32 inf, const uint4* __restrict__ bigmask) {
    if (inf >= BIG) return bigmask[inf & ~BIG];
    u32 m[4] = {0, 0, 0, 0};
    const u32 n = (inf >> 29) + 1;
    for (u32 x = 0; x < n; ++x) {
        u32 id = (inf >> (7 * x)) & 127u;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] |= (id >> 5) == (u32)k ? (1u << (id & 31)) : 0u;
    }
    return make_uint4(m[0], m[1], m[2], m[3]);
}

// Large postings: the whole wave expands one match, cA x cB counter updates.
// SELF: both postings are the same key of the same block -> only pairs row < column.
// Dense form (postings of up to DENSE_MAX sources on the column side): both member sets are
// compacted into id lists in LDS and the lanes walk the cA x cB grid in 8 x 8 patches — every LDS
// atomic carries up to 64 updates, however sparse the masks are.  Related sources sit in the same
// block (source reordering), so postings of 5 .. 40 sources are the common case.
// Row form (fuller masks): lanes own the columns (lane, lane + 64), rows come from a scalar walk
// over the bits of mask A; every LDS atomic touches 64 consecutive counters (conflict free).
constexpr u32 DENSE_MAX = 48;        // two different postings (cA x cB grid)
constexpr u32 DENSE_MAX_SELF = 128;  // one posting against itself (triangle: half the patches)
__device__ inline void mask_to_list(unsigned char* l, const u32 m0, const u32 m1, const u32 m2, const u32 m3, const int lane) {
    const u32 below_lo = __builtin_amdgcn_mbcnt_hi(m1, __builtin_amdgcn_mbcnt_lo(m0, 0));   // members among columns < lane
    const u32 below_hi = __builtin_amdgcn_mbcnt_hi(m3, __builtin_amdgcn_mbcnt_lo(m2, 0));   // ... among columns 64 .. 64 + lane - 1
    const u32 wlo = lane < 32 ? m0 : m1, whi = lane < 32 ? m2 : m3;
    if ((wlo >> (lane & 31)) & 1u) l[below_lo] = (unsigned char)lane;
    if ((whi >> (lane & 31)) & 1u) l[(u32)__popc(m0) + (u32)__popc(m1) + below_hi] = (unsigned char)(lane + 64);
}
template <bool SELF, bool C16>
__device__ inline void add_masks(u32* S, unsigned char* lst, uint4 mA, uint4 mB, u32 w, int lane) {
    const u32 a0 = __builtin_amdgcn_readfirstlane(mA.x), a1 = __builtin_amdgcn_readfirstlane(mA.y);
    const u32 a2 = __builtin_amdgcn_readfirstlane(mA.z), a3 = __builtin_amdgcn_readfirstlane(mA.w);
    const u32 b0 = __builtin_amdgcn_readfirstlane(mB.x), b1 = __builtin_amdgcn_readfirstlane(mB.y);
    const u32 b2 = __builtin_amdgcn_readfirstlane(mB.z), b3 = __builtin_amdgcn_readfirstlane(mB.w);
    const u32 ca = (u32)(__popc(a0) + __popc(a1) + __popc(a2) + __popc(a3));
    const u32 cb = SELF ? ca : (u32)(__popc(b0) + __popc(b1) + __popc(b2) + __popc(b3));
    if (cb <= (SELF ? DENSE_MAX_SELF : DENSE_MAX)) {
        unsigned char* la = lst;
        unsigned char* lb = SELF ? lst : lst + TB;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        mask_to_list(la, a0, a1, a2, a3, lane);
        if (!SELF) mask_to_list(lb, b0, b1, b2, b3, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const u32 li = (u32)lane >> 3, lj = (u32)lane & 7u;
        for (u32 bi = 0; bi < ca; bi += 8) {
            const u32 i = bi + li;
            const u32 row = (u32)la[min(i, (u32)TB - 1u)] * TB;
            for (u32 bj = SELF ? bi : 0u; bj < cb; bj += 8) {
                const u32 j = bj + lj;
                const u32 col = lb[min(j, (u32)TB - 1u)];
                if (i < ca && j < cb && (!SELF || i < j)) s_add<C16>(S, row + col, w);   // (lists ascend: row < col in a self tile)
            }
        }
        return;
    }
    const u32 bw0 = lane < 32 ? b0 : b1, bw1 = lane < 32 ? b2 : b3;
    const bool c0 = (bw0 >> (lane & 31)) & 1u, c1 = (bw1 >> (lane & 31)) & 1u;
    const u32 words[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u32 word = words[k];
        while (word) {
            const u32 r = 32u * k + (u32)__builtin_ctz(word);
            word &= word - 1;
            const u32 row = r * TB;
            if (c0 && (!SELF || (u32)lane > r)) s_add<C16>(S, row + lane, w);
            if (c1 && (!SELF || (u32)lane + 64u > r)) s_add<C16>(S, row + lane + 64, w);
        }
    }
}

// One match per lane, posting words fetched one step ahead of their use.
struct Pending {
    u32 ia, ib, w;
    bool valid;
};

// Apply the pending matches of the wave to the tile.  Fast path (both keys held by a
// single source of their block): one LDS atomic per lane, no loop.  Postings with 2..4
// sources: nested loops with wave-uniform trip counts.  Larger postings: the whole wave
// expands one match at a time from the 128-bit masks.
// postings with 2..4 sources, or > 4 (masks): out of line, the hot path stays small
template <bool C16>
__device__ inline void pending_apply_complex(u32* S, unsigned char* lst, const uint4* __restrict__ bigmask, const u32 qia,
                                                   const u32 qib, const u32 qw, const bool cx, int lane) {
    const u32 both = qia | qib;
    const bool small = cx && both < BIG;
    const u32 nA = small ? (qia >> 29) + 1 : 0, nB = small ? (qib >> 29) + 1 : 0;
    for (u32 x = 0; x < INLINE_MAX; ++x) {
        if (__ballot(x < nA) == 0) break;
        const u32 row = ((qia >> (7 * x)) & 127u) << 7;
        for (u32 y = 0; y < INLINE_MAX; ++y) {
            const bool act = x < nA && y < nB;
            if (__ballot(act) == 0) break;
            if (act) s_add<C16>(S, row | ((qib >> (7 * y)) & 127u), qw);
        }
    }
    const bool large = cx && !small;
    unsigned long long todo = __ballot(large);
    if (todo == 0) return;
    // every lane fetches the masks of its own match first: one round of memory latency for the
    // whole wave instead of one per match inside the serial loop below
    uint4 mA = make_uint4(0, 0, 0, 0), mB = mA;
    if (large) { mA = posting_mask(qia, bigmask); mB = posting_mask(qib, bigmask); }
    while (todo) {   // wave-cooperative expansion, one match at a time
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint4 a = make_uint4(__builtin_amdgcn_readlane(mA.x, src), __builtin_amdgcn_readlane(mA.y, src),
                                   __builtin_amdgcn_readlane(mA.z, src), __builtin_amdgcn_readlane(mA.w, src));
        const uint4 b = make_uint4(__builtin_amdgcn_readlane(mB.x, src), __builtin_amdgcn_readlane(mB.y, src),
                                   __builtin_amdgcn_readlane(mB.z, src), __builtin_amdgcn_readlane(mB.w, src));
        add_masks<false, C16>(S, lst, a, b, __builtin_amdgcn_readlane(qw, src), lane);
    }
}

// Apply the pending matches of the wave to the tile.  Fast path (both keys held by a
// single source of their block): one LDS atomic per lane, no loop.
template <bool C16>
__device__ inline void pending_apply(u32* S, unsigned char* lst, const uint4* __restrict__ bigmask, const Pending& q, int lane) {
    const u32 both = q.ia | q.ib;
    const bool simple = q.valid && both < 128u;
    if (simple) s_add<C16>(S, (q.ia << 7) | q.ib, q.w);
    const bool cx = q.valid && !simple;
    if (__ballot(cx) != 0) pending_apply_complex<C16>(S, lst, bigmask, q.ia, q.ib, q.w, cx, lane);
}

// Per-wave LDS state of the join.
struct WaveLds {
    Window win;                 // B window as a 4-ary search tree
    unsigned short mq[WIN];     // match queue: (A slot << 8) | B slot
    unsigned char lst[2 * TB];  // member lists of the two postings being expanded (add_masks)
};

// One step: every lane looks its 4 A keys (chunk base `ca`) up in the B window (base `cb`);
// matches are queued, their posting words are fetched (consumed by the NEXT step) and the
// previous step's matches are applied to the tile.
template <bool W, bool C16>
__device__ inline void match_step(const JoinArgs& a, u32* S, WaveLds& wl, const uint4 A0, const uint4 B0,
                                  const u32 ca, const u32 cb, const bool newB, u32& s0, u32& s1, u32& s2,
                                  Pending& pend, const int lane) {
    Window& wn = wl.win;
    if (newB) {   // (re)build the window of this wave
        u32* l2w = reinterpret_cast<u32*>(wn.l2);
        u32* l1w = reinterpret_cast<u32*>(wn.l1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        wn.leaf[lane] = B0;
        l2w[lane] = B0.w;
        if ((lane & 3) == 3) l1w[lane >> 2] = B0.w;
        s0 = __builtin_amdgcn_readlane(B0.w, 15);
        s1 = __builtin_amdgcn_readlane(B0.w, 31);
        s2 = __builtin_amdgcn_readlane(B0.w, 47);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    bool h0, h1, h2, h3;
#ifdef KSP_ABLATE
    if (a.dbg & 4) { asm volatile("" :: "v"(A0.x), "v"(A0.y), "v"(A0.z), "v"(A0.w)); return; }
#endif
    const u32 p0 = window_find(wn, s0, s1, s2, A0.x, h0);
    const u32 p1 = window_find(wn, s0, s1, s2, A0.y, h1);
    const u32 p2 = window_find(wn, s0, s1, s2, A0.z, h2);
    const u32 p3 = window_find(wn, s0, s1, s2, A0.w, h3);
#ifdef KSP_ABLATE
    if (a.dbg & 8) { h0 = h1 = h2 = h3 = false; asm volatile("" :: "v"(p0), "v"(p1), "v"(p2), "v"(p3)); }
#endif
    // compact the matches (A slot 0..255, B slot 0..255) into the wave's queue
    unsigned short* q16 = wl.mq;
    u32 cnt = 0;
    {
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
        const u32 lo = 4u * lane;
        if (h0) q16[cnt + __builtin_amdgcn_mbcnt_hi((u32)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m0, 0))] = (unsigned short)(((lo) << 8) | p0);
        cnt += (u32)__popcll(m0);
        if (h1) q16[cnt + __builtin_amdgcn_mbcnt_hi((u32)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m1, 0))] = (unsigned short)(((lo + 1) << 8) | p1);
        cnt += (u32)__popcll(m1);
        if (h2) q16[cnt + __builtin_amdgcn_mbcnt_hi((u32)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m2, 0))] = (unsigned short)(((lo + 2) << 8) | p2);
        cnt += (u32)__popcll(m2);
        if (h3) q16[cnt + __builtin_amdgcn_mbcnt_hi((u32)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m3, 0))] = (unsigned short)(((lo + 3) << 8) | p3);
        cnt += (u32)__popcll(m3);
    }
    // matches of the previous step: their posting words have arrived by now
#ifdef KSP_ABLATE
    if (!(a.dbg & 1))
#endif
    pending_apply<C16>(S, wl.lst, a.bigmask, pend, lane);
    pend.valid = false;
#ifdef KSP_ABLATE
    if (a.dbg & 2) cnt = 0;
#endif
    // fetch the posting words of this step's matches (consumed by the next step).  Straight-line
    // for the first 64; the rare surplus (> 64 matches in one step) is gathered and applied at once.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (cnt > 64) {
        for (u32 base = 64; base < cnt; base += 64) {
            Pending extra;
            extra.valid = base + (u32)lane < cnt;
            extra.ia = 0; extra.ib = 0; extra.w = 1;
            if (extra.valid) {
                const u32 e = q16[base + lane];
                const u32 qa = ca + (e >> 8), qb = cb + (e & 255u);
                extra.ia = a.info[qa];
                extra.ib = a.info[qb];
                if (W) extra.w = a.bw[qa];
            }
            pending_apply<C16>(S, wl.lst, a.bigmask, extra, lane);
        }
    }
    {
        const bool v = (u32)lane < cnt;
        pend.valid = v;
        const u32 e = v ? q16[lane] : 0u;
        const u32 qa = v ? ca + (e >> 8) : ca, qb = v ? cb + (e & 255u) : cb;   // always a valid address
        pend.ia = a.info[qa];
        pend.ib = a.info[qb];
        if (W) pend.w = a.bw[qa];
    }
}

// Sliding-window merge of two rank ranges (general: any key distribution).
template <bool W, bool C16>
__device__ inline void join_windows(const JoinArgs& a, u32* S, WaveLds& wl, const u32 I, const u32 J, const int wv,
                                    const int lane, const u32 sub, const u32 sp) {
    const u32 stride = a.ncell / NP;   // NP coarse rank ranges out of the fine cell index
    const u32* cI = a.cidx + (size_t)I * (a.ncell + 1);
    const u32* cJ = a.cidx + (size_t)J * (a.ncell + 1);
    const int pbeg = (int)((NP * sub) / sp), pend_ = (int)((NP * (sub + 1)) / sp);   // this workgroup's share
    for (int p = pbeg + wv; p < pend_; p += JW) {   // equal shares of the key space: static round-robin over waves
        const u32 pa = __builtin_amdgcn_readfirstlane(cI[p * stride]);
        const u32 ea = __builtin_amdgcn_readfirstlane(cI[(p + 1) * stride]);
        const u32 pb = __builtin_amdgcn_readfirstlane(cJ[p * stride]);
        const u32 eb = __builtin_amdgcn_readfirstlane(cJ[(p + 1) * stride]);
        if (pa >= ea || pb >= eb) continue;
        u32 ca = pa & ~3u, cb = pb & ~3u;   // 16-byte aligned bases of the current chunk / window
        uint4 A0 = load_a(a, ca, pa, ea, lane);
        uint4 A1 = load_a(a, ca + WIN, pa, ea, lane);
        uint4 A2 = load_a(a, ca + 2 * WIN, pa, ea, lane);
        uint4 B0 = load_b(a, cb, eb, lane);
        uint4 B1 = load_b(a, cb + WIN, eb, lane);
        uint4 B2 = load_b(a, cb + 2 * WIN, eb, lane);
        Pending pend;
        pend.ia = 0; pend.ib = 0; pend.w = 1; pend.valid = false;
        bool newB = true;
        u32 s0 = 0, s1 = 0, s2 = 0;
        while (true) {
            match_step<W, C16>(a, S, wl, A0, B0, ca, cb, newB, s0, s1, s2, pend, lane);
            // advance whichever side ends first (both on a tie)
            const bool afin = ca + WIN >= ea, bfin = cb + WIN >= eb;
            const u32 aLast = afin ? INF_A : (u32)__builtin_amdgcn_readlane(A0.w, 63);
            const u32 bLast = bfin ? INF_B : (u32)__builtin_amdgcn_readlane(B0.w, 63);
            const bool advA = aLast <= bLast, advB = bLast <= aLast;
            if ((advA && afin) || (advB && bfin)) break;
            newB = advB;
            if (advA) {
                ca += WIN;
                A0 = A1; A1 = A2;
                A2 = load_a(a, ca + 2 * WIN, pa, ea, lane);
            }
            if (advB) {
                cb += WIN;
                B0 = B1; B1 = B2;
                B2 = load_b(a, cb + 2 * WIN, eb, lane);
            }
        }
        pending_apply<C16>(S, wl.lst, a.bigmask, pend, lane);
    }
}

// Rank-aligned cells: both lists are cut at the same rank boundaries (every m-th entry of the
// fine cell index, m chosen per tile so that a cell holds ~176 keys of the longer list).  A cell
// is one A chunk against one B window: every A key is searched once, nothing is advanced, and
// the next two cells are prefetched into two alternating register sets (no register shifting).
struct CellLoad {
    uint4 A, B;
    u32 ca, cb, a0, a1, b0, b1;
    bool simple, work;
};
__device__ inline void cell_fetch(const JoinArgs& a, const u32* cI, const u32* cJ, const u32 c, const u32 cend,
                                  const u32 m, const int lane, CellLoad& L) {
    L.simple = false;
    L.work = false;
    L.A = make_uint4(INF_A, INF_A, INF_A, INF_A);
    L.B = make_uint4(PAD, PAD, PAD, PAD);
    L.ca = L.cb = L.a0 = L.a1 = L.b0 = L.b1 = 0;
    if (c >= cend) return;   // (only in the last iteration of a wave's range)
    const u32 f0 = c * m, f1 = min(a.ncell, f0 + m);
    L.a0 = __builtin_amdgcn_readfirstlane(cI[f0]);
    L.a1 = __builtin_amdgcn_readfirstlane(cI[f1]);
    L.b0 = __builtin_amdgcn_readfirstlane(cJ[f0]);
    L.b1 = __builtin_amdgcn_readfirstlane(cJ[f1]);
    L.ca = L.a0 & ~3u;
    L.cb = L.b0 & ~3u;
    L.work = L.a1 > L.a0 && L.b1 > L.b0;
    L.simple = L.work && (L.a1 - L.ca <= (u32)WIN) && (L.b1 - L.cb <= (u32)WIN);
    // always two loads (a fixed instruction stream lets hipcc count its waits); unused ones hit
    // the cell's own 16-byte aligned start, which is always inside the padded arrays
    L.A = reinterpret_cast<const uint4*>(a.brk)[(L.ca >> 2) + lane];
    L.B = reinterpret_cast<const uint4*>(a.brk)[(L.cb >> 2) + lane];
    if (L.simple && (L.ca < L.a0 || L.ca + WIN > L.a1)) {   // mask A keys outside the cell
        const u32 p0 = L.ca + 4u * lane;
        L.A.x = (p0 >= L.a0 && p0 < L.a1) ? L.A.x : INF_A;
        L.A.y = (p0 + 1 >= L.a0 && p0 + 1 < L.a1) ? L.A.y : INF_A;
        L.A.z = (p0 + 2 >= L.a0 && p0 + 2 < L.a1) ? L.A.z : INF_A;
        L.A.w = (p0 + 3 >= L.a0 && p0 + 3 < L.a1) ? L.A.w : INF_A;
    }
}

// oversized cell (skewed key distribution): all chunk x window combinations; rare, kept out of line
template <bool W, bool C16>
__device__ inline void cell_process_big(const JoinArgs& a, u32* S, WaveLds& wl, const CellLoad& L, u32& s0,
                                              u32& s1, u32& s2, Pending& pend, const int lane) {
    for (u32 wb = L.cb; wb < L.b1; wb += WIN) {
        const uint4 B = load_b(a, wb, L.b1, lane);
        bool first = true;
        for (u32 ca = L.ca; ca < L.a1; ca += WIN) {
            const uint4 A = load_a(a, ca, L.a0, L.a1, lane);
            match_step<W, C16>(a, S, wl, A, B, ca, wb, first, s0, s1, s2, pend, lane);
            first = false;
        }
    }
}

template <bool W, bool C16>
__device__ inline void cell_process(const JoinArgs& a, u32* S, WaveLds& wl, const CellLoad& L, u32& s0, u32& s1,
                                    u32& s2, Pending& pend, const int lane) {
    if (L.simple) match_step<W, C16>(a, S, wl, L.A, L.B, L.ca, L.cb, true, s0, s1, s2, pend, lane);
    else if (L.work) cell_process_big<W, C16>(a, S, wl, L, s0, s1, s2, pend, lane);
}

template <bool W, bool C16>
__device__ inline void join_cells(const JoinArgs& a, u32* S, WaveLds& wl, const u32 I, const u32 J, const int wv,
                                  const int lane, const u32 sub, const u32 sp) {
    const u32* cI = a.cidx + (size_t)I * (a.ncell + 1);
    const u32* cJ = a.cidx + (size_t)J * (a.ncell + 1);
    const u32 kI = a.blk_raw[I + 1] - a.blk_raw[I], kJ = a.blk_raw[J + 1] - a.blk_raw[J];
    const u32 kmax = max(max(kI, kJ), 1u);
    // fine cells per coarse cell: ~176 keys of the longer list (253 fit a window whatever its alignment)
    u32 m = (u32)(((u64)CELL_TARGET * a.ncell) / kmax);
    m = __builtin_amdgcn_readfirstlane(max(1u, min(m, a.ncell)));
    const u32 ncoarse = (a.ncell + m - 1) / m;
    // this workgroup's share of the coarse cells (all of them unless it is a tail split), cut into JW wave ranges
    const u32 wbeg = (u32)(((u64)ncoarse * sub) / sp), wend = (u32)(((u64)ncoarse * (sub + 1)) / sp);
    const u32 cbeg = wbeg + (u32)(((u64)(wend - wbeg) * wv) / JW), cend = wbeg + (u32)(((u64)(wend - wbeg) * (wv + 1)) / JW);
    // two pending sets, one per unrolled half: a step's posting words are consumed two steps later
    Pending pend0, pend1;
    pend0.ia = 0; pend0.ib = 0; pend0.w = 1; pend0.valid = false;
    pend1 = pend0;
    u32 s0 = 0, s1 = 0, s2 = 0;
    CellLoad L0, L1;
    cell_fetch(a, cI, cJ, cbeg, cend, m, lane, L0);
    cell_fetch(a, cI, cJ, cbeg + 1, cend, m, lane, L1);
    for (u32 c = cbeg; c < cend; c += 2) {
        cell_process<W, C16>(a, S, wl, L0, s0, s1, s2, pend0, lane);
        cell_fetch(a, cI, cJ, c + 2, cend, m, lane, L0);
        cell_process<W, C16>(a, S, wl, L1, s0, s1, s2, pend1, lane);
        cell_fetch(a, cI, cJ, c + 3, cend, m, lane, L1);
    }
    pending_apply<C16>(S, wl.lst, a.bigmask, pend0, lane);
    pending_apply<C16>(S, wl.lst, a.bigmask, pend1, lane);
}

// Compact the non-zero counters of one tile into (source_1, source_2, shared) records:
// ballot + popcount prefix inside the wave, one global atomic per wave for the output slot.
template <class Get>
__device__ inline void emit_tile(const JoinArgs& a, const u32 I, const u32 J, const int tid, const int lane, Get get) {
    const u32 gi0 = I * TB, gj0 = J * TB;
    for (int base = 0; base < TB * TB; base += JW * 64) {
        const int idx = base + tid;
        const u32 v = get(idx);
        const bool nz = v != 0;
        const unsigned long long mask = __ballot(nz);
        if (mask == 0) continue;
        unsigned long long wbase = 0;
        if (lane == 0) wbase = atomicAdd(a.out_count, (unsigned long long)__popcll(mask));
        wbase = __shfl(wbase, 0);
        if (nz) {
            const u64 pos = wbase + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < a.cap) {
                // back from the engine's source order to the caller's ids
                const u32 o1 = a.inv[gi0 + (u32)(idx / TB)], o2 = a.inv[gj0 + (u32)(idx % TB)];
                ksp_edge e;
                e.source_1 = min(o1, o2);
                e.source_2 = max(o1, o2);
                e.shared = v;
                a.out[pos] = e;
            }
        }
    }
}

// ---- diagonal tile of an unweighted block, bit-sliced ------------------------------------------
// With related sources in one block, a diagonal tile is a dense problem: counts = M^T M for the 0/1
// membership matrix M (keys x 128 sources).  Instead of one LDS atomic per pair update, the masks of
// 64 keys are transposed into bit columns (lane-parallel 64 x 64 bit transpose, 6 butterfly steps), a
// chunk of columns is staged in LDS as col[group][source] (64 keys per 64-bit word), and every thread
// owns a 4 x 4 patch of source pairs: per group 8 column words, 16 x popcount(a & b).  The 496
// patches above the diagonal go to threads 0..495; the 192 pairs inside the 32 diagonal patches go
// one each to threads 0..191.  Results leave as edges straight from registers.
__device__ inline u64 transpose64_step(u64 x, const int lane, const int j, const u64 m) {
    const u64 p = __shfl_xor(x, j);
    return (lane & j) == 0 ? ((x & m) | ((p & m) << j)) : (((p >> j) & m) | (x & (m << j)));
}
__device__ inline u64 transpose64(u64 x, const int lane) {   // bit b of lane r  <->  bit r of lane b
    x = transpose64_step(x, lane, 32, 0x00000000FFFFFFFFull);
    x = transpose64_step(x, lane, 16, 0x0000FFFF0000FFFFull);
    x = transpose64_step(x, lane, 8, 0x00FF00FF00FF00FFull);
    x = transpose64_step(x, lane, 4, 0x0F0F0F0F0F0F0F0Full);
    x = transpose64_step(x, lane, 2, 0x3333333333333333ull);
    x = transpose64_step(x, lane, 1, 0x5555555555555555ull);
    return x;
}
__device__ inline void emit_value(const JoinArgs& a, const u32 gi, const u32 gj, const u32 v, const int lane) {
    const bool nz = v != 0;
    const unsigned long long mask = __ballot(nz);
    if (mask == 0) return;
    unsigned long long wbase = 0;
    if (lane == 0) wbase = atomicAdd(a.out_count, (unsigned long long)__popcll(mask));
    wbase = __shfl(wbase, 0);
    if (nz) {
        const u64 pos = wbase + __popcll(mask & ((1ull << lane) - 1ull));
        if (pos < a.cap) {
            const u32 o1 = a.inv[gi], o2 = a.inv[gj];
            ksp_edge e;
            e.source_1 = min(o1, o2);
            e.source_2 = max(o1, o2);
            e.shared = v;
            a.out[pos] = e;
        }
    }
}
// A share (sub of sp) takes a range of the block's keys; with several shares the partial counts are
// added into the tile's global buffer `dst` (the caller's last-share logic emits them), otherwise the
// edges leave straight from the registers.
template <int SMEM_BYTES>
__device__ inline void self_tile_popc(const JoinArgs& a, unsigned char* smem, const u32 I, const u32 sub, const u32 sp,
                                      u32* __restrict__ dst, const int tid, const int lane, const int wv) {
    static_assert(TB == 128, "bit-sliced diagonal path: two 64-bit words per membership mask");
    constexpr u32 G = SMEM_BYTES / 1024;   // 64-key groups per chunk (1 KB = 128 columns x 8 B each)
    u64* col = reinterpret_cast<u64*>(smem);
    const u32 kall = a.blk_raw[I + 1] - a.blk_raw[I];
    const u32 kfirst = (u32)(((u64)kall * sub) / sp), klast = (u32)(((u64)kall * (sub + 1)) / sp);
    const u32 kb0 = a.blk_pos[I] + kfirst, klen = klast - kfirst;   // this share's keys
    // patches above the diagonal: (ti, tj), ti < tj < 32
    u32 ti = 0, tj = 1;
    const bool off = tid < 496;
    if (off) { tile_decode((u64)tid, 31, ti, tj); tj += 1; }
    // pairs inside the diagonal patches: patch d = tid / 6, pair q = tid % 6 of its 4 sources
    const bool dia = tid < 192;
    const u32 dq = (u32)tid % 6u, dblk = (u32)tid / 6u;
    const u32 da = dq < 3 ? 0u : dq < 5 ? 1u : 2u;                 // (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)
    const u32 db = dq < 3 ? dq + 1u : dq < 5 ? dq - 1u : 3u;
    const u32 s0 = 4u * dblk + da, s1 = 4u * dblk + db;
    u32 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    u32 dacc = 0;
    for (u32 base = 0; base < klen; base += G * 64) {
        const u32 ng = min(G, (klen - base + 63u) / 64u);
        // transpose 64 masks into 128 column words; a wave takes groups wv, wv + JW, ...  All posting
        // words, then all masks of the wave's groups are requested before the first is used (two
        // rounds of memory latency per chunk, not two per group)
        constexpr u32 GW = (G + JW - 1) / JW, GB = 3;   // groups per wave and chunk, in batches of GB (register budget)
        for (u32 q0 = 0; q0 < GW; q0 += GB) {
            u32 inf[GB];
            uint4 mk[GB];
#pragma unroll
            for (u32 q = 0; q < GB; ++q) {
                const u32 g = (u32)wv + (q0 + q) * JW, k = base + 64u * g + (u32)lane;
                inf[q] = (g < ng && k < klen) ? a.info[kb0 + k] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (u32 q = 0; q < GB; ++q) {
                mk[q] = make_uint4(0, 0, 0, 0);
                if (inf[q] != 0xFFFFFFFFu && inf[q] >= BIG) mk[q] = a.bigmask[inf[q] & ~BIG];
            }
#pragma unroll
            for (u32 q = 0; q < GB; ++q) {
                const u32 g = (u32)wv + (q0 + q) * JW;
                if (g < ng) {
                    uint4 m = mk[q];
                    if (inf[q] != 0xFFFFFFFFu && inf[q] < BIG) m = posting_mask(inf[q], a.bigmask);   // inline ids -> mask
                    const u64 lo = transpose64((u64)m.x | ((u64)m.y << 32), lane);
                    const u64 hi = transpose64((u64)m.z | ((u64)m.w << 32), lane);
                    col[g * 128u + (u32)lane] = lo;
                    col[g * 128u + 64u + (u32)lane] = hi;
                }
            }
        }
        __syncthreads();
        if (off) {
            for (u32 g = 0; g < ng; ++g) {
                const ulonglong2* r = reinterpret_cast<const ulonglong2*>(col + g * 128u + 4u * ti);
                const ulonglong2* c = reinterpret_cast<const ulonglong2*>(col + g * 128u + 4u * tj);
                const ulonglong2 r01 = r[0], r23 = r[1], c01 = c[0], c23 = c[1];
                const u64 rr[4] = {r01.x, r01.y, r23.x, r23.y};
                const u64 cc[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[4 * x + y] += (u32)__popcll(rr[x] & cc[y]);
            }
        }
        if (dia) {
            for (u32 g = 0; g < ng; ++g) dacc += (u32)__popcll(col[g * 128u + s0] & col[g * 128u + s1]);
        }
        __syncthreads();
    }
    if (dst) {   // one of several shares: partial counts into the tile's buffer
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (off && acc[4 * x + y]) atomicAdd(&dst[(4u * ti + (u32)x) * TB + 4u * tj + (u32)y], acc[4 * x + y]);
        if (dia && dacc) atomicAdd(&dst[s0 * TB + s1], dacc);
        return;
    }
    const u32 g0 = I * TB;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) emit_value(a, g0 + 4u * ti + (u32)x, g0 + 4u * tj + (u32)y, off ? acc[4 * x + y] : 0u, lane);
    emit_value(a, g0 + s0, g0 + s1, dia ? dacc : 0u, lane);
}

// ---- off-diagonal tile of unweighted blocks: collect the matches, accumulate them bit-sliced ------
// Expanding one match costs cA x cB counter updates — quadratic in the cluster size once related
// sources share a block.  Instead the workgroup works in rounds: every wave searches one step (a
// 256-key chunk of A against a 256-key window of B) and appends its matches (two list positions) to
// an LDS buffer; then, 512 matches at a time, the waves fetch the two membership masks of their
// matches, transpose them into bit columns (colA[g][source of I], colB[g][source of J], 64 matches
// per word) and every thread adds popcount(a & b) for its two 4 x 4 patches of the 128 x 128 tile.
// The cost per match no longer depends on the size of the postings.
constexpr u32 MCAP = 2048;    // matches per round: 8 waves x (at most 256 per step)
constexpr u32 MSUB = 512;     // matches per accumulation batch: 8 groups of 64, one per wave
struct CollectLds {
    uint2 match[MCAP];            // (position in list I, position in list J)
    u64 col[2][MSUB / 64][TB];    // [A / B][group][source]
};
// C16: the tile's counts stay below 2^16 (one block has no source with >= 2^16 k-mers), so two
// 16-bit counters share a register: 16 instead of 32 accumulator registers per thread.
template <int S_BYTES, bool C16>
__device__ inline void join_cells_collect(const JoinArgs& a, unsigned char* smem, WaveLds& wl, const u32 I, const u32 J,
                                          const u32 sub, const u32 sp, u32* __restrict__ dst, const int tid, const int lane,
                                          const int wv) {
    static_assert(sizeof(CollectLds) <= (size_t)S_BYTES, "collect buffers must fit the counter tile's LDS");
    static_assert(TB == 128 && JW == 8, "bit-sliced accumulation: 128 x 128 tile, 8 waves");
    CollectLds& cl = *reinterpret_cast<CollectLds*>(smem);
    __shared__ u32 s_n, s_more;
    const u32* cI = a.cidx + (size_t)I * (a.ncell + 1);
    const u32* cJ = a.cidx + (size_t)J * (a.ncell + 1);
    const u32 kI = a.blk_raw[I + 1] - a.blk_raw[I], kJ = a.blk_raw[J + 1] - a.blk_raw[J];
    const u32 kmax = max(max(kI, kJ), 1u);
    u32 m = (u32)(((u64)CELL_TARGET * a.ncell) / kmax);
    m = __builtin_amdgcn_readfirstlane(max(1u, min(m, a.ncell)));
    const u32 ncoarse = (a.ncell + m - 1) / m;
    const u32 wbeg = (u32)(((u64)ncoarse * sub) / sp), wend = (u32)(((u64)ncoarse * (sub + 1)) / sp);
    // this wave's cells: wbeg + wv, + JW, ...; inside a cell the steps (chunk x window) in order
    u32 c = wbeg + (u32)wv;
    u32 a0 = 0, a1 = 0, b0 = 0, b1 = 0, ca = 0, cb = 0;   // current cell / step
    bool have = false;
    auto open_cell = [&]() {
        have = false;
        while (c < wend) {
            const u32 f0 = c * m, f1 = min(a.ncell, f0 + m);
            a0 = __builtin_amdgcn_readfirstlane(cI[f0]); a1 = __builtin_amdgcn_readfirstlane(cI[f1]);
            b0 = __builtin_amdgcn_readfirstlane(cJ[f0]); b1 = __builtin_amdgcn_readfirstlane(cJ[f1]);
            if (a1 > a0 && b1 > b0) { ca = a0 & ~3u; cb = b0 & ~3u; have = true; return; }
            c += JW;
        }
    };
    open_cell();
    // the two 4 x 4 patches of this thread: rows 4 pi .. (block I), columns 4 pj .. (block J)
    const u32 pi0 = (u32)tid >> 5, pi1 = pi0 + 16u, pj = (u32)tid & 31u;
    constexpr int NA = C16 ? 8 : 16;   // C16: acc[k] = pairs (x, 2k) and (x, 2k + 1) ... see below
    u32 acc0[NA], acc1[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { acc0[i] = 0; acc1[i] = 0; }
    u32 s0 = 0, s1 = 0, s2 = 0;
    while (true) {
        __syncthreads();   // (everyone has read the previous round's s_n / s_more)
        if (tid == 0) { s_n = 0; s_more = 0; }
        __syncthreads();
        if (have) {
            // one step: chunk [ca, ca + 256) of the cell's A keys against window [cb, cb + 256) of its B keys
            const uint4 A = load_a(a, ca, a0, a1, lane);
            const uint4 B = load_b(a, cb, b1, lane);
            Window& wn = wl.win;
            u32* l2w = reinterpret_cast<u32*>(wn.l2);
            u32* l1w = reinterpret_cast<u32*>(wn.l1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            wn.leaf[lane] = B;
            l2w[lane] = B.w;
            if ((lane & 3) == 3) l1w[lane >> 2] = B.w;
            s0 = __builtin_amdgcn_readlane(B.w, 15);
            s1 = __builtin_amdgcn_readlane(B.w, 31);
            s2 = __builtin_amdgcn_readlane(B.w, 47);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            bool h0, h1, h2, h3;
            const u32 p0 = window_find(wn, s0, s1, s2, A.x, h0);
            const u32 p1 = window_find(wn, s0, s1, s2, A.y, h1);
            const u32 p2 = window_find(wn, s0, s1, s2, A.z, h2);
            const u32 p3 = window_find(wn, s0, s1, s2, A.w, h3);
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            const u32 n0 = (u32)__popcll(m0), n1 = (u32)__popcll(m1), n2 = (u32)__popcll(m2), n3 = (u32)__popcll(m3);
            const u32 cnt = n0 + n1 + n2 + n3;
            u32 base = 0;
            if (lane == 0 && cnt) base = atomicAdd(&s_n, cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            const u32 qa = ca + 4u * (u32)lane;
            if (h0) cl.match[base + __builtin_amdgcn_mbcnt_hi((u32)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m0, 0))] = make_uint2(qa, cb + p0);
            base += n0;
            if (h1) cl.match[base + __builtin_amdgcn_mbcnt_hi((u32)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m1, 0))] = make_uint2(qa + 1, cb + p1);
            base += n1;
            if (h2) cl.match[base + __builtin_amdgcn_mbcnt_hi((u32)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m2, 0))] = make_uint2(qa + 2, cb + p2);
            base += n2;
            if (h3) cl.match[base + __builtin_amdgcn_mbcnt_hi((u32)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m3, 0))] = make_uint2(qa + 3, cb + p3);
            // next step of this wave: next A chunk of the window, next window, next cell
            ca += WIN;
            if (ca >= a1) {
                ca = a0 & ~3u;
                cb += WIN;
                if (cb >= b1) { c += JW; open_cell(); }
            }
            if (have && lane == 0) s_more = 1;
        }
        __syncthreads();
        const u32 n = s_n;
        const bool more = s_more != 0;
        for (u32 mb = 0; mb < n; mb += MSUB) {
            // masks of 64 matches per wave -> bit columns
            const u32 mi = mb + 64u * (u32)wv + (u32)lane;
            uint4 ma = make_uint4(0, 0, 0, 0), mbm = ma;
            if (mi < n) {
                const uint2 q = cl.match[mi];
                const u32 ia = a.info[q.x], ib = a.info[q.y];
                ma = posting_mask(ia, a.bigmask);
                mbm = posting_mask(ib, a.bigmask);
            }
            if (mb + 64u * (u32)wv < n) {   // wave-uniform; one transpose at a time keeps the register count down
                cl.col[0][wv][lane] = transpose64((u64)ma.x | ((u64)ma.y << 32), lane);
                __builtin_amdgcn_sched_barrier(0);
                cl.col[0][wv][64 + lane] = transpose64((u64)ma.z | ((u64)ma.w << 32), lane);
                __builtin_amdgcn_sched_barrier(0);
                cl.col[1][wv][lane] = transpose64((u64)mbm.x | ((u64)mbm.y << 32), lane);
                __builtin_amdgcn_sched_barrier(0);
                cl.col[1][wv][64 + lane] = transpose64((u64)mbm.z | ((u64)mbm.w << 32), lane);
            }
            __syncthreads();
            const u32 ng = min((u32)(MSUB / 64), (n - mb + 63u) / 64u);
#pragma unroll 1
            for (u32 g = 0; g < ng; ++g) {
                const ulonglong2* cc = reinterpret_cast<const ulonglong2*>(&cl.col[1][g][4u * pj]);
                const ulonglong2 c01 = cc[0], c23 = cc[1];
                const u64 cw[4] = {c01.x, c01.y, c23.x, c23.y};
                {
                    const ulonglong2* r0 = reinterpret_cast<const ulonglong2*>(&cl.col[0][g][4u * pi0]);
                    const ulonglong2 x01 = r0[0], x23 = r0[1];
                    const u64 rw[4] = {x01.x, x01.y, x23.x, x23.y};
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; y += 2) {
                            const u32 v0 = (u32)__popcll(rw[x] & cw[y]), v1 = (u32)__popcll(rw[x] & cw[y + 1]);
                            if (C16) acc0[2 * x + y / 2] += v0 | (v1 << 16);
                            else { acc0[4 * x + y] += v0; acc0[4 * x + y + 1] += v1; }
                        }
                }
                {
                    const ulonglong2* r1 = reinterpret_cast<const ulonglong2*>(&cl.col[0][g][4u * pi1]);
                    const ulonglong2 y01 = r1[0], y23 = r1[1];
                    const u64 rw[4] = {y01.x, y01.y, y23.x, y23.y};
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; y += 2) {
                            const u32 v0 = (u32)__popcll(rw[x] & cw[y]), v1 = (u32)__popcll(rw[x] & cw[y + 1]);
                            if (C16) acc1[2 * x + y / 2] += v0 | (v1 << 16);
                            else { acc1[4 * x + y] += v0; acc1[4 * x + y + 1] += v1; }
                        }
                }
            }
            __syncthreads();
        }
        if (!more) break;
    }
    // results: partial counts into the tile's buffer (one of several shares) or straight to edges
    const u32 gi = I * TB, gj = J * TB;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const u32 r0 = 4u * pi0 + (u32)x, r1 = 4u * pi1 + (u32)x, cc = 4u * pj + (u32)y;
            const u32 v0 = C16 ? (acc0[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc0[C16 ? 0 : 4 * x + y];
            const u32 v1 = C16 ? (acc1[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc1[C16 ? 0 : 4 * x + y];
            if (dst) {
                if (v0) atomicAdd(&dst[r0 * TB + cc], v0);
                if (v1) atomicAdd(&dst[r1 * TB + cc], v1);
            } else {
                emit_value(a, gi + r0, gj + cc, v0, lane);
                emit_value(a, gi + r1, gj + cc, v1, lane);
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps hipcc from hoisting all 64 id look-ups: registers)
        }
}

template <bool W, bool C16, bool CELLS>
__global__ __launch_bounds__(JW * 64, 6) void k_join(JoinArgs a) {   // (6 waves per SIMD = three workgroups per CU: caps the unweighted variant at 80 VGPRs)
    // pair counters (32 KB packed 16-bit / 64 KB 32-bit) + 8 x (1.3 KB B window + 0.5 KB match
    // queue): three (C16) or two workgroups per CU
    constexpr int S_BYTES = (C16 ? TB * TB / 2 : TB * TB) * 4;
    constexpr int SMEM_BYTES = S_BYTES + (int)sizeof(WaveLds) * JW;
    __shared__ __align__(16) unsigned char smem[SMEM_BYTES];   // (the bit-sliced diagonal path uses all of it as one buffer)
    u32* S = reinterpret_cast<u32*>(smem);
    WaveLds* wlds = reinterpret_cast<WaveLds*>(smem + S_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps index arithmetic and loads scalar
    // Tail splitting: the tiles of the last, partially filled round of workgroup slots are cut
    // into tail_sp rank-range shares each, so that the round takes 1/tail_sp of a tile time.
    u32 sub = 0, sp = 1, tail_id = 0xFFFFFFFFu;
    u32 I, J;
    if (a.sched) {
        const u32 wg = a.wg0 + blockIdx.x;
        const u32* t = a.act + 4 * (size_t)a.sched[wg];
        I = t[0]; J = t[1];
        sub = wg - t[2];
        sp = t[6] - t[2];                       // (next tile's first workgroup)
        if (sp > 1) tail_id = t[3] - a.split0;
    } else {
        u64 tile = a.tile_begin + blockIdx.x;
        if (blockIdx.x >= a.n_normal) {
            const u32 r = blockIdx.x - a.n_normal;
            tail_id = r / a.tail_sp;
            sub = r % a.tail_sp;
            sp = a.tail_sp;
            tile = a.tile_begin + a.n_normal + tail_id;
        }
        tile_decode(tile, a.nb, I, J);
        if (I == J && sub != 0) return;   // (dense mode) a diagonal tail tile is done by its first share alone
        if (I == J) sp = 1;
    }
    // 16-bit counters are exact iff one of the two blocks has no source with >= 2^16 k-mers
    if ((min(a.blk_max[I], a.blk_max[J]) < 65536u) != C16) return;

    const bool popc = !W && a.collect && (I == J || CELLS);
    if (popc) {
        // unweighted tiles: bit-sliced accumulation in registers (no counter tile, no LDS atomics)
        u32* dst = tail_id != 0xFFFFFFFFu ? a.tailbuf + (size_t)tail_id * (TB * TB) : nullptr;
        if (I == J) self_tile_popc<SMEM_BYTES>(a, smem, I, sub, sp, dst, tid, lane, wv);
        else join_cells_collect<S_BYTES, C16>(a, smem, wlds[wv], I, J, sub, sp, dst, tid, lane, wv);
        if (!dst) return;
    } else {
    for (int i = tid; i < (C16 ? TB * TB / 2 : TB * TB); i += JW * 64) S[i] = 0;
    __syncthreads();

    if (I == J) {
        // self tile: every key of the block matches itself; only keys held by >= 2 sources
        // produce pairs (counted in the upper triangle: posting ids follow the caller's source order,
        // not the engine's)
        const u32 kb0 = a.blk_pos[I], klen = a.blk_raw[I + 1] - a.blk_raw[I];
        const u32 kb = kb0 + (u32)(((u64)klen * sub) / sp), ke = kb0 + (u32)(((u64)klen * (sub + 1)) / sp);   // this share's keys
        for (u32 k0 = kb; k0 < ke; k0 += JW * 64) {   // uniform trip count: the big path is wave-wide
            const u32 k = k0 + tid;
            u32 inf = 0;
            u32 w = 1;
            if (k < ke) { inf = a.info[k]; if (W) w = a.bw[k]; }
            const bool big = inf >= BIG;
            if (!big) {
                const u32 n = (inf >> 29) + 1;
                for (u32 x = 0; x + 1 < n; ++x) {
                    const u32 ix = (inf >> (7 * x)) & 127u;
                    for (u32 y = x + 1; y < n; ++y) {
                        const u32 iy = (inf >> (7 * y)) & 127u;
                        s_add<C16>(S, min(ix, iy) * TB + max(ix, iy), w);
                    }
                }
            }
            unsigned long long todo = __ballot(big);
            uint4 mk = make_uint4(0, 0, 0, 0);
            if (big) mk = a.bigmask[inf & ~BIG];   // all masks of the wave in one round of memory latency
            while (todo) {
                const int src = __builtin_ctzll(todo);
                todo &= todo - 1;
                const u32 sw = __builtin_amdgcn_readlane(w, src);
                const uint4 m = make_uint4(__builtin_amdgcn_readlane(mk.x, src), __builtin_amdgcn_readlane(mk.y, src),
                                           __builtin_amdgcn_readlane(mk.z, src), __builtin_amdgcn_readlane(mk.w, src));
                add_masks<true, C16>(S, wlds[wv].lst, m, m, sw, lane);
            }
        }
    } else if (CELLS) {
        join_cells<W, C16>(a, S, wlds[wv], I, J, wv, lane, sub, sp);
    } else {
        join_windows<W, C16>(a, S, wlds[wv], I, J, wv, lane, sub, sp);
    }
    }   // !popc
    __syncthreads();

    if (tail_id != 0xFFFFFFFFu) {
        // tail share: add the partial counters into the tile's global buffer (k_tail_emit compacts it)
        u32* dst = a.tailbuf + (size_t)tail_id * (TB * TB);
        for (int base = 0; !popc && base < TB * TB; base += JW * 64) {
            const int idx = base + tid;
            const u32 v = C16 ? ((S[idx >> 1] >> ((idx & 1) * 16)) & 0xFFFFu) : S[idx];
            if (v) atomicAdd(&dst[idx], v);
        }
        if (!a.tail_done) return;   // (dense mode: k_tail_emit compacts the buffer)
        // work-list mode: the share that finishes last turns the summed counters into edges
        __shared__ u32 s_last;
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&a.tail_done[tail_id], 1u) == sp - 1 ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        emit_tile(a, I, J, tid, lane, [&](int idx) { return __hip_atomic_load(&dst[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
        return;
    }
    // flush: compact the non-zero counters of the tile into edges
    emit_tile(a, I, J, tid, lane, [&](int idx) { return C16 ? ((S[idx >> 1] >> ((idx & 1) * 16)) & 0xFFFFu) : S[idx]; });
}

// one workgroup per tail tile: its summed counters -> edges
__global__ __launch_bounds__(JW * 64) void k_tail_emit(JoinArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    u32 I, J;
    tile_decode(a.tile_begin + a.n_normal + blockIdx.x, a.nb, I, J);
    const u32* src = a.tailbuf + (size_t)blockIdx.x * (TB * TB);
    emit_tile(a, I, J, tid, lane, [&](int idx) { return src[idx]; });
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
    u64 n_kept = 0;               // entries whose key is held by >= 2 sources (the others are pruned)
    bool weighted = false;
    bool built = false;
    u32 nparts = 1, part_id = 0;  // key-range slice mode (multi-GPU build)
    u64 max_key = 0;
    bool have_max_key = false;
    bool slice_ready = false;
    const u32 *post_off = nullptr, *post_src = nullptr, *post_w = nullptr;   // postings input of the build in progress (device)
    u32 post_nkeys = 0;
    int slice_phase = 0;          // 1: build_slice done, waiting for ksp_engine_slice_finish
    u64 slice_hdr[4] = {0, 0, 0, 0};   // padded length, distinct keys (U), big postings, block keys
    u32 ncell = ksp::NP;          // fine rank cells per block (power of two)
    bool use_cells = true;        // rank-aligned cell join (KSP_JOIN=window selects the sliding-window merge)
    bool full_sort = false;       // keys defeat the 32-bit prefix sort: use all bits
    bool reorder = true;          // order the sources by shared-key label before cutting blocks (KSP_REORDER=0: off)
    bool need32 = false;          // some tile pairs two blocks that both hold a source with >= 2^16 k-mers
    int key_bits = 64;
    std::vector<u64> h_off;
    std::vector<u32> h_blk_off;   // distinct-key offsets of the block lists (host copy)
    std::vector<u32> h_blk_max;   // per block: largest per-source k-mer count / weight sum
    // workspace
    ksp::Buf d_off, KA, KB, VA, VB, R1, FK, FT, asm_small, tmp, bkeys, info, bw, mm, blk_raw, blk_pos, blk_max, part, scalars, count, tailbuf, smap;
    u32 slots = 0;                // workgroups of k_join the chip holds at once (occupancy x CUs)
    // work list of the join (built by finish_build; empty -> dense mode: every tile is visited)
    bool collect = false;         // unweighted off-diagonal tiles: collect matches + bit-sliced accumulation (multi-source
                                  // postings dominate) instead of LDS counters (single-source postings dominate)
    bool sched_on = false;
    hipStream_t sched_stream = nullptr;   // stream of the build that produced the work list (its uploads are ordered on it)
    bool have_bits = false;       // tbits / dwork hold this build's tile bitmap and diagonal work
    std::vector<u64> act_tid;     // active tiles (row-major tile ids, ascending)
    std::vector<u32> act_rec;     // per active tile: I, J, first workgroup, split index (+ one sentinel record)
    ksp::Buf tbits, dwork, d_act, d_wg;
    unsigned char* h_stage = nullptr;   // pinned: diagonal work + overflow flag, then the tile bitmap
    size_t h_stage_bytes = 0;
    std::vector<u32> wg_host;           // share -> active tile (kept alive for the asynchronous upload)
    unsigned long long* h_count = nullptr;   // pinned
    u64* h_scal = nullptr;                   // pinned: [0] max key, [1] Ktot, [2] U
    u64 sort_entries = 0;                    // entries / key bits of the last global radix sort (stats)
    int sort_bits = 0;
    u64 h_scal_words = 0, h_scal_keys = 0;   // list words / distinct shared keys of the lists being finished
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // build, join, first radix sort
    ksp_stats st{};
};

namespace ksp {

static inline unsigned grid_for(u64 n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

template <bool W>
static int build_impl(ksp_engine* e, const u64* d_keys, const u32* d_w, hipStream_t st, const int phase) {
    // phase 0: the whole of stage 1;  1: up to the source labels (key-range slice, before the labels of all
    // slices are combined);  2: the rest (source order from the final labels, block lists);  3: postings
    // input (ksp_engine_build_postings: sorting and pruning are already done by the caller's inverted index)
    typedef typename std::conditional<W, u64, u32>::type V;
    const u64 n = e->n_entries;
    const u32 N = e->n_sources, nb = e->nb;
    const u64 lmax = n + (u64)nb * (WIN + 4) + 4 * WIN;   // upper bound of the padded layout (+ read slack)
    int rc;
    if ((rc = e->KA.ensure((n + 4) * 8))) return rc;
    if ((rc = e->KB.ensure((n + 4) * 8))) return rc;
    if ((rc = e->VA.ensure((n + 4) * sizeof(V)))) return rc;
    if ((rc = e->VB.ensure((n + 4) * sizeof(V)))) return rc;
    if ((rc = e->bkeys.ensure(lmax * 4))) return rc;
    if ((rc = e->info.ensure(lmax * 4))) return rc;
    if (W && (rc = e->bw.ensure(lmax * 4))) return rc;
    if ((rc = e->mm.ensure((n / (INLINE_MAX + 1) + 16) * 16))) return rc;   // 128-bit masks of big postings
    if ((rc = e->blk_raw.ensure(((size_t)nb + 2) * 4))) return rc;
    if ((rc = e->blk_pos.ensure(((size_t)nb + 2) * 4))) return rc;
    if ((rc = e->part.ensure(((size_t)nb + 1) * ((size_t)e->ncell + 1) * 4))) return rc;
    if ((rc = e->blk_max.ensure(((size_t)nb + 2) * 4))) return rc;
    if ((rc = e->scalars.ensure(128))) return rc;
    if ((rc = e->R1.ensure((n + 4) * 4))) return rc;

    u64* KA = e->KA.as<u64>();
    V* VA = e->VA.as<V>();
    V* VB = e->VB.as<V>();
    u64* d_off = e->d_off.as<u64>();
    u64* scal = e->scalars.as<u64>();   // [0] max key, [1] Ktot, [2] U, [3] padded length, [4] overflow, [5] fix count,
                                        // [6] kept entries, [7] big postings, [8] entries of the slice
    u32* blk_raw = e->blk_raw.as<u32>();
    u32* blk_pos = e->blk_pos.as<u32>();
    const unsigned bs = 256;
    u32* rank1 = e->R1.as<u32>();
    const size_t NN = ((size_t)N + 64) & ~(size_t)63;
    if ((rc = e->smap.ensure(6 * NN * 4))) return rc;
    // per-source maps: [0] label, [1] iota, [2] sorted labels, [3] order (= engine index -> source id),
    // [4] newidx (source id -> engine index), [5] bound of a source's pair counters
    u32* sm = e->smap.as<u32>();
    u32 *label = sm, *iota = sm + NN, *labs = sm + 2 * NN, *order = sm + 3 * NN, *newidx = sm + 4 * NN, *sbound = sm + 5 * NN;
    const bool reorder = e->reorder;
    int bbits = 1;
    while ((1u << bbits) < nb) ++bbits;
    size_t tb = 0;
    u64 m = e->n_kept;   // (phase 2: set by phase 1)
    if (phase == 3) {
        m = n;
        e->n_kept = m;
        hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, iota, N);
        hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, order, N);
        hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, newidx, N);
        hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, label, N);
        KSP_HIP(hipMemsetAsync(e->blk_max.p, 0, ((size_t)nb + 1) * 4, st));
        KSP_HIP(hipMemsetAsync(sbound, 0, (size_t)N * 4, st));
        const u32 nk = e->post_nkeys;
        KSP_HIP(hipMemsetAsync(scal + 4, 0, 8, st));
        hipLaunchKernelGGL((k_post_expand<V, W>), dim3(grid_for(nk, bs)), dim3(bs), 0, st, e->post_off, e->post_src,
                           e->post_w, VA, rank1, sbound, nk, N, (u32*)(scal + 4));
        {   // U = number of keys (what the prune scan reports on the sketch path)
            e->h_scal[2] = nk;
            KSP_HIP(hipMemcpyAsync(scal + 2, e->h_scal + 2, 8, hipMemcpyHostToDevice, st));
        }
        if (reorder) {
            const u32 skip = m / std::max<u32>(1, N) >= 512 ? 7u : 0u;
            hipLaunchKernelGGL((k_label<V>), dim3(grid_for(m, bs)), dim3(bs), 0, st, rank1, VA, e->post_off, label, skip, m);
        } else {
            hipLaunchKernelGGL(k_blk_bound, dim3(grid_for(N, bs)), dim3(bs), 0, st, sbound, newidx, e->blk_max.as<u32>(), N);
        }
    } else if (phase != 2) {

    // key range (one 8-byte D2H, unless the caller passed key_bits)
    if (e->key_bits <= 0) {
        KSP_HIP(hipMemsetAsync(scal, 0, 8, st));
        hipLaunchKernelGGL(k_max_last, dim3(grid_for(N, bs)), dim3(bs), 0, st, d_keys, d_off, (unsigned long long*)scal, N);
        KSP_HIP(hipMemcpyAsync(e->h_scal, scal, 8, hipMemcpyDeviceToHost, st));
        KSP_HIP(hipStreamSynchronize(st));
        u64 mx = e->h_scal[0];
        e->max_key = mx;
        e->have_max_key = true;
        int bits = 1;
        while (bits < 64 && (mx >> bits)) ++bits;
        e->key_bits = bits;
    }
    const int kbits = e->key_bits;
    if (W || e->nparts == 1)   // (weighted slices still need the per-source weight sums of all entries)
        hipLaunchKernelGGL((k_tag<W>), dim3(N), dim3(256), 0, st, d_off, d_w, W ? nullptr : (u32*)VA,
                           W ? (u64*)VA : nullptr, sbound);
    if (!W) hipLaunchKernelGGL(k_src_size, dim3(grid_for(N, bs)), dim3(bs), 0, st, d_off, sbound, N);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, iota, N);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, order, N);    // identity until the labels are known
    hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, newidx, N);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(N, bs)), dim3(bs), 0, st, label, N);
    KSP_HIP(hipMemsetAsync(e->blk_max.p, 0, ((size_t)nb + 1) * 4, st));
    if (!reorder) hipLaunchKernelGGL(k_blk_bound, dim3(grid_for(N, bs)), dim3(bs), 0, st, sbound, newidx, e->blk_max.as<u32>(), N);
    // slice mode (multi-GPU build): keep only the entries of this part's key range — one contiguous
    // sub-run per (sorted) source, so the cost is proportional to the slice, not to the sketch set
    const u64* keys_in = d_keys;
    const V* tags_in = VA;
    u64 nw = n;
    if (e->nparts > 1) {
        // equal shares of [0, largest key] (or of [0, 2^key_bits) when the caller fixed key_bits)
        const unsigned __int128 span = e->have_max_key ? (unsigned __int128)e->max_key + 1
                                       : kbits >= 64   ? ((unsigned __int128)1 << 64)
                                                       : ((unsigned __int128)1 << kbits);
        const u64 lo = (u64)((span * e->part_id) / e->nparts);
        const u64 hi = (u64)((span * (e->part_id + 1)) / e->nparts - 1);
        u32* first = (u32*)e->KB.p;                 // N
        u32* cnt = (u32*)e->KB.p + (N + 2);         // N
        u32* fpos = (u32*)e->KB.p + 2 * ((size_t)N + 2);   // N   (KB holds 2(n+4) u32 >= 3(N+2) whenever n >= 2N; checked below)
        if ((rc = e->KB.ensure(std::max<size_t>((n + 4) * 8, 3 * ((size_t)N + 2) * 4)))) return rc;
        first = (u32*)e->KB.p; cnt = first + (N + 2); fpos = first + 2 * ((size_t)N + 2);
        hipLaunchKernelGGL(k_range_bounds, dim3(grid_for(N, bs)), dim3(bs), 0, st, d_keys, d_off, lo, hi, first, cnt, N);
        tb = 0;
        KSP_HIP(rocprim::exclusive_scan(nullptr, tb, cnt, fpos, (u32)0, (size_t)N, rocprim::plus<u32>(), st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb, cnt, fpos, (u32)0, (size_t)N, rocprim::plus<u32>(), st));
        hipLaunchKernelGGL(k_range_total, dim3(1), dim3(64), 0, st, fpos, cnt, scal, N);
        KSP_HIP(hipMemcpyAsync(e->h_scal + 8, scal + 8, 8, hipMemcpyDeviceToHost, st));
        KSP_HIP(hipStreamSynchronize(st));
        nw = e->h_scal[8];
        if ((rc = e->FK.ensure((nw + 4) * 8))) return rc;
        if ((rc = e->FT.ensure((nw + 4) * sizeof(V)))) return rc;
        hipLaunchKernelGGL((k_range_copy<V, W>), dim3(N), dim3(128), 0, st, d_keys, d_w, d_off, first, cnt, fpos,
                           e->FK.as<u64>(), e->FT.as<V>());
        keys_in = e->FK.as<u64>();
        tags_in = e->FT.as<V>();
    }
    e->n_kept = 0;
    m = 0;
    if (nw == 0) return KSP_OK;   // (slice mode only) no key of this range: labels stay the identity
    // sort 1: all entries by the top 32 significant key bits (payload = tag [+weight]):
    // d_keys,VA -> KA,VB; then order the rare mixed runs by the full key (k_fix_runs)
    // (rocPRIM 4.2 / ROCm 7.2 mis-sorts 64-bit keys on any bit range [b > 0, 64) below ~1M items —
    //  found with a stand-alone sweep on MI355X; ranges ending below bit 64 are fine — so keys
    //  that use all 64 bits take the full-width sort.)
    const int shift = (e->full_sort || kbits >= 64) ? 0 : std::max(0, kbits - 32);
    u32* d_ovf = (u32*)(scal + 4);   // set by k_fix_runs when a run is too long; checked at the end of the build
    KSP_HIP(hipMemsetAsync(d_ovf, 0, 8, st));
    tb = 0;
    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, KA, tags_in, VB, nw, shift, kbits, st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(hipEventRecord(e->ev[4], st));
    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, keys_in, KA, tags_in, VB, nw, shift, kbits, st));
    KSP_HIP(hipEventRecord(e->ev[5], st));
    e->sort_entries = nw;
    e->sort_bits = kbits - shift;
    if (shift > 0) {
        // KB is free until the rank scan: use it for the work list of mixed runs
        u32* fixlist = (u32*)e->KB.p;
        const u32 fixcap = (u32)std::min<u64>(2 * nw, 0x7FFFFFFFull);   // positions are < 2^30 (DROP bit is free)
        u32* d_cnt = (u32*)(scal + 5);
        KSP_HIP(hipMemsetAsync(d_cnt, 0, 8, st));
        hipLaunchKernelGGL(k_find_mixed, dim3(grid_for(nw, 4096)), dim3(1024), 0, st, KA, nw, shift, fixlist, d_cnt,
                           fixcap, d_ovf);
        hipLaunchKernelGGL(k_mark_first, dim3(4096), dim3(64), 0, st, KA, shift, fixlist, d_cnt, fixcap, d_ovf);
        hipLaunchKernelGGL((k_fix_runs<V>), dim3(2048), dim3(256), 0, st, KA, VB, nw, shift, fixlist, d_cnt, fixcap,
                           d_ovf);
    }
    // singleton pruning + dense ranks of the kept keys (packed counters, one scan):  KA,VB -> R1 (ranks), VA (tags)
    // (the scan's output iterator scatters entry e as soon as its prefix sums are known: no second pass)
    if ((rc = e->FK.ensure((nw / 2 + 8) * 4))) return rc;
    u32* first = (u32*)e->FK.p;            // first kept entry of every rank (FK: the slice's input keys are dead after sort 1)
    {
        auto pf = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u64>(0), PruneFn{KA, nw});
        PruneScatterIt<V> out{{KA, VB, VA, rank1, first, scal, nw}, 0};
        tb = 0;
        KSP_HIP(rocprim::inclusive_scan(nullptr, tb, pf, out, nw, rocprim::plus<u64>(), st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::inclusive_scan(e->tmp.p, tb, pf, out, nw, rocprim::plus<u64>(), st));
    }
    KSP_HIP(hipMemcpyAsync(e->h_scal + 6, scal + 6, 8, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipStreamSynchronize(st));   // the kept-entry count sizes every later pass
    m = e->h_scal[6];
    e->n_kept = m;
    if (m == 0 && phase == 0) return KSP_OK;   // no key is shared by two sources: no pair at all
    if (reorder && m) {
        // label = smallest source id among the holders of a source's shared keys
        // (sources with hundreds of shared keys: every 8th key says as much about a source's relatives as all)
        const u32 skip = m / std::max<u32>(1, N) >= 512 ? 7u : 0u;
        hipLaunchKernelGGL((k_label<V>), dim3(grid_for(m, bs)), dim3(bs), 0, st, rank1, VA, first, label, skip, m);
    }
    if (phase == 1) return KSP_OK;
    }   // phase != 2
    if (reorder) {
        // order the sources by (label, id) and move the kept entries to the new indices
        int lbits = 1;
        while (lbits < 32 && (N >> lbits)) ++lbits;
        tb = 0;
        KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, label, labs, iota, order, (size_t)N, 0, lbits, st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, label, labs, iota, order, (size_t)N, 0, lbits, st));
        hipLaunchKernelGGL(k_perm, dim3(grid_for(N, bs)), dim3(bs), 0, st, order, newidx, N);
        if (m) hipLaunchKernelGGL((k_retag<V>), dim3(grid_for(m, bs)), dim3(bs), 0, st, VA, newidx, m);
        hipLaunchKernelGGL(k_blk_bound, dim3(grid_for(N, bs)), dim3(bs), 0, st, sbound, newidx, e->blk_max.as<u32>(), N);
    }
    if (m == 0) return KSP_OK;
    // sort 2: stable by block id (bits [8, 8+bbits) of the tag), payload = rank:  VA,rank1 -> VB,rk2
    u32* rk2 = (u32*)KA;                   // KA (sorted keys) is dead from here on
    tb = 0;
    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, VA, VB, rank1, rk2, m, 8, 8 + bbits, st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, VA, VB, rank1, rk2, m, 8, 8 + bbits, st));
    // now: rk2 = ranks sorted by (block, rank); VB = tags in the same order.  KB, VA, R1 are free.
    V* T = VB;
    u32* flag = (u32*)VA;
    u32* didx = (u32*)e->KB.p;             // m
    u32* estart = (u32*)e->KB.p + (n + 2); // up to m+1
    const HeadFn<V> head{rk2, T};
    auto hf = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u64>(0), head);
    tb = 0;
    KSP_HIP(rocprim::exclusive_scan(nullptr, tb, hf, didx, (u32)0, m, rocprim::plus<u32>(), st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb, hf, didx, (u32)0, m, rocprim::plus<u32>(), st));
    hipLaunchKernelGGL((k_ktot<V>), dim3(1), dim3(64), 0, st, head, didx, estart, scal, m);
    hipLaunchKernelGGL((k_blk_raw<V>), dim3(grid_for((u64)nb + 1, bs)), dim3(bs), 0, st, T, didx, scal, blk_raw, nb, m);
    hipLaunchKernelGGL(k_blk_pos, dim3(1), dim3(64), 0, st, blk_raw, blk_pos, scal, nb);
    hipLaunchKernelGGL(k_pad, dim3(nb + 1), dim3(256), 0, st, blk_raw, blk_pos, e->bkeys.as<u32>(), nb, PAD);
    hipLaunchKernelGGL((k_emit_keys<V>), dim3(grid_for(m, bs)), dim3(bs), 0, st, rk2, T, head, didx, blk_raw,
                       blk_pos, e->bkeys.as<u32>(), estart, m);
    // the number of distinct (block, key) groups sizes the posting passes (after the source reordering it is
    // an order of magnitude below the entry count: one 8-byte read-back pays for itself)
    KSP_HIP(hipMemcpyAsync(e->h_scal + 1, scal + 1, 8, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipStreamSynchronize(st));
    const u64 K = std::max<u64>(1, e->h_scal[1]);
    u32* mmsz = flag;                      // (VA is free: the head flags are computed on the fly)
    u32* mmoff = (u32*)KA;                 // rk2 is dead after k_emit_keys
    hipLaunchKernelGGL(k_bigflag, dim3(grid_for(K, bs)), dim3(bs), 0, st, estart, scal, mmsz, K);
    tb = 0;
    KSP_HIP(rocprim::exclusive_scan(nullptr, tb, mmsz, mmoff, (u32)0, K, rocprim::plus<u32>(), st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb, mmsz, mmoff, (u32)0, K, rocprim::plus<u32>(), st));
    hipLaunchKernelGGL(k_nbig, dim3(1), dim3(64), 0, st, mmsz, mmoff, scal);
    hipLaunchKernelGGL((k_emit_info<V, W>), dim3(grid_for(K, bs)), dim3(bs), 0, st, estart, mmoff, scal, T,
                       blk_raw, blk_pos, e->info.as<u32>(), e->mm.as<uint4>(), W ? e->bw.as<u32>() : nullptr);
    // rank-range partition of every block list
    hipLaunchKernelGGL(k_cidx, dim3(grid_for((u64)nb * (e->ncell + 1), bs)), dim3(bs), 0, st, e->bkeys.as<u32>(),
                       blk_raw, blk_pos, scal, e->part.as<u32>(), nb, e->ncell);
    KSP_HIP(hipGetLastError());
    return KSP_OK;
}

// Last step of stage 1 (single build and assemble alike): the bitmap of block pairs that share a
// key and the pair-update count of every diagonal tile; finish_build turns them into the work list.
static int launch_sched_kernels(ksp_engine* e, hipStream_t st) {
    e->have_bits = false;
    e->sched_stream = st;
    const u32 nb = e->nb;
    const u64 K = e->h_scal_words;   // list words (set by the caller)
    const u64 U = e->h_scal_keys;
    const u64 T = (u64)nb * (nb + 1) / 2;
    if (std::getenv("KSP_NO_SCHED") || K == 0 || U == 0) return KSP_OK;
    // a key in many blocks means most block pairs are active anyway (and the pair walk below is
    // quadratic in the blocks per key): leave such inputs to the dense mode
    if (K > 6 * U || T > (1ull << 26)) return KSP_OK;
    int rc;
    const size_t bit_words = (size_t)(((T + 63) / 64) * 2 + 2);
    if ((rc = e->tbits.ensure(bit_words * 4 + T + 64))) return rc;   // packed bitmap, then one flag byte per tile
    if ((rc = e->dwork.ensure(((size_t)nb + 2) * 8))) return rc;
    if ((rc = e->KA.ensure((K + 4) * 8))) return rc;
    if ((rc = e->KB.ensure((K + 4) * 8))) return rc;
    u32 *pr = (u32*)e->KA.p, *pb = pr + (K + 4), *pr2 = (u32*)e->KB.p, *pb2 = pr2 + (K + 4);
    unsigned char* flags = (unsigned char*)e->tbits.p + bit_words * 4;
    KSP_HIP(hipMemsetAsync(e->tbits.p, 0, bit_words * 4 + T + 64, st));
    KSP_HIP(hipMemsetAsync(e->dwork.p, 0, ((size_t)nb + 2) * 8, st));
    const u32 shares = (u32)std::min<u64>(64, std::max<u64>(1, 2048 / nb));
    hipLaunchKernelGGL(k_list_pairs, dim3(nb, shares), dim3(256), 0, st, e->bkeys.as<u32>(), e->info.as<u32>(),
                       e->mm.as<uint4>(), e->blk_raw.as<u32>(), e->blk_pos.as<u32>(), pr, pb,
                       e->dwork.as<unsigned long long>());
    int rbits = 1;
    while (rbits < 32 && (U >> rbits)) ++rbits;
    size_t tb = 0;
    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, pr, pr2, pb, pb2, (size_t)K, 0, rbits, st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, pr, pr2, pb, pb2, (size_t)K, 0, rbits, st));
    hipLaunchKernelGGL(k_tile_flags, dim3(grid_for(K, 256)), dim3(256), 0, st, pr2, pb2, K, nb, flags);
    hipLaunchKernelGGL(k_pack_flags, dim3(grid_for(T, 256)), dim3(256), 0, st, flags, T, e->tbits.as<u32>());
    // results to pinned host memory in stream order: the caller's end-of-build synchronisation covers them
    const size_t stage_bytes = bit_words * 4 + ((size_t)nb + 2) * 8;
    if (e->h_stage_bytes < stage_bytes) {
        if (e->h_stage) (void)hipHostFree(e->h_stage);
        e->h_stage = nullptr; e->h_stage_bytes = 0;
        KSP_HIP(hipHostMalloc((void**)&e->h_stage, stage_bytes + stage_bytes / 4 + 4096));
        e->h_stage_bytes = stage_bytes + stage_bytes / 4 + 4096;
    }
    KSP_HIP(hipMemcpyAsync(e->h_stage, e->dwork.p, ((size_t)nb + 2) * 8, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipMemcpyAsync(e->h_stage + ((size_t)nb + 2) * 8, e->tbits.p, bit_words * 4, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipGetLastError());
    e->have_bits = true;
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
    KSP_HIP(hipHostMalloc((void**)&e->h_scal, 128));
    for (int i = 0; i < 6; ++i) KSP_HIP(hipEventCreate(&e->ev[i]));
    *out = e;
    return KSP_OK;
}

void ksp_engine_destroy(ksp_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    ksp::Buf* bufs[] = {&e->d_off, &e->KA, &e->KB, &e->VA, &e->VB, &e->R1, &e->FK, &e->FT, &e->asm_small, &e->tmp, &e->bkeys, &e->info,
                        &e->bw, &e->mm, &e->blk_raw, &e->blk_pos, &e->blk_max, &e->part, &e->scalars, &e->count, &e->tailbuf, &e->smap, &e->tbits, &e->dwork, &e->d_act,
                        &e->d_wg};
    for (auto* b : bufs) b->release();
    if (e->h_count) (void)hipHostFree(e->h_count);
    if (e->h_scal) (void)hipHostFree(e->h_scal);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    for (int i = 0; i < 6; ++i) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    delete e;
}

static int query_slots(ksp_engine* e) {
    if (e->slots) return KSP_OK;
    int per_cu = 0, cus = 0;
    KSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_join<false, true, true>, JW * 64, 0));
    KSP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
    e->slots = (u32)std::max(1, per_cu * cus);
    return KSP_OK;
}

// The join's work list: the tiles that have something to count (block pairs sharing a key, and the
// diagonal tiles of blocks with a multi-source key), each cut into shares by estimated work so that
// a few heavy tiles (related sources end up in the same block: the diagonal ones) do not serialise
// the launch.  Costs are in rough LDS-pipe cycles: ~8 per list word streamed and searched, ~1/2 per
// pair update of a diagonal tile, ~2000 per workgroup (zeroing and flushing the counter tile).
static int build_schedule(ksp_engine* e) {
    e->sched_on = false;
    e->collect = false;
    e->act_tid.clear(); e->act_rec.clear();
    e->st.n_active_tiles = e->st.n_tiles;
    if (!e->have_bits) return KSP_OK;
    const u32 nb = e->nb;
    const u64 T = (u64)nb * (nb + 1) / 2;
    const size_t words = (size_t)(((T + 63) / 64) * 2);
    // (copied to pinned memory by launch_sched_kernels; the build has been synchronised since)
    const unsigned long long* dw = reinterpret_cast<const unsigned long long*>(e->h_stage);
    const u32* bits = reinterpret_cast<const u32*>(e->h_stage + ((size_t)nb + 2) * 8);
    // average holders per list word: expanding a match of two k-holder postings costs k^2 counter updates,
    // collecting it costs the same whatever k is — worth it from about two holders per word (measured on
    // C2: 13 holders/word after the source reordering -> 0.46 vs 1.1 ms; 1.2 in the caller's order -> 20 vs 5 ms)
    e->collect = !e->weighted && e->h_scal_words && dw[nb] >= 2 * e->h_scal_words;
    if (const char* cm = std::getenv("KSP_COLLECT")) e->collect = !e->weighted && std::atoi(cm) != 0;   // diagnostic / tests
    u64 active = 0;
    for (size_t i = 0; i < words; ++i) active += (u64)__builtin_popcount(bits[i]);
    for (u32 b = 0; b < nb; ++b) active += dw[b] != 0;
    if (active * 2 > T) return KSP_OK;   // mostly dense: the plain tile walk is as good
    int rc;
    if ((rc = query_slots(e))) return rc;
    auto words_of = [&](u32 b) { return (u64)(e->h_blk_off[b + 1] - e->h_blk_off[b]); };
    auto cost_of = [&](u32 I, u32 J) -> u64 {
        if (I == J && !e->weighted) return 40 * words_of(I) + 20000;   // bit-sliced: 16 popcounts x 528 patches per 64 keys
        return I == J ? dw[I] / 2 + 10 * words_of(I) + 20000 : 8 * (words_of(I) + words_of(J)) + 20000;
    };
    // pass 1: active tiles and total cost
    u64 total = 0;
    e->act_tid.reserve((size_t)active);
    const char* only = std::getenv("KSP_DEBUG_ONLY");   // timing experiments: "diag" / "off" (results are incomplete)
    const bool skip_diag = only && std::string(only) == "off", skip_off = only && std::string(only) == "diag";
    for (u32 I = 0; I < nb; ++I) {
        const u64 row = tile_row_start(I, nb);
        if (dw[I] && !skip_diag) { e->act_tid.push_back(row); total += cost_of(I, I); }
        for (u64 t = row + 1; !skip_off && t < row + (nb - I);) {
            const u32 wrd = bits[t >> 5] >> (t & 31);
            if (!wrd) { t = (t | 31) + 1; continue; }
            const u64 tt = t + (u64)__builtin_ctz(wrd);
            if (tt >= row + (nb - I)) break;
            e->act_tid.push_back(tt);
            total += cost_of(I, I + (u32)(tt - row));
            t = tt + 1;
        }
    }
    const size_t A = e->act_tid.size();
    // (a sharded job joins 1/nparts of the list per GPU: size the shares for that)
    u64 quarter_shares = 2;   // shares per workgroup slot of the chip, in quarters: half a wave of workgroups measured best
                              // on C2 (0.46 ms vs 0.55 at one per slot, 0.84 at three); KSP_DEBUG_SHARES: experiments
    if (const char* sf = std::getenv("KSP_DEBUG_SHARES")) quarter_shares = std::max(1, std::atoi(sf));
    const u64 target = std::max<u64>(4 * total / ((u64)e->slots * quarter_shares * std::max<u32>(1, e->nparts)) + 1, 100000);
    // pass 2: shares
    std::vector<u32>& wg = e->wg_host;
    wg.clear();
    wg.reserve(A + (size_t)e->slots * 4);
    e->act_rec.resize(4 * (A + 1));
    u32 nsplit = 0;
    for (size_t i = 0; i < A; ++i) {
        u32 I, J;
        tile_decode(e->act_tid[i], nb, I, J);
        u64 sp = (cost_of(I, J) + target - 1) / target;
        sp = std::min<u64>(std::max<u64>(sp, 1), 32);
        u32* r = &e->act_rec[4 * i];
        r[0] = I; r[1] = J; r[2] = (u32)wg.size(); r[3] = nsplit;
        if (sp > 1) ++nsplit;
        for (u64 q = 0; q < sp; ++q) wg.push_back((u32)i);
        if (wg.size() > 0x7FFFFFF0ull) return KSP_OK;   // (absurdly many shares: stay dense)
    }
    u32* r = &e->act_rec[4 * A];
    r[0] = 0; r[1] = 0; r[2] = (u32)wg.size(); r[3] = nsplit;
    if ((rc = e->d_act.ensure(e->act_rec.size() * 4))) return rc;
    if ((rc = e->d_wg.ensure(std::max<size_t>(1, wg.size()) * 4))) return rc;
    KSP_HIP(hipMemcpyAsync(e->d_act.p, e->act_rec.data(), e->act_rec.size() * 4, hipMemcpyHostToDevice, e->sched_stream));
    if (!wg.empty()) KSP_HIP(hipMemcpyAsync(e->d_wg.p, wg.data(), wg.size() * 4, hipMemcpyHostToDevice, e->sched_stream));
    e->sched_on = true;
    e->st.n_active_tiles = A;
    return KSP_OK;
}

// host-side bookkeeping once the full block lists sit in the engine's arrays
static int finish_build(ksp_engine* e) {
    e->st.key_bits = e->key_bits;
    KSP_HIP(hipMemcpy(e->h_blk_max.data(), e->blk_max.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToHost));
    {
        u32 big_blocks = 0;
        for (u32 b = 0; b < e->nb; ++b) big_blocks += e->h_blk_max[b] >= 65536u;
        e->need32 = big_blocks >= 1;   // a big block pairs with itself (diagonal tile) at least
    }
    e->h_blk_off.resize((size_t)e->nb + 1);
    KSP_HIP(hipMemcpy(e->h_blk_off.data(), e->blk_raw.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToHost));
    e->st.n_block_keys = e->h_blk_off[e->nb];
    int rc = build_schedule(e);
    if (rc) return rc;
    e->built = true;
    return KSP_OK;
}

// common front end of build_blocks / build_slice
static int build_common(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights, const uint64_t* h_offsets,
                        uint32_t n_sources, int key_bits, u32 part, u32 nparts, hipStream_t st, const bool slice = false) {
    if (!e || !h_offsets) { set_error("build: NULL argument"); return KSP_E_ARG; }
    if (nparts == 0 || part >= nparts) { set_error("build: bad part / nparts"); return KSP_E_ARG; }
    KSP_HIP(hipSetDevice(e->device));
    e->built = false;
    e->slice_ready = false;
    for (u32 s = 0; s < n_sources; ++s)
        if (h_offsets[s + 1] < h_offsets[s]) { set_error("build: offsets not monotone"); return KSP_E_ARG; }
    const u64 n = n_sources ? h_offsets[n_sources] - h_offsets[0] : 0;
    if (n_sources && h_offsets[0] != 0) { set_error("build: offsets[0] must be 0"); return KSP_E_ARG; }
    if (n >= (1ull << 30)) { set_error("build: more than 2^30 key entries per call"); return KSP_E_LIMIT; }
    if (n && !d_keys) { set_error("build: d_keys is NULL"); return KSP_E_ARG; }
    e->n_sources = n_sources;
    e->n_entries = n;
    e->nb = (n_sources + TB - 1) / TB;
    e->weighted = d_weights != nullptr;
    e->key_bits = key_bits;
    e->have_max_key = false;
    e->nparts = nparts;
    e->part_id = part;
    e->h_off.assign(h_offsets, h_offsets + n_sources + 1);
    if (std::getenv("KSP_FULL_SORT")) e->full_sort = true;   // diagnostic: sort on all key bits
    e->st = ksp_stats{};
    e->st.n_sources = n_sources;
    e->st.n_entries = n;
    e->sort_entries = 0;
    e->st.n_blocks = e->nb;
    e->st.n_tiles = (u64)e->nb * (e->nb + 1) / 2;
    e->st.weighted = e->weighted;
    e->n_kept = 0;
    e->h_blk_off.assign((size_t)e->nb + 1, 0);
    e->h_blk_max.assign((size_t)e->nb + 1, 0);
    std::memset(e->slice_hdr, 0, sizeof e->slice_hdr);
    if (n == 0 || e->nb == 0) return KSP_OK;   // nothing can intersect
    int rc;
    if ((rc = e->d_off.ensure(((size_t)n_sources + 1) * 8))) return rc;
    KSP_HIP(hipEventRecord(e->ev[0], st));
    KSP_HIP(hipMemcpyAsync(e->d_off.p, h_offsets, ((size_t)n_sources + 1) * 8, hipMemcpyHostToDevice, st));
    {   // fine cells: ~32 entries of the largest block per cell, power of two, index kept below 1 GiB
        u64 dmax = 0;
        for (u32 b = 0; b < e->nb; ++b) {
            u64 lo = h_offsets[(u64)b * TB], hi = h_offsets[std::min<u64>(n_sources, (u64)(b + 1) * TB)];
            dmax = std::max(dmax, hi - lo);
        }
        u32 nc = NP;
        while ((u64)nc * 32 < dmax && nc < (1u << 17)) nc <<= 1;
        while (nc > NP && (u64)nc * e->nb > (1ull << 28)) nc >>= 1;
        e->ncell = nc;
        const char* jm = std::getenv("KSP_JOIN");
        e->use_cells = !(jm && std::string(jm) == "window");
    }
    if ((rc = e->blk_max.ensure(((size_t)e->nb + 2) * 4))) return rc;
    if (const char* ro = std::getenv("KSP_REORDER")) e->reorder = std::atoi(ro) != 0;   // diagnostic / tests
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int phase = slice ? 1 : 0;   // a slice stops at the source labels (ksp_engine_slice_finish does the rest)
        rc = e->weighted ? build_impl<true>(e, d_keys, d_weights, st, phase) : build_impl<false>(e, d_keys, d_weights, st, phase);
        if (rc) return rc;
        KSP_HIP(hipMemcpyAsync(e->h_scal + 1, e->scalars.as<u64>() + 1, 64, hipMemcpyDeviceToHost, st));
        KSP_HIP(hipStreamSynchronize(st));
        if ((u32)e->h_scal[4] == 0) break;
        // pathological key distribution (thousands of distinct keys share their top 32 bits):
        // redo with a full-width sort and remember it for later builds on this engine
        e->full_sort = true;
    }
    e->have_bits = false;
    if (!slice && e->n_kept) {   // the work list of the join (slices: after the assemble)
        e->h_scal_words = e->h_scal[1];
        e->h_scal_keys = e->h_scal[2];
        if ((rc = launch_sched_kernels(e, st))) return rc;
    }
    KSP_HIP(hipEventRecord(e->ev[1], st));
    KSP_HIP(hipStreamSynchronize(st));
    KSP_HIP(hipEventElapsedTime(&e->st.ms_build, e->ev[0], e->ev[1]));
    e->st.key_bits = e->key_bits;
    e->st.ms_sort = 0;
    e->st.sort_entries = e->sort_entries;
    e->st.sort_bits = e->sort_bits;
    if (e->sort_entries) KSP_HIP(hipEventElapsedTime(&e->st.ms_sort, e->ev[4], e->ev[5]));
    return KSP_OK;
}

int ksp_engine_build_postings(ksp_engine* e, const uint64_t* h_key_off, const uint32_t* d_sources,
                              const uint32_t* d_key_weights, uint32_t n_keys, uint32_t n_sources, void* stream) {
    if (!e || (n_keys && (!h_key_off || !d_sources))) { set_error("build_postings: NULL argument"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    e->built = false;
    e->slice_ready = false;
    e->slice_phase = 0;
    const u64 n = n_keys ? h_key_off[n_keys] : 0;
    if (n_keys && h_key_off[0] != 0) { set_error("build_postings: key_off[0] must be 0"); return KSP_E_ARG; }
    for (u32 k = 0; k < n_keys; ++k)
        if (h_key_off[k + 1] < h_key_off[k] + 2) { set_error("build_postings: every key needs at least two holders"); return KSP_E_ARG; }
    if (n >= (1ull << 30)) { set_error("build_postings: more than 2^30 entries per call"); return KSP_E_LIMIT; }
    e->n_sources = n_sources;
    e->n_entries = n;
    e->nb = (n_sources + TB - 1) / TB;
    e->weighted = d_key_weights != nullptr;
    e->key_bits = 0;
    e->have_max_key = false;
    e->nparts = 1;
    e->part_id = 0;
    e->h_off.clear();
    e->st = ksp_stats{};
    e->st.n_sources = n_sources;
    e->st.n_entries = n;
    e->sort_entries = 0;
    e->st.n_blocks = e->nb;
    e->st.n_tiles = (u64)e->nb * (e->nb + 1) / 2;
    e->st.weighted = e->weighted;
    e->n_kept = 0;
    e->h_blk_off.assign((size_t)e->nb + 1, 0);
    e->h_blk_max.assign((size_t)e->nb + 1, 0);
    e->have_bits = false;
    if (n == 0 || e->nb == 0) {   // nothing can intersect
        e->st.n_block_keys = 0;
        e->need32 = false;
        e->built = true;
        return KSP_OK;
    }
    int rc;
    KSP_HIP(hipEventRecord(e->ev[0], st));
    {   // fine cells as in build_common, from the mean block size (the holders are not grouped by source here)
        const u64 dmax = 2 * (n / e->nb + 1);
        u32 nc = NP;
        while ((u64)nc * 32 < dmax && nc < (1u << 17)) nc <<= 1;
        while (nc > NP && (u64)nc * e->nb > (1ull << 28)) nc >>= 1;
        e->ncell = nc;
        const char* jm = std::getenv("KSP_JOIN");
        e->use_cells = !(jm && std::string(jm) == "window");
    }
    if ((rc = e->blk_max.ensure(((size_t)e->nb + 2) * 4))) return rc;
    if (const char* ro = std::getenv("KSP_REORDER")) e->reorder = std::atoi(ro) != 0;
    // key offsets as 32-bit device array (= first entry of every rank)
    if ((rc = e->d_off.ensure(((size_t)n_keys + 2) * 4))) return rc;
    {
        std::vector<u32> off32((size_t)n_keys + 1);
        for (u32 k = 0; k <= n_keys; ++k) off32[k] = (u32)h_key_off[k];
        KSP_HIP(hipMemcpyAsync(e->d_off.p, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, st));
        KSP_HIP(hipStreamSynchronize(st));   // (off32 is a local)
    }
    e->post_off = e->d_off.as<u32>();
    e->post_src = d_sources;
    e->post_w = d_key_weights;
    e->post_nkeys = n_keys;
    rc = e->weighted ? build_impl<true>(e, nullptr, nullptr, st, 3) : build_impl<false>(e, nullptr, nullptr, st, 3);
    e->post_off = e->post_src = e->post_w = nullptr;
    if (rc) return rc;
    KSP_HIP(hipMemcpyAsync(e->h_scal + 1, e->scalars.as<u64>() + 1, 64, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipStreamSynchronize(st));
    if ((u32)e->h_scal[4]) { set_error("build_postings: a source index is >= n_sources"); return KSP_E_ARG; }
    if (e->n_kept) {
        e->h_scal_words = e->h_scal[1];
        e->h_scal_keys = e->h_scal[2];
        if ((rc = launch_sched_kernels(e, st))) return rc;
    }
    KSP_HIP(hipEventRecord(e->ev[1], st));
    KSP_HIP(hipStreamSynchronize(st));
    KSP_HIP(hipEventElapsedTime(&e->st.ms_build, e->ev[0], e->ev[1]));
    return finish_build(e);
}

int ksp_engine_build_blocks(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights,
                            const uint64_t* h_offsets, uint32_t n_sources, int key_bits, void* stream) {
    int rc = build_common(e, d_keys, d_weights, h_offsets, n_sources, key_bits, 0, 1, (hipStream_t)stream);
    if (rc) return rc;
    if (e->n_entries == 0 || e->nb == 0 || e->n_kept == 0) {   // no key is held by two sources: all counts are zero
        e->st.n_block_keys = 0;
        e->need32 = false;
        e->built = true;
        return KSP_OK;
    }
    return finish_build(e);
}

// ---- key-range slices: build one part, export it, assemble all parts ----------------------
int ksp_engine_build_slice(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights,
                           const uint64_t* h_offsets, uint32_t n_sources, int key_bits, uint32_t part,
                           uint32_t nparts, void* stream) {
    int rc = build_common(e, d_keys, d_weights, h_offsets, n_sources, key_bits, part, nparts, (hipStream_t)stream, true);
    if (rc) return rc;
    e->slice_phase = 1;
    return KSP_OK;
}

static u32* label_array(ksp_engine* e) { return e->smap.as<u32>(); }   // [0] of the per-source maps

int ksp_engine_slice_labels(ksp_engine* e, uint32_t* d_labels, void* stream) {
    if (!e || !d_labels) { set_error("slice_labels: NULL argument"); return KSP_E_ARG; }
    if (e->slice_phase != 1) { set_error("slice_labels: build_slice has not been run"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    if (e->n_entries == 0 || e->nb == 0) {   // nothing was built: identity labels
        hipLaunchKernelGGL(k_iota, dim3(grid_for(std::max<u32>(1, e->n_sources), 256)), dim3(256), 0, st, d_labels, e->n_sources);
    } else {
        KSP_HIP(hipMemcpyAsync(d_labels, label_array(e), (size_t)e->n_sources * 4, hipMemcpyDeviceToDevice, st));
    }
    KSP_HIP(hipStreamSynchronize(st));
    return KSP_OK;
}

int ksp_engine_slice_finish(ksp_engine* e, const uint32_t* d_labels, void* stream) {
    if (!e) { set_error("slice_finish: NULL argument"); return KSP_E_ARG; }
    if (e->slice_phase != 1) { set_error("slice_finish: build_slice has not been run"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    e->slice_phase = 0;
    if (e->n_entries == 0 || e->nb == 0) { e->slice_ready = true; return KSP_OK; }
    int rc;
    KSP_HIP(hipEventRecord(e->ev[0], st));
    if (d_labels) KSP_HIP(hipMemcpyAsync(label_array(e), d_labels, (size_t)e->n_sources * 4, hipMemcpyDeviceToDevice, st));
    rc = e->weighted ? build_impl<true>(e, nullptr, nullptr, st, 2) : build_impl<false>(e, nullptr, nullptr, st, 2);
    if (rc) return rc;
    KSP_HIP(hipMemcpyAsync(e->h_scal + 1, e->scalars.as<u64>() + 1, 64, hipMemcpyDeviceToHost, st));
    if (e->n_kept == 0) {
        // empty slice: valid (all-pad) lists so that export / assemble need no special case
        if ((rc = e->blk_raw.ensure(((size_t)e->nb + 2) * 4))) return rc;
        KSP_HIP(hipMemsetAsync(e->blk_raw.p, 0, ((size_t)e->nb + 2) * 4, st));
        hipLaunchKernelGGL(k_blk_pos, dim3(1), dim3(64), 0, st, e->blk_raw.as<u32>(), e->blk_pos.as<u32>(),
                           e->scalars.as<u64>(), e->nb);
        const u64 lpad = (u64)e->nb * (WIN + 4) + 4 * WIN;
        hipLaunchKernelGGL(k_fill, dim3(grid_for(lpad, 256)), dim3(256), 0, st, e->bkeys.as<u32>(), PAD, lpad);
        KSP_HIP(hipMemcpyAsync(e->h_scal + 3, e->scalars.as<u64>() + 3, 8, hipMemcpyDeviceToHost, st));
    }
    KSP_HIP(hipEventRecord(e->ev[1], st));
    KSP_HIP(hipStreamSynchronize(st));
    float ms = 0;
    KSP_HIP(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    e->st.ms_build += ms;
    e->slice_hdr[0] = e->h_scal[3];           // padded length
    if (e->n_kept == 0) {
        e->slice_hdr[1] = e->slice_hdr[2] = e->slice_hdr[3] = 0;
    } else {
        e->slice_hdr[1] = e->h_scal[2];       // distinct keys (U)
        e->slice_hdr[2] = e->h_scal[7];       // big postings
        e->slice_hdr[3] = e->h_scal[1];       // block keys
    }
    e->slice_ready = true;
    return KSP_OK;
}

int ksp_engine_slice_sizes(const ksp_engine* e, uint64_t out[4]) {
    if (!e || !out) return KSP_E_ARG;
    if (!e->slice_ready) { set_error("slice_sizes: build_slice has not been run"); return KSP_E_ARG; }
    for (int i = 0; i < 4; ++i) out[i] = e->slice_hdr[i];
    return KSP_OK;
}

int ksp_engine_slice_export(ksp_engine* e, uint32_t* d_brk, uint32_t* d_info, uint32_t* d_bw, uint32_t* d_blk_raw,
                            uint32_t* d_blk_pos, void* d_big, void* stream) {
    if (!e || !e->slice_ready) { set_error("slice_export: build_slice has not been run"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    if (e->n_entries == 0 || e->nb == 0) return KSP_OK;
    const size_t L = (size_t)e->slice_hdr[0];
    KSP_HIP(hipMemcpyAsync(d_brk, e->bkeys.p, L * 4, hipMemcpyDeviceToDevice, st));
    if (e->slice_hdr[3]) {
        KSP_HIP(hipMemcpyAsync(d_info, e->info.p, L * 4, hipMemcpyDeviceToDevice, st));
        if (e->weighted && d_bw) KSP_HIP(hipMemcpyAsync(d_bw, e->bw.p, L * 4, hipMemcpyDeviceToDevice, st));
    }
    KSP_HIP(hipMemcpyAsync(d_blk_raw, e->blk_raw.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToDevice, st));
    KSP_HIP(hipMemcpyAsync(d_blk_pos, e->blk_pos.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (e->slice_hdr[2]) KSP_HIP(hipMemcpyAsync(d_big, e->mm.p, (size_t)e->slice_hdr[2] * 16, hipMemcpyDeviceToDevice, st));
    KSP_HIP(hipStreamSynchronize(st));
    return KSP_OK;
}

int ksp_engine_assemble(ksp_engine* e, uint32_t nparts, const uint64_t* h_sizes /* nparts x 4 */,
                        const uint32_t* d_brk_all, const uint32_t* d_info_all, const uint32_t* d_bw_all,
                        uint64_t lstride, const uint32_t* d_blk_raw_all, const uint32_t* d_blk_pos_all,
                        const void* d_big_all, uint64_t bigstride, void* stream) {
    if (!e || !h_sizes || nparts == 0) { set_error("assemble: bad argument"); return KSP_E_ARG; }
    if (!e->slice_ready) { set_error("assemble: build_slice must run on this engine first (it sets the geometry)"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    e->built = false;
    if (e->n_entries == 0 || e->nb == 0) { e->built = true; return KSP_OK; }
    const u32 nb = e->nb;
    u64 ktot = 0, utot = 0, bigtot = 0;
    std::vector<u32> off(2 * (size_t)nparts);
    for (u32 p = 0; p < nparts; ++p) {
        off[p] = (u32)utot;             // rank offset of part p
        off[nparts + p] = (u32)bigtot;  // mask index offset of part p
        utot += h_sizes[4 * p + 1];
        bigtot += h_sizes[4 * p + 2];
        ktot += h_sizes[4 * p + 3];
    }
    if (utot >= (1ull << 30)) { set_error("assemble: more than 2^30 distinct keys"); return KSP_E_LIMIT; }
    e->n_kept = ktot;   // (>0 iff some key is shared)
    if (ktot == 0) {
        e->st.n_block_keys = 0;
        e->need32 = false;
        e->h_blk_off.assign((size_t)nb + 1, 0);
        e->built = true;
        return KSP_OK;
    }
    int rc;
    const u64 lmax = ktot + (u64)nb * (WIN + 4) + 4 * WIN;
    if ((rc = e->bkeys.ensure(lmax * 4))) return rc;
    if ((rc = e->info.ensure(lmax * 4))) return rc;
    if (e->weighted && (rc = e->bw.ensure(lmax * 4))) return rc;
    if ((rc = e->mm.ensure((bigtot + 16) * 16))) return rc;
    if ((rc = e->asm_small.ensure(off.size() * 4))) return rc;
    KSP_HIP(hipEventRecord(e->ev[0], st));
    KSP_HIP(hipMemcpyAsync(e->asm_small.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
    u64* scal = e->scalars.as<u64>();
    hipLaunchKernelGGL(k_asm_counts, dim3(1), dim3(64), 0, st, d_blk_raw_all, nb + 1, nparts, nb, e->blk_raw.as<u32>());
    hipLaunchKernelGGL(k_blk_pos, dim3(1), dim3(64), 0, st, e->blk_raw.as<u32>(), e->blk_pos.as<u32>(), scal, nb);
    hipLaunchKernelGGL(k_pad, dim3(nb + 1), dim3(256), 0, st, e->blk_raw.as<u32>(), e->blk_pos.as<u32>(), e->bkeys.as<u32>(), nb, PAD);
    const u32* roff = e->asm_small.as<u32>();
    if (e->weighted)
        hipLaunchKernelGGL((k_asm_copy<true>), dim3(nb, nparts), dim3(256), 0, st, d_brk_all, d_info_all, d_bw_all,
                           (size_t)lstride, d_blk_raw_all, d_blk_pos_all, nb + 1, roff, roff + nparts,
                           e->blk_pos.as<u32>(), e->bkeys.as<u32>(), e->info.as<u32>(), e->bw.as<u32>());
    else
        hipLaunchKernelGGL((k_asm_copy<false>), dim3(nb, nparts), dim3(256), 0, st, d_brk_all, d_info_all, d_bw_all,
                           (size_t)lstride, d_blk_raw_all, d_blk_pos_all, nb + 1, roff, roff + nparts,
                           e->blk_pos.as<u32>(), e->bkeys.as<u32>(), e->info.as<u32>(), (u32*)nullptr);
    for (u32 p = 0; p < nparts; ++p)
        if (h_sizes[4 * p + 2])
            KSP_HIP(hipMemcpyAsync((char*)e->mm.p + (size_t)off[nparts + p] * 16,
                                   (const char*)d_big_all + (size_t)p * bigstride * 16, (size_t)h_sizes[4 * p + 2] * 16,
                                   hipMemcpyDeviceToDevice, st));
    e->h_scal[1] = ktot;
    e->h_scal[2] = utot;
    KSP_HIP(hipMemcpyAsync(scal + 1, e->h_scal + 1, 16, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_cidx, dim3(grid_for((u64)nb * (e->ncell + 1), 256)), dim3(256), 0, st, e->bkeys.as<u32>(),
                       e->blk_raw.as<u32>(), e->blk_pos.as<u32>(), scal, e->part.as<u32>(), nb, e->ncell);
    KSP_HIP(hipGetLastError());
    e->h_scal_words = ktot;
    e->h_scal_keys = utot;
    if ((rc = launch_sched_kernels(e, st))) return rc;
    KSP_HIP(hipEventRecord(e->ev[1], st));
    KSP_HIP(hipStreamSynchronize(st));
    float ms = 0;
    KSP_HIP(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    e->st.ms_build += ms;   // slice build + assemble
    return finish_build(e);
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

uint64_t ksp_engine_edge_bound(const ksp_engine* e, uint64_t t0, uint64_t t1) {
    if (!e || !e->nb || !e->built) return 0;
    if (!e->sched_on) return ksp_engine_tile_pairs(e, t0, t1);
    const size_t a0 = std::lower_bound(e->act_tid.begin(), e->act_tid.end(), t0) - e->act_tid.begin();
    const size_t a1 = std::lower_bound(e->act_tid.begin(), e->act_tid.end(), t1) - e->act_tid.begin();
    const u64 last = e->n_sources - (u64)(e->nb - 1) * TB;
    u64 pairs = 0;
    for (size_t i = a0; i < a1; ++i) {
        const u32 I = e->act_rec[4 * i], J = e->act_rec[4 * i + 1];
        const u64 nI = (I == e->nb - 1) ? last : TB, nJ = (J == e->nb - 1) ? last : TB;
        pairs += (I == J) ? nI * (nI - 1) / 2 : nI * nJ;
    }
    return pairs;
}

int ksp_engine_balanced_cuts(const ksp_engine* e, uint32_t nparts, uint64_t* cuts) {
    if (!e || !cuts || nparts == 0) { set_error("balanced_cuts: bad argument"); return KSP_E_ARG; }
    if (!e->built) { set_error("balanced_cuts: build_blocks / assemble has not been run"); return KSP_E_ARG; }
    const u64 T = ksp_engine_num_tiles(e);
    if (!e->sched_on) {   // dense mode: equal tile counts
        for (u32 p = 0; p <= nparts; ++p) cuts[p] = (T * p) / nparts;
        return KSP_OK;
    }
    // work-list mode: equal numbers of workgroups (shares are sized by estimated work), whole tiles only
    const size_t A = e->act_tid.size();
    const u64 W = e->act_rec[4 * A + 2];
    cuts[0] = 0;
    size_t i = 0;
    for (u32 p = 1; p < nparts; ++p) {
        const u64 want = (W * p) / nparts;
        while (i < A && e->act_rec[4 * i + 2] < want) ++i;
        cuts[p] = i < A ? e->act_tid[i] : T;
    }
    cuts[nparts] = T;
    return KSP_OK;
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
    if (tile_begin >= tile_end || e->n_entries == 0 || e->n_kept == 0) return KSP_OK;
    if (capacity && !d_edges) { set_error("join: d_edges is NULL"); return KSP_E_ARG; }
    int rc;
    if ((rc = e->count.ensure(64))) return rc;
    JoinArgs a;
    a.brk = e->bkeys.as<u32>();
    a.info = e->info.as<u32>();
    a.bw = e->weighted ? e->bw.as<u32>() : nullptr;
    a.bigmask = e->mm.as<uint4>();
    a.blk_raw = e->blk_raw.as<u32>();
    a.blk_pos = e->blk_pos.as<u32>();
    a.cidx = e->part.as<u32>();
    a.ncell = e->ncell;
    a.nb = e->nb;
    a.n_sources = e->n_sources;
    a.tile_begin = tile_begin;
    a.out = d_edges;
    a.cap = capacity;
    a.out_count = e->count.as<unsigned long long>();
    a.dbg = 0;
    if (const char* dbg = std::getenv("KSP_DEBUG_ABLATE")) a.dbg = (u32)std::atoi(dbg);
    KSP_HIP(hipMemsetAsync(a.out_count, 0, 8, st));
    KSP_HIP(hipEventRecord(e->ev[2], st));
    dim3 block(JW * 64);
    a.blk_max = e->blk_max.as<u32>();
    a.inv = e->smap.as<u32>() + 3 * (((size_t)e->n_sources + 64) & ~(size_t)63);
    a.sched = nullptr; a.act = nullptr; a.wg0 = 0; a.split0 = 0; a.tail_done = nullptr;
    a.collect = e->collect ? 1u : 0u;
    auto launch = [&](bool c16, dim3 grid, const JoinArgs& args) {
        if (e->use_cells) {
            if (e->weighted) { if (c16) hipLaunchKernelGGL((k_join<true, true, true>), grid, block, 0, st, args); else hipLaunchKernelGGL((k_join<true, false, true>), grid, block, 0, st, args); }
            else { if (c16) hipLaunchKernelGGL((k_join<false, true, true>), grid, block, 0, st, args); else hipLaunchKernelGGL((k_join<false, false, true>), grid, block, 0, st, args); }
        } else {
            if (e->weighted) { if (c16) hipLaunchKernelGGL((k_join<true, true, false>), grid, block, 0, st, args); else hipLaunchKernelGGL((k_join<true, false, false>), grid, block, 0, st, args); }
            else { if (c16) hipLaunchKernelGGL((k_join<false, true, false>), grid, block, 0, st, args); else hipLaunchKernelGGL((k_join<false, false, false>), grid, block, 0, st, args); }
        }
    };
    // HIP caps a launch at 2^32 threads: at most 4 Mi workgroups of 512 threads per launch (a larger grid
    // silently runs only part of its blocks — seen with 30.5 M tiles on MI355X / ROCm 7.2)
    const u64 kMaxTilesPerLaunch = 4ull << 20;
    size_t act0 = 0, act1 = 0;
    if (e->sched_on && st != e->sched_stream) KSP_HIP(hipStreamSynchronize(e->sched_stream));   // (work list uploaded on the build's stream)
    if (e->sched_on) {
        // work-list mode: the active tiles of [tile_begin, tile_end), each in its shares
        act0 = std::lower_bound(e->act_tid.begin(), e->act_tid.end(), tile_begin) - e->act_tid.begin();
        act1 = std::lower_bound(e->act_tid.begin(), e->act_tid.end(), tile_end) - e->act_tid.begin();
        const u32 wgA = e->act_rec[4 * act0 + 2], wgB = e->act_rec[4 * act1 + 2];
        const u32 spA = e->act_rec[4 * act0 + 3], spB = e->act_rec[4 * act1 + 3];
        a.sched = e->d_wg.as<u32>();
        a.act = e->d_act.as<u32>();
        a.split0 = spA;
        a.n_normal = 0; a.tail_sp = 1;
        a.tailbuf = nullptr;
        const u32 nsplit = spB - spA;
        if (nsplit) {
            if ((rc = e->tailbuf.ensure(((size_t)nsplit * TB * TB + nsplit + 16) * 4))) return rc;
            a.tailbuf = e->tailbuf.as<u32>();
            a.tail_done = a.tailbuf + (size_t)nsplit * TB * TB;
            KSP_HIP(hipMemsetAsync(a.tailbuf, 0, ((size_t)nsplit * TB * TB + nsplit + 16) * 4, st));
        }
        for (int pass = 0; pass < (e->need32 ? 2 : 1); ++pass)
            for (u64 w = wgA; w < wgB; w += kMaxTilesPerLaunch) {
                a.wg0 = (u32)w;
                launch(pass == 0, dim3((u32)std::min<u64>(kMaxTilesPerLaunch, wgB - w)), a);
            }
        KSP_HIP(hipGetLastError());
    }
    for (u64 chunk_begin = tile_begin; !e->sched_on && chunk_begin < tile_end; chunk_begin += kMaxTilesPerLaunch) {
    const u64 chunk_end = std::min(tile_end, chunk_begin + kMaxTilesPerLaunch);
    a.tile_begin = chunk_begin;
    const u32 ntiles = (u32)(chunk_end - chunk_begin);
    // Tile splitting for small launches: when there are fewer tiles than workgroup slots on the chip,
    // every tile is cut into `sp` rank-range shares (partial counters are summed in a global buffer).
    u32 n_tail = 0, sp = 1;
    if ((rc = query_slots(e))) return rc;
    // (Measured on C2, 3160 tiles on 768 slots: splitting the 88 "remainder" tiles does not pay — tiles
    //  finish at different times and the dispatcher back-fills — so only under-filled launches split.)
    if (!std::getenv("KSP_NO_TAIL_SPLIT")) {
        const u32 r = ntiles < e->slots ? ntiles : 0;
        if (r > 0 && (u64)r * 4 <= (u64)e->slots * 3) {
            sp = std::min<u32>(8, e->slots / r);
            if (sp >= 2) n_tail = r; else sp = 1;
        }
    }
    a.n_normal = ntiles - n_tail;
    a.tail_sp = sp;
    a.tailbuf = nullptr;
    if (n_tail) {
        if ((rc = e->tailbuf.ensure((size_t)n_tail * TB * TB * 4))) return rc;
        a.tailbuf = e->tailbuf.as<u32>();
        KSP_HIP(hipMemsetAsync(a.tailbuf, 0, (size_t)n_tail * TB * TB * 4, st));
    }
    dim3 grid(a.n_normal + n_tail * sp);
    // packed 16-bit counters wherever they are exact; 32-bit counters for the other tiles
    launch(true, grid, a);
    if (n_tail) hipLaunchKernelGGL(k_tail_emit, dim3(n_tail), block, 0, st, a);
    if (e->need32) {   // tiles whose two blocks both hold huge sketches: one workgroup per tile, no tail split
        JoinArgs b = a;
        b.n_normal = ntiles;
        b.tail_sp = 1;
        b.tailbuf = nullptr;
        launch(false, dim3(ntiles), b);
    }
    KSP_HIP(hipGetLastError());
    }   // launch chunks
    KSP_HIP(hipEventRecord(e->ev[3], st));
    KSP_HIP(hipMemcpyAsync(e->h_count, a.out_count, 8, hipMemcpyDeviceToHost, st));
    KSP_HIP(hipStreamSynchronize(st));
    KSP_HIP(hipEventElapsedTime(&e->st.ms_join, e->ev[2], e->ev[3]));
    *h_count = *e->h_count;
    e->st.last_tiles = tile_end - tile_begin;
    e->st.last_active_tiles = e->sched_on ? (u64)(act1 - act0) : tile_end - tile_begin;
    e->st.last_pairs = ksp_engine_tile_pairs(e, tile_begin, tile_end);
    e->st.last_edges = *h_count;
    {   // bytes the kernel streams from both block lists; self tiles read info [+ weight] only
        const u64 per = 4;   // only the 32-bit ranks are streamed; posting words are gathered on matches
        u64 bytes = 0;
        for (size_t i = act0; e->sched_on && i < act1; ++i) {   // work-list mode: the active tiles only
            const u32 I = e->act_rec[4 * i], J = e->act_rec[4 * i + 1];
            const u64 kI = e->h_blk_off[I + 1] - e->h_blk_off[I], kJ = e->h_blk_off[J + 1] - e->h_blk_off[J];
            bytes += (J == I) ? kI * (e->weighted ? 8 : 4) : (kI + kJ) * per;
        }
        for (u64 t = tile_begin; !e->sched_on && t < tile_end;) {
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
    KSP_HIP(hipMemcpy(h_blk_off, e->blk_raw.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToHost));
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

// join every tile of a built engine and bring the edges to the host, sorted by (source_1, source_2)
static int collect_all_edges(ksp_engine* e, int device, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats) {
    void* d_edges = nullptr;
    std::vector<ksp_edge> all;
    int rc = KSP_OK;
    do {
        // Tile ranges as large as possible: start with everything; when the edge buffer overflows, grow it
        // to the reported count if memory allows, otherwise halve the range (sparse inputs need one launch,
        // dense ones are cut into ranges whose non-zero pairs fit).
        const u64 T = ksp_engine_num_tiles(e);
        u64 cap = std::min<u64>(1ull << 24, ksp_engine_edge_bound(e, 0, T) + 1);   // at most 256 MiB to begin with
        if ((rc = ksp_device_malloc(device, cap * sizeof(ksp_edge), &d_edges))) break;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const u64 max_cap = std::max<u64>(cap, (u64)(free_b / 2) / sizeof(ksp_edge));
        u64 t = 0;
        u64 step = std::max<u64>(T, 1);
        ksp_stats acc{};
        while (t < T && !rc) {
            u64 t1 = std::min(T, t + step);
            u64 cnt = 0;
            rc = ksp_engine_join(e, t, t1, (ksp_edge*)d_edges, cap, &cnt, nullptr);
            if (rc == KSP_E_OVERFLOW) {
                rc = KSP_OK;
                acc.ms_join += e->st.ms_join;
                if (cnt <= max_cap) {
                    (void)hipFree(d_edges);
                    d_edges = nullptr;
                    cap = std::min<u64>(max_cap, cnt + cnt / 16 + 1024);
                    if ((rc = ksp_device_malloc(device, cap * sizeof(ksp_edge), &d_edges))) break;
                } else if (t1 - t > 1) {
                    step = (t1 - t) / 2;
                } else {
                    set_error("pairwise_host: a single tile yields more edges than fit in device memory");
                    rc = KSP_E_LIMIT;
                }
                continue;
            }
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
    if (d_edges) (void)hipFree(d_edges);
    return rc;
}

int ksp_pairwise_host(const uint64_t* keys, const uint32_t* weights, const uint64_t* offsets, uint32_t n_sources,
                      int device, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats) {
    if (!offsets || !out_edges || !n_edges) { set_error("pairwise_host: NULL argument"); return KSP_E_ARG; }
    *out_edges = nullptr;
    *n_edges = 0;
    ksp_engine* e = nullptr;
    int rc = ksp_engine_create(device, &e);
    if (rc) return rc;
    const u64 n = n_sources ? offsets[n_sources] : 0;
    void *d_keys = nullptr, *d_w = nullptr;
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
        rc = collect_all_edges(e, device, out_edges, n_edges, stats);
    } while (0);
    if (d_keys) (void)hipFree(d_keys);
    if (d_w) (void)hipFree(d_w);
    ksp_engine_destroy(e);
    return rc;
}

int ksp_pairwise_postings_host(const uint64_t* key_off, const uint32_t* sources, const uint32_t* key_weights,
                               uint32_t n_keys, uint32_t n_sources, int device, ksp_edge** out_edges,
                               uint64_t* n_edges, ksp_stats* stats) {
    if (!out_edges || !n_edges || (n_keys && (!key_off || !sources))) { set_error("pairwise_postings_host: NULL argument"); return KSP_E_ARG; }
    *out_edges = nullptr;
    *n_edges = 0;
    ksp_engine* e = nullptr;
    int rc = ksp_engine_create(device, &e);
    if (rc) return rc;
    const u64 n = n_keys ? key_off[n_keys] : 0;
    for (u64 i = 0; i < n; ++i)
        if (sources[i] >= n_sources) { set_error("pairwise_postings_host: source index out of range"); ksp_engine_destroy(e); return KSP_E_ARG; }
    void *d_src = nullptr, *d_w = nullptr;
    do {
        if (n) {
            if ((rc = ksp_device_malloc(device, n * 4, &d_src))) break;
            if ((rc = ksp_memcpy_h2d(d_src, sources, n * 4))) break;
            if (key_weights) {
                if ((rc = ksp_device_malloc(device, (u64)n_keys * 4, &d_w))) break;
                if ((rc = ksp_memcpy_h2d(d_w, key_weights, (u64)n_keys * 4))) break;
            }
        }
        if ((rc = ksp_engine_build_postings(e, key_off, (const u32*)d_src, (const u32*)d_w, n_keys, n_sources, nullptr))) break;
        rc = collect_all_edges(e, device, out_edges, n_edges, stats);
    } while (0);
    if (d_src) (void)hipFree(d_src);
    if (d_w) (void)hipFree(d_w);
    ksp_engine_destroy(e);
    return rc;
}

}  // extern "C"
