// Stage 1, first step: the entries (key, source tag) are PARTITIONED by key into hash buckets of ~2 000
// entries, which k_bucket_group then groups by full key in LDS.  Hand-written for gfx950; replaces the
// rocPRIM radix partition (histogram pass + two 8-bit onesweep passes over 8-byte key + tag, and a
// tagging pass in front of them).  Included by engine.hip inside namespace ksp.
//
// What the reference does here: src/pairwise.cpp:194-209 turns the colour map into a vector and cuts it
// into per-thread slices — it brings the sources of equal colours together.  The sketch path has to bring
// equal hashes together first; this is that step.
//
//   bucket(key) = floor(key * nbuckets / (maxkey + 1))          (one 64 x 64 -> high 64 multiply: monotone
//                                                                in the key, uniform over the real hash
//                                                                range, which is not a power of two; any
//                                                                bucket count, not just powers of two)
//   level 1  k_part1   one workgroup per 4 096 consecutive entries of the sketch array.  Every source's
//            run is sorted, so the entries of one level-1 bucket (bucket id >> pb2, at most 256 of them)
//            form runs of consecutive entries: a wave finds the run heads with one ballot, the head lane
//            counts the run into an LDS histogram, and each (workgroup, bucket) pair reserves its place
//            with ONE global atomic.  No histogram pass and no capacity guess: a level-1 bucket is a list
//            of 4 096-entry PAGES taken from a pool on demand (the reservation that crosses a page start
//            allocates that page and publishes it in the list's page table; nobody else ever waits for
//            anything before publishing, so the short, bounded spin of the others cannot deadlock).  Each
//            bucket has 8 sub-lists, chosen by blockIdx % 8 — workgroups b and b + 8 share an XCD, so the
//            lines at a sub-list's tail are filled inside ONE L2 — with cursors and page pools on memory
//            lines of their own.  The source tag is made on the fly (no tagging pass; the chunk's first
//            source comes from a table built by k_part_src); the low pb2 bits of the bucket id go to a byte
//            array beside the keys.
//   level 2  k_hist2   one workgroup per level-1 bucket counts the entries of its <= 256 final buckets from
//            the digit bytes (1/8 of the bytes of the keys); a scan gives every final bucket its exact place
//            in the dense output (bstart[], what k_bucket_bounds used to search for).
//            k_scatter2   one workgroup per PAGE: orders its 4 096 entries by final bucket in LDS (counting
//            sort with LDS atomics; the order inside a bucket is irrelevant), reserves the run of every
//            bucket with one global atomic on that bucket's cursor and writes the runs out.
//
// HBM traffic per entry (8-byte key, 2-byte tag): level 1 reads 8, writes 11; level 2 reads 1 + 10,
// writes 10 — 40 bytes, against 8 + 2 (tagging) + 8 (histogram) + 2 x 20 (two passes) = 58 before.
// Anything the page tables cannot hold (a key distribution more than ~16 x off uniform) raises the
// overflow word: the build falls back to the rocPRIM partition, which stays in the engine.
#pragma once

constexpr u32 P1_CH = 4096, P1_THREADS = 512, P1_EPT = P1_CH / P1_THREADS;
constexpr u32 P1_R = 8;           // sub-lists per level-1 bucket (one per XCD)
constexpr u32 P1_LINE = 32;       // u32 words per cursor: a memory line of its own
constexpr u32 P1_PLOG = 12;       // page = 4 096 entries
constexpr u32 P1_PAGE = 1u << P1_PLOG;
constexpr u32 P1_PTW_MAX = 8192;  // page-table entries per sub-list (8 x the expected number + 16, at most this)
constexpr u32 P2_THREADS = 512, P2_TILE = P1_PAGE, P2_EPT = P2_TILE / P2_THREADS;   // one page per workgroup

// control words inside the engine's scalar block (u64 units): [0] largest key (k_max_last),
// PC_MULT the multiplier, PC_MODE: low word 1 = identity buckets, PC_OVF: low word = overflow
constexpr u32 PC_MULT = 12, PC_MODE = 13, PC_OVF = 14;

__device__ inline u32 part_bucket(const u64 key, const u64 mult, const u32 ident, const u32 nbm1) {
    return ident ? (u32)min(key, (u64)nbm1) : (u32)__umul64hi(key, mult);
}

// mult = floor(nbuckets * 2^64 / (maxkey + 1)); fewer key values than buckets: every key its own bucket
__device__ inline void part_prep(u64* __restrict__ scal, const u32 nbuckets) {
    const u64 maxkey = scal[0];
    u64 mult = 0;
    u32 ident = 0;
    if (maxkey <= (u64)nbuckets) ident = 1;
    else if (maxkey == ~0ull) mult = nbuckets;                     // umulhi(key, nbuckets): the top fraction of the key
    else {
        const u64 M = maxkey + 1;
        u64 rem = nbuckets, q = 0;                                 // (nbuckets : 0) / M, nbuckets < M
        for (int i = 63; i >= 0; --i) {
            const u64 carry = rem >> 63;
            rem <<= 1;
            if (carry || rem >= M) { rem -= M; q |= 1ull << i; }
        }
        mult = q;
    }
    scal[PC_MULT] = mult;
    reinterpret_cast<u32*>(scal + PC_MODE)[0] = ident;
}

// last source s with off[s] <= pos  (off[0] = 0 <= pos < off[n_sources])
__device__ inline u32 part_source_of(const u64* __restrict__ off, const u32 n_sources, const u64 pos) {
    u32 lo = 0, hi = n_sources;
    while (hi - lo > 1) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ inline u32 part_wait_page(u32* __restrict__ pt, const u32 row, const u32 q, const u32 ptw, u32* __restrict__ ovf) {
    if (q >= ptw) { __hip_atomic_store(ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return ~0u; }
    for (u32 spin = 0; spin < (1u << 22); ++spin) {   // (bounded: a page that never comes ends as an overflow, not a hang)
        const u32 x = __hip_atomic_load(&pt[row + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x) return x - 1;
        if (__hip_atomic_load(ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return ~0u;   // (its allocator gave up)
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(ovf, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ~0u;
}

// One paged level of the partition: `subs` sub-lists per bucket (list = bucket * subs + sub), each a sequence of
// pages drawn from the pool of its sub-list class.
struct PartLists {
    u32* pools;      // `subs` page counters, one memory line each
    u32* cursors;    // entries reserved so far, one memory line per list
    u32* pt;         // lists x ptw: physical page + 1 (0: not allocated yet)
    u32* owner;      // per physical page: list * ptw + index in the list's page table + 1 (0: unused)
    u32 ptw, pool_pages, subs;
};
// reserve `cnt` (1 .. 4096) consecutive places in list L: v = first place; g0 / g1 = physical pages of the first and
// the last place (~0 after an overflow).  The reservation that covers a page's first place allocates that page and
// publishes it BEFORE it waits for anything.
__device__ inline void part_reserve(const PartLists& pl, const u32 L, const u32 sub, const u32 cnt, u32* __restrict__ ovf, u32& v,
                                    u32& g0, u32& g1) {
    v = atomicAdd(&pl.cursors[(size_t)L * P1_LINE], cnt);
    const u32 q0 = v >> P1_PLOG, q1 = (v + cnt - 1) >> P1_PLOG;
    const u32 mine = (v & (P1_PAGE - 1)) == 0 ? q0 : (q1 != q0 ? q1 : ~0u);
    if (mine != ~0u) {
        bool ok = false;
        if (mine < pl.ptw) {
            const u32 ph = atomicAdd(&pl.pools[(size_t)sub * P1_LINE], 1u);
            if (ph < pl.pool_pages) {
                pl.owner[sub * pl.pool_pages + ph] = L * pl.ptw + mine + 1;   // (read by the next level, after this kernel)
                __hip_atomic_store(&pl.pt[(size_t)L * pl.ptw + mine], sub * pl.pool_pages + ph + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                ok = true;
            }
        }
        if (!ok) __hip_atomic_store(ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");   // (every lane has published its page before any lane starts to wait)
    g0 = part_wait_page(pl.pt, L * pl.ptw, q0, pl.ptw, ovf);
    g1 = q1 == q0 ? g0 : part_wait_page(pl.pt, L * pl.ptw, q1, pl.ptw, ovf);
}

// The overflow word as ONE decision per workgroup.  Other workgroups of the same launch raise it while this one runs;
// waves that read it on their own could disagree, and barriers skip exited waves — the survivors would go on with LDS
// the exited waves never wrote (addresses among it).  Thread 0 reads, everybody branches on the copy in LDS.
__device__ inline bool part_ovf_uniform(const u32* __restrict__ ovf) {
    __shared__ u32 s_ovf_seen;
    if (threadIdx.x == 0) s_ovf_seen = __hip_atomic_load(ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return s_ovf_seen != 0;
}

// source of the first entry of every chunk (tbl[nchunks] = the last source): the bisection is done once, by a
// kernel of its own, instead of sitting at the start of every chunk's critical path
__global__ void k_part_src(const u64* __restrict__ off, const u32 n_sources, const u32 nchunks, u32* __restrict__ tbl,
                           u64* __restrict__ scal, const u32 nbuckets) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) part_prep(scal, nbuckets);   // (the multiplier of this build, from the key range k_max_last found)
    if (c > nchunks) return;
    tbl[c] = c < nchunks ? part_source_of(off, n_sources, (u64)c * P1_CH) : n_sources - 1;
}

// SORTED: the chunk is put in level-1 bucket order in LDS before it is written, so that the ~16 entries a chunk sends to a
// bucket leave as ONE run from neighbouring lanes.  For inputs whose sources are so short that consecutive entries almost
// never share a bucket (1 M read groups of <= 256 hashes over 256 buckets: runs of one entry): written where they stand,
// every store instruction of a wave scattered 64 single entries over 64 pages — 5.8 GB written for 1.8 GB of entries.
template <class V, bool SORTED = false>
__global__ __launch_bounds__(P1_THREADS) void k_part1(const u64* __restrict__ keys, const u64* __restrict__ off,
                                                      const u32 n_sources, const u32 n, u64* __restrict__ scal,
                                                      const int sh1, const int pb2, const u32 nbm1, const PartLists pl,
                                                      const u32* __restrict__ src_tbl, u64* __restrict__ Kp,
                                                      V* __restrict__ Tp, u8* __restrict__ Dp) {
    __shared__ __attribute__((aligned(16))) u32 s_src[P1_CH];
    __shared__ u32 s_hist[256], s_base[256], s_pg0[256], s_pg1[256];
    __shared__ u32 s_wmax[P1_THREADS / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 c0 = blockIdx.x * P1_CH, cn = min(P1_CH, n - c0);
    u64 key[P1_EPT];
#pragma unroll
    for (u32 j = 0; j < P1_EPT; ++j) {
        const u32 idx = j * P1_THREADS + tid;
        key[j] = idx < cn ? __builtin_nontemporal_load(keys + c0 + idx) : 0;   // (read once)
    }
    const u64 mult = scal[PC_MULT];
    const u32 ident = reinterpret_cast<const u32*>(scal + PC_MODE)[0];
    const u32 rsub = blockIdx.x & (pl.subs - 1);              // sub-list class: workgroups b and b + 8 share an XCD
                                                              // (one page pool per class: a single word would see
                                                              //  every page allocation of the launch)
    u32* const ovf = reinterpret_cast<u32*>(scal + PC_OVF);
    if (tid < 256) s_hist[tid] = 0;
    // first source of this chunk and of the next (the last entry's source is that one or an earlier one)
    const u32 s_lo = src_tbl[blockIdx.x], s_hi = src_tbl[blockIdx.x + 1];
    __syncthreads();   // (the histogram is zero before any wave counts into it)
    if (s_lo != s_hi) {   // several sources in this chunk: source of every entry = prefix maximum of the run starts
        for (u32 i = tid; i < P1_CH; i += P1_THREADS) s_src[i] = 0;
        __syncthreads();
        if (tid == 0) s_src[0] = s_lo;
        for (u32 s = s_lo + 1 + tid; s <= s_hi; s += P1_THREADS) {
            const u64 o = off[s];   // (> c0: s_lo is the last source that starts at or before the chunk)
            if (off[s + 1] > o && o < (u64)c0 + cn) s_src[(u32)(o - c0)] = s;   // (non-empty sources start at distinct entries)
        }
        __syncthreads();
        uint4 a = reinterpret_cast<uint4*>(s_src)[2 * tid], b = reinterpret_cast<uint4*>(s_src)[2 * tid + 1];
        u32 v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (u32 q = 1; q < 8; ++q) v[q] = max(v[q], v[q - 1]);
        u32 inc = v[7];
        for (int o = 1; o < 64; o <<= 1) { const u32 up = __shfl_up(inc, o); if ((int)lane >= o) inc = max(inc, up); }
        if (lane == 63) s_wmax[wv] = inc;
        __syncthreads();
        u32 before = __shfl_up(inc, 1);
        if (lane == 0) before = 0;
        for (u32 w = 0; w < wv; ++w) before = max(before, s_wmax[w]);
#pragma unroll
        for (u32 q = 0; q < 8; ++q) v[q] = max(v[q], before);
        reinterpret_cast<uint4*>(s_src)[2 * tid] = make_uint4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<uint4*>(s_src)[2 * tid + 1] = make_uint4(v[4], v[5], v[6], v[7]);
        __syncthreads();
    }
    // runs of equal level-1 bucket among 64 consecutive entries: the head lane counts the run
    u32 info[P1_EPT];   // level-1 bucket | digit byte << 8 | place inside (workgroup, bucket) << 16
#pragma unroll
    for (u32 j = 0; j < P1_EPT; ++j) {
        const u32 idx = j * P1_THREADS + tid;
        const bool valid = idx < cn;
        const u32 b = part_bucket(key[j], mult, ident, nbm1);
        const u32 d1 = valid ? (b >> sh1) : 0x1FFu, d2 = b & ((1u << pb2) - 1u);
        const u32 prev = __shfl_up(d1, 1);
        const bool head = lane == 0 || prev != d1;
        const unsigned long long hm = __ballot(head);
        const unsigned long long upto = (2ull << lane) - 1ull;   // lanes <= mine (lane 63: all ones)
        const u32 hl = 63u - (u32)__builtin_clzll(hm & upto);
        const unsigned long long above = hm & ~upto;
        const u32 nxt = above ? (u32)__builtin_ctzll(above) : 64u;
        u32 rb = 0;
        if (head && valid) rb = atomicAdd(&s_hist[d1], nxt - lane);
        rb = __shfl(rb, hl);
        info[j] = (d1 & 0xFFu) | (d2 << 8) | ((rb + lane - hl) << 16);
    }
    __syncthreads();
    if (tid < 256) {   // one reservation per (workgroup, level-1 bucket); the one that crosses a page start allocates the page
        const u32 cnt = s_hist[tid];
        u32 v = 0, g0 = ~0u, g1 = ~0u;
        if (cnt) part_reserve(pl, tid * pl.subs + rsub, rsub, cnt, ovf, v, g0, g1);
        s_base[tid] = v; s_pg0[tid] = g0; s_pg1[tid] = g1;
    }
    __syncthreads();
    if (SORTED) {
        __shared__ u64 s_key[SORTED ? P1_CH : 1];
        __shared__ V s_tg[SORTED ? P1_CH : 1];
        __shared__ u8 s_dg[SORTED ? P1_CH : 1], s_bn[SORTED ? P1_CH : 1];
        __shared__ u32 s_ls[SORTED ? 256 : 1];
        if (wv == 0) {   // where every bucket's run starts inside the chunk
            u32 c[4], t = 0;
#pragma unroll
            for (u32 i = 0; i < 4; ++i) { c[i] = s_hist[4 * lane + i]; t += c[i]; }
            u32 inc = t;
            inc = wave_scan_add(inc);
            u32 run = inc - t;
#pragma unroll
            for (u32 i = 0; i < 4; ++i) { s_ls[4 * lane + i] = run; run += c[i]; }
        }
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < P1_EPT; ++j) {
            const u32 idx = j * P1_THREADS + tid;
            if (idx >= cn) continue;
            const u32 d1 = info[j] & 0xFFu, slot = s_ls[d1] + (info[j] >> 16);
            const u32 src = s_lo == s_hi ? s_lo : s_src[idx];
            s_key[slot] = key[j];
            s_tg[slot] = make_tag<V>(((src / TB) << 8) | (src % TB), 0u);
            s_dg[slot] = (u8)(info[j] >> 8);
            s_bn[slot] = (u8)d1;
        }
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < P1_EPT; ++j) {
            const u32 i = j * P1_THREADS + tid;
            if (i >= cn) continue;
            const u32 d1 = s_bn[i], base = s_base[d1], v = base + (i - s_ls[d1]);
            const u32 ph = (v >> P1_PLOG) == (base >> P1_PLOG) ? s_pg0[d1] : s_pg1[d1];
            if (ph == ~0u) continue;   // (overflow: the build is repeated with the library partition)
            const size_t a = ((size_t)ph << P1_PLOG) | (v & (P1_PAGE - 1));
            Kp[a] = s_key[i];
            Tp[a] = s_tg[i];
            Dp[a] = s_dg[i];
        }
        return;
    }
#pragma unroll
    for (u32 j = 0; j < P1_EPT; ++j) {
        const u32 idx = j * P1_THREADS + tid;
        if (idx >= cn) continue;
        const u32 d1 = info[j] & 0xFFu, base = s_base[d1], v = base + (info[j] >> 16);
        const u32 ph = (v >> P1_PLOG) == (base >> P1_PLOG) ? s_pg0[d1] : s_pg1[d1];
        if (ph == ~0u) continue;   // (overflow: the build is repeated with the library partition)
        const size_t a = ((size_t)ph << P1_PLOG) | (v & (P1_PAGE - 1));
        const u32 src = s_lo == s_hi ? s_lo : s_src[idx];
        Kp[a] = key[j];
        Tp[a] = make_tag<V>(((src / TB) << 8) | (src % TB), 0u);
        Dp[a] = (u8)(info[j] >> 8);
    }
}

// level 2, counting: one workgroup per page — entries per final bucket from the page's 4 096 digit bytes
// (an LDS histogram, then one global add per bucket present).  gcnt is zero at launch.
__global__ __launch_bounds__(256) void k_hist2(const u64* __restrict__ scal, const PartLists pl, const int pb2,
                                               const u8* __restrict__ Dp, u32* __restrict__ gcnt) {
    __shared__ u32 s_hist[256];
    const u32 ow = pl.owner[blockIdx.x];
    if (!ow || reinterpret_cast<const u32*>(scal + PC_OVF)[0]) return;   // (after an overflow: nothing is counted, every bucket stays empty)
    const u32 L = (ow - 1) / pl.ptw, q = (ow - 1) % pl.ptw, len = pl.cursors[(size_t)L * P1_LINE];
    if (len <= (q << P1_PLOG)) return;
    const u32 m = min(P1_PAGE, len - (q << P1_PLOG)), tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    const u32 i0 = 16 * tid;
    if (i0 < m) {
        const uint4 w4 = *reinterpret_cast<const uint4*>(Dp + ((size_t)blockIdx.x << P1_PLOG) + i0);
        const u32 w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (u32 k = 0; k < 16; ++k)
            if (i0 + k < m) atomicAdd(&s_hist[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu], 1u);
    }
    __syncthreads();
    const u32 c = s_hist[tid];
    if (c) atomicAdd(&gcnt[((L / pl.subs) << pb2) + tid], c);
}

// entries in front of every bucket prefix (= group of `subs` lists of the last paged level): one workgroup, exclusive
// scan of the groups' totals; gbase[ngroups] = all entries (every entry was counted exactly once by its list's cursor)
__global__ __launch_bounds__(1024) void k_group_base(u64* __restrict__ scal, const PartLists pl, const u32 ngroups, const u32 n,
                                                     u32* __restrict__ gbase) {
    __shared__ u32 s_w[16], s_carry;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool bad = reinterpret_cast<const u32*>(scal + PC_OVF)[0] != 0;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (u32 g0 = 0; g0 < ngroups; g0 += 1024) {
        const u32 g = g0 + tid;
        u32 tot = 0;
        if (g < ngroups && !bad)
            for (u32 r = 0; r < pl.subs; ++r) tot += pl.cursors[(size_t)(g * pl.subs + r) * P1_LINE];
        u32 inc = tot;
        inc = wave_scan_add(inc);
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        u32 run = s_carry + inc - tot;
        for (u32 w = 0; w < wv; ++w) run += s_w[w];
        if (g < ngroups) gbase[g] = run;
        __syncthreads();
        if (tid == 1023) s_carry = run + tot;
        __syncthreads();
    }
    if (tid == 0) {
        gbase[ngroups] = s_carry;
        if (!bad && s_carry != n) reinterpret_cast<u32*>(scal + PC_OVF)[0] = 2;   // (cannot happen: a defect, reported as such)
    }
}

// the exact start of every final bucket in the dense output: one workgroup per bucket prefix scans its <= 256
// buckets.  Writes bstart (what k_bucket_bounds used to search for) and turns gcnt into the cursors of the scatter.
__global__ __launch_bounds__(256) void k_scan2(const u32* __restrict__ gbase, const int pb2, const u32 nbuckets, const u32 ngroups,
                                               u32* __restrict__ gcnt, u32* __restrict__ bstart) {
    __shared__ u32 s_w[4];
    const u32 B = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 b = (B << pb2) + tid;
    const bool in = tid < (1u << pb2) && b < nbuckets;
    const u32 c = in ? gcnt[b] : 0u;
    u32 inc = c;
    inc = wave_scan_add(inc);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    u32 run = gbase[B] + inc - c;
    for (u32 w = 0; w < wv; ++w) run += s_w[w];
    if (in) { bstart[b] = run; gcnt[b] = run; }
    if (B == ngroups - 1 && tid == 0) bstart[nbuckets] = gbase[ngroups];
}

// level 2, scatter: one workgroup per page.  owner[page] = (sub-list, index in its page table) + 1, 0 = unused.
template <class V>
__global__ __launch_bounds__(P2_THREADS) void k_scatter2(const u64* __restrict__ scal, const PartLists pl, const int pb2,
                                                         const u32 nbm1, const u64* __restrict__ Kp,
                                                         const V* __restrict__ Tp, u32* __restrict__ gcur,
                                                         u64* __restrict__ K2, V* __restrict__ T2) {
    __shared__ u64 s_key[P2_TILE];
    __shared__ V s_tag[P2_TILE];
    __shared__ u8 s_bin[P2_TILE];
    __shared__ u32 s_ls[256], s_cnt[256], s_gb[256];
    const u32 ow = pl.owner[blockIdx.x];
    if (!ow || reinterpret_cast<const u32*>(scal + PC_OVF)[0]) return;   // (after an overflow the build is repeated)
    const u32 L = (ow - 1) / pl.ptw, q = (ow - 1) % pl.ptw, len = pl.cursors[(size_t)L * P1_LINE];
    if (len <= (q << P1_PLOG)) return;
    const u32 m = min(P1_PAGE, len - (q << P1_PLOG));
    const u32 B = L / pl.subs, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 nb2m1 = (1u << pb2) - 1u;
    const u64 mult = scal[PC_MULT];
    const u32 ident = reinterpret_cast<const u32*>(scal + PC_MODE)[0];
    const size_t page = (size_t)blockIdx.x << P1_PLOG;
    u64 key[P2_EPT];
    V tag[P2_EPT];
    u32 rk[P2_EPT];
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        key[k] = 0; tag[k] = V(0);
        if (i < m) { key[k] = __builtin_nontemporal_load(Kp + page + i); tag[k] = __builtin_nontemporal_load(Tp + page + i); }
    }
    if (tid < 256) s_cnt[tid] = 0;
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        rk[k] = part_bucket(key[k], mult, ident, nbm1) & nb2m1;
        if (i < m) rk[k] |= atomicAdd(&s_cnt[rk[k]], 1u) << 8;
    }
    __syncthreads();
    if (wv == 0) {   // where every bucket's run starts inside the tile
        u32 c[4], t = 0;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { c[i] = s_cnt[4 * lane + i]; t += c[i]; }
        u32 inc = t;
        inc = wave_scan_add(inc);
        u32 run = inc - t;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { s_ls[4 * lane + i] = run; run += c[i]; }
    } else if (tid - 64 < 256) {   // ... and in the dense output: one atomic per bucket present in the tile
        const u32 bin = tid - 64, c = s_cnt[bin];
        if (c) s_gb[bin] = atomicAdd(&gcur[(B << pb2) + bin], c);
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = rk[k] & 0xFFu, slot = s_ls[bin] + (rk[k] >> 8);
            s_key[slot] = key[k];
            s_tag[slot] = tag[k];
            s_bin[slot] = (u8)bin;
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = s_bin[i], dst = s_gb[bin] + (i - s_ls[bin]);
            K2[dst] = s_key[i];
            T2[dst] = s_tag[i];
        }
    }
}

// middle level (sets of more than 65 536 buckets): one workgroup per level-1 page.  The page is ordered by the next
// `pbm` bits of the bucket id in LDS, every group of entries reserves its run in the list of its (level-1 bucket,
// middle digit) — pages from a second pool, same reservation as level 1 — and the runs (~4 096 / 2^pbm entries)
// are written out: keys, tags and digit bytes.  The final level then works on these pages.
template <class V>
__global__ __launch_bounds__(P2_THREADS) void k_part_mid(const u64* __restrict__ scal, const PartLists pa, const PartLists pm,
                                                         const int pb2, const int pbm, const u32 nbm1,
                                                         const u64* __restrict__ Ka, const V* __restrict__ Ta,
                                                         const u8* __restrict__ Da, u64* __restrict__ Km,
                                                         V* __restrict__ Tm, u8* __restrict__ Dm) {
    __shared__ u64 s_key[P2_TILE];
    __shared__ V s_tag[P2_TILE];
    __shared__ u8 s_dig[P2_TILE], s_bin[P2_TILE];
    __shared__ u32 s_ls[256], s_cnt[256], s_v[256], s_g0[256], s_g1[256];
    const u32 ow = pa.owner[blockIdx.x];
    u32* const ovf = const_cast<u32*>(reinterpret_cast<const u32*>(scal + PC_OVF));
    if (part_ovf_uniform(ovf) || !ow) return;
    const u32 L = (ow - 1) / pa.ptw, q = (ow - 1) % pa.ptw, len = pa.cursors[(size_t)L * P1_LINE];
    if (len <= (q << P1_PLOG)) return;
    const u32 m = min(P1_PAGE, len - (q << P1_PLOG));
    const u32 BA = L / pa.subs, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 nmid = 1u << pbm, sub = blockIdx.x & (pm.subs - 1);
    const u64 mult = scal[PC_MULT];
    const u32 ident = reinterpret_cast<const u32*>(scal + PC_MODE)[0];
    const size_t page = (size_t)blockIdx.x << P1_PLOG;
    u64 key[P2_EPT];
    V tag[P2_EPT];
    u32 rk[P2_EPT], dig[P2_EPT];
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        key[k] = 0; tag[k] = V(0); dig[k] = 0;
        if (i < m) {
            key[k] = __builtin_nontemporal_load(Ka + page + i);
            tag[k] = __builtin_nontemporal_load(Ta + page + i);
            dig[k] = Da[page + i];
        }
    }
    if (tid < 256) s_cnt[tid] = 0;
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        rk[k] = (part_bucket(key[k], mult, ident, nbm1) >> pb2) & (nmid - 1);
        if (i < m) rk[k] |= atomicAdd(&s_cnt[rk[k]], 1u) << 8;
    }
    __syncthreads();
    if (wv == 0) {
        u32 c[4], t = 0;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { c[i] = s_cnt[4 * lane + i]; t += c[i]; }
        u32 inc = t;
        inc = wave_scan_add(inc);
        u32 run = inc - t;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { s_ls[4 * lane + i] = run; run += c[i]; }
    } else if (tid - 64 < 256) {   // one reservation per middle digit present in the page
        const u32 bin = tid - 64, c = s_cnt[bin];
        u32 v = 0, g0 = ~0u, g1 = ~0u;
        if (c) part_reserve(pm, ((BA << pbm) | bin) * pm.subs + sub, sub, c, ovf, v, g0, g1);
        s_v[bin] = v; s_g0[bin] = g0; s_g1[bin] = g1;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = rk[k] & 0xFFu, slot = s_ls[bin] + (rk[k] >> 8);
            s_key[slot] = key[k];
            s_tag[slot] = tag[k];
            s_dig[slot] = (u8)dig[k];
            s_bin[slot] = (u8)bin;
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = s_bin[i], base = s_v[bin], v = base + (i - s_ls[bin]);
            const u32 ph = (v >> P1_PLOG) == (base >> P1_PLOG) ? s_g0[bin] : s_g1[bin];
            if (ph != ~0u) {
                const size_t a = ((size_t)ph << P1_PLOG) | (v & (P1_PAGE - 1));
                Km[a] = s_key[i];
                Tm[a] = s_tag[i];
                Dm[a] = s_dig[i];
            }
        }
    }
}

// ---- level 1 without a copy: the segments of the sorted runs (round 2) ------------------------------------------
// Every source's run is sorted and the bucket id is monotone in the key, so the entries of one source that belong to
// one level-1 bucket ("range": 2^pb2 consecutive final buckets) ARE a contiguous segment of its run — the sketch array
// already is partitioned at level 1, source by source.  Instead of copying the entries into level-1 pages (k_part1:
// 8 bytes read + 11 written per entry, then 11 read again), only the segment boundaries are written down
// (k_seg_bounds: one streaming read of the keys, (ranges + 1) words per source), and the final scatter gathers its
// tile straight from the sketches: workgroup (group of consecutive sources g, range B) reads the B-segments of its
// sources — ~4 096 entries — and from there on is k_scatter2 (LDS counting sort by final bucket, one reservation per
// bucket present, runs written out).  There is no histogram pass either: every final bucket owns a fixed number of
// places (half as many again as the mean) and its cursor counts what arrived; a bucket or a tile that does not fit
// raises the overflow word and the build is repeated with the paged partition above (the engine remembers).
// Bytes per entry (8-byte key, 2-byte tag): 8 read + ~10 read (segments of ~200 bytes end in partial lines) + 10
// written = 28, against 40.  For runs long enough to leave segments of a dozen entries or more (sourmash-style
// sketches: C2, 5 000 hashes in 196 ranges); everything else keeps the paged levels.
constexpr u32 SEG_SMAX = 512;     // sources per group (LDS: start slot and address of every segment)
constexpr u32 SEG_FILL = 3072;    // mean entries of a tile; the tile holds P2_TILE (uniform hashes: +18 sigma)
constexpr u32 SEG_MIN_LEN = 12;   // mean segment length from which the path is used

// the multiplier of this build and, per range j, the smallest key that belongs to range j or a later one:
// umulhi(key, mult) >= j << pb2  <=>  key >= ceil((j << pb2) * 2^64 / mult)   (kmin[0] = 0; j << pb2 < nbuckets <= mult)
__global__ __launch_bounds__(256) void k_seg_prep(u64* __restrict__ scal, const u32 nbuckets, const int pb2, const u32 nb1,
                                                  u64* __restrict__ kmin) {
    if (threadIdx.x == 0) part_prep(scal, nbuckets);
    __threadfence_block();
    __syncthreads();
    const u64 mult = scal[PC_MULT];
    const u32 ident = reinterpret_cast<const u32*>(scal + PC_MODE)[0];
    for (u32 j = threadIdx.x; j <= nb1; j += 256) {
        const u64 T = (u64)j << pb2;
        u64 k = T;   // (identity buckets: bucket = min(key, nbuckets - 1))
        if (j >= nb1) k = ~0ull;   // (sentinel: never read as a boundary)
        else if (!ident && T) {
            u64 rem = T, q = 0;   // (T : 0) / mult, T < mult
            for (int i = 63; i >= 0; --i) {
                const u64 carry = rem >> 63;
                rem <<= 1;
                if (carry || rem >= mult) { rem -= mult; q |= 1ull << i; }
            }
            k = q + (rem ? 1ull : 0ull);
        }
        kmin[j] = k;
    }
}

// one workgroup (four waves) per 2 048 consecutive entries of a source (the chunks come from a host table that is
// rebuilt when the offsets change).  Window w of the chunk goes to wave w % 4, so every load
// instruction of the workgroup covers 2 KB of consecutive memory and all eight of a wave are requested together.  The
// keys of a window ascend, so the window crosses the boundaries between the range of the entry in front of it (the last
// key of the previous window: exchanged through LDS) and the range of its own last key — uniform 64 x 64 multiplies on
// the scalar unit instead of one per key — and boundary j sits behind the keys below kmin[j]: one compare + ballot
// each.  bnd[s][j] = first entry of range j or a later one; bnd[s][0] = 0, bnd[s][nb1] = length.
constexpr u32 SEG_BW = 4;          // waves per workgroup
constexpr u32 SEG_BWIN = 8;        // windows of 64 entries per wave
constexpr u32 SEG_BPART = 64 * SEG_BWIN * SEG_BW;   // entries per workgroup
__device__ inline u32 seg_range_of(const u64 key, const u64 mult, const u32 ident, const u32 nbm1, const int pb2) {
    return part_bucket(key, mult, ident, nbm1) >> pb2;
}
__global__ __launch_bounds__(64 * SEG_BW) void k_seg_bounds(const u64* __restrict__ keys, const uint4* __restrict__ chunks,
                                                            const u64* __restrict__ scal, const u64* __restrict__ kmin, const int pb2,
                                                            const u32 nbm1, const u32 nb1, u32* __restrict__ bnd) {
    __shared__ u32 s_rl[SEG_BW * SEG_BWIN + 1];   // [1 + w]: range of the last key of window w; [0]: of the entry in front of the chunk
    __shared__ u32 s_bnd[260];                    // the boundaries this chunk crosses: they leave in one piece (single words per
                                                  // boundary were 2 M partial-line writes per build)
    const u32 lane = threadIdx.x & 63;
    const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this workgroup's chunk (host table: no empty workgroups, no offsets to fetch first):
    // first entry in the sketch array, position inside its run, source, entries | last chunk of the run << 31
    const uint4 ch = chunks[blockIdx.x];
    const u64 b = (u64)ch.x - ch.y;   // start of the run
    const u32 s = ch.z, a0 = ch.y, a1 = a0 + (ch.w & 0x7FFFFFFFu);
    const bool last_chunk = (ch.w >> 31) != 0;
    const u32 len = a1;   // (only used by the last chunk)
    const u64 mult = scal[PC_MULT];
    const u32 ident = reinterpret_cast<const u32*>(scal + PC_MODE)[0];
    u32* const row = bnd + (size_t)s * (nb1 + 1);
    if (a0 == 0 && threadIdx.x == 0) row[0] = 0;
    (void)kmin;   // (the boundary keys of round 2's scalar formulation; k_seg_scatter does not need them either)
    u64 k[SEG_BWIN];
#pragma unroll
    for (u32 q = 0; q < SEG_BWIN; ++q) {
        const u32 i = a0 + (q * SEG_BW + wv) * 64 + lane;
        k[q] = i < a1 ? __builtin_nontemporal_load(keys + b + i) : 0;
    }
    if (wv == 0) {
        u32 c0 = 0;
        if (a0) {
            const u64 kp = keys[b + a0 - 1];
            c0 = seg_range_of((u64)(u32)__builtin_amdgcn_readfirstlane((u32)kp) | ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(kp >> 32)) << 32), mult,
                              ident, nbm1, pb2);
        }
        if (lane == 0) s_rl[0] = c0;
    }
    // Every lane works out the range of its own keys on the VECTOR unit (a 64 x 64 -> high 64 multiply each).  Round 2 did
    // this once per window on the scalar unit (the keys of a window ascend: only its last key matters) and then looped over
    // the boundaries with compare + ballot + popcount — ~70 scalar instructions per window, and a compute unit issues ONE
    // scalar instruction per cycle for all its 32 waves: the counters showed 6.5e7 scalar against 2.2e7 vector instructions
    // per launch, 554 per wave x 32 resident waves = the waves' whole lifetime.  The scalar unit was the kernel's limit.
    u32 r[SEG_BWIN];
#pragma unroll
    for (u32 q = 0; q < SEG_BWIN; ++q) {
        const u32 w = q * SEG_BW + wv, w0 = a0 + w * 64;
        r[q] = seg_range_of(k[q], mult, ident, nbm1, pb2);
        if (w0 < a1 && lane == min(64u, a1 - w0) - 1) s_rl[1 + w] = r[q];   // the window's last key
    }
    __syncthreads();
#pragma unroll
    for (u32 q = 0; q < SEG_BWIN; ++q) {
        const u32 w = q * SEG_BW + wv, w0 = a0 + w * 64;
        if (w0 >= a1) break;   // (uniform)
        const u32 i = w0 + lane;
        // the range of the entry in front of mine: my left neighbour's, or (lane 0) the last key of the window before
        const u32 up = __shfl_up(r[q], 1);
        const u32 prev = lane ? up : s_rl[w];
        if (i < a1)
            for (u32 j = prev + 1; j <= r[q]; ++j) s_bnd[j] = i;   // I am the first entry of range j or a later one (ranges ascend with the keys)
        if (w0 + 64 >= a1 && last_chunk) {   // the run's last window: the ranges behind its last key are empty, they begin at the end
            const u32 rlast = __shfl(r[q], min(64u, a1 - w0) - 1);
            for (u32 j = rlast + 1 + lane; j <= nb1; j += 64) row[j] = len;
        }
    }
    __syncthreads();
    {   // boundaries (range in front of the chunk, range of its last key]: written by exactly one window each
        const u32 c0 = s_rl[0], nw = (a1 - a0 + 63) / 64, c1 = nw ? s_rl[nw] : c0;
        for (u32 j = c0 + 1 + threadIdx.x; j <= c1; j += 64 * SEG_BW) row[j] = s_bnd[j];
    }
    if (a1 == 0 && wv == 0)   // (an empty run has one chunk of no entries)
        for (u32 j = 1 + lane; j <= nb1; j += 64) row[j] = 0;
}

// Tile (group of sources g, range B) of a workgroup of k_seg_scatter / k_seg_mid (grid = groups x nb1).  Workgroups are
// handed to the eight XCDs (one L2 each) in turn.  An XCD gets a STRIP of ranges, for every group of sources: the tiles
// (g, B), (g, B + 1) — whose segments meet inside a memory line — follow each other on the same XCD, and so do (g, B),
// (g + 1, B), whose runs meet inside the lines of the same buckets (partial lines merge in that L2 instead of leaving
// eight of them one by one).  Bijective for any grid; only the last strip is narrower.
__device__ inline void seg_tile_of(const u32 nb1, u32& g, u32& B) {
    const u32 nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7u, x = blockIdx.x & 7u;
    const u32 vid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + (blockIdx.x >> 3);
    const u32 ng = nwg / nb1, W = max(1u, nb1 >> 3), t = min(vid / (ng * W), (nb1 - 1) / W);
    const u32 rem = vid - t * ng * W, wt = min(W, nb1 - t * W);
    g = rem / wt;
    B = t * W + rem % wt;
}

// Segment of every entry of a thread (entry i lies in segment sg with s_pre[sg] <= i < s_pre[sg + 1]; s_pre[ns] = m is
// above every entry): all of the thread's searches advance in step, so a halving costs ONE round trip to LDS with
// P2_EPT reads in flight.  (Round 2 bisected entry by entry: loops of data-dependent length the compiler cannot
// interleave — P2_EPT x log2(ns) dependent LDS reads, ~60 round trips per thread in front of the key loads.)
template <u32 EPT, u32 NT>
__device__ inline void seg_of_entries(const u32* s_pre, const u32 ns, const u32 tid, u32 (&sg)[EPT]) {
#pragma unroll
    for (u32 k = 0; k < EPT; ++k) sg[k] = 0;
    u32 st = ns > 1 ? 1u << (31 - __builtin_clz(ns - 1)) : 0u;   // (uniform: the largest power of two below ns)
    for (; st; st >>= 1) {
#pragma unroll
        for (u32 k = 0; k < EPT; ++k) {
            const u32 c = min(sg[k] + st, ns);
            if (s_pre[c] <= k * NT + tid) sg[k] = c;
        }
    }
}

// one workgroup per (group of sources, range): gather the tile from the segments, then the counting sort of k_scatter2.
// gcur[b] (zero at launch) counts the entries of final bucket b, which owns the places [b x cap, (b + 1) x cap).
template <class V>
__global__ __launch_bounds__(P2_THREADS) void k_seg_scatter(const u64* __restrict__ keys, const u64* __restrict__ off,
                                                            const u32* __restrict__ bnd, const u32* __restrict__ groups,
                                                            u64* __restrict__ scal, const int pb2, const u32 nbm1, const u32 nb1,
                                                            const u32 cap, u32* __restrict__ gcur,
                                                            u64* __restrict__ K2, V* __restrict__ T2) {
    __shared__ u64 s_key[P2_TILE];
    __shared__ V s_tag[P2_TILE];
    __shared__ u8 s_bin[P2_TILE];
    __shared__ u32 s_ls[256], s_cnt[256], s_gb[256];
    __shared__ u32 s_pre[SEG_SMAX + 1], s_addr[SEG_SMAX];
    __shared__ u32 s_w[P2_THREADS / 64];
    u32* const ovf = reinterpret_cast<u32*>(scal + PC_OVF);
    if (part_ovf_uniform(ovf)) return;
    ST_BEGIN();
    u32 g, B;
    seg_tile_of(nb1, g, B);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 s0 = groups[g], ns = groups[g + 1] - s0;        // (<= SEG_SMAX by construction)
    // the segments: start address and length, exclusive scan of the lengths = start slot inside the tile
    u32 len = 0;
    if (tid < ns) {
        const u32* row = bnd + (size_t)(s0 + tid) * (nb1 + 1) + B;
        const u32 lo = row[0], hi = row[1];
        len = hi - lo;
        s_addr[tid] = (u32)off[s0 + tid] + lo;   // (fewer than 2^30 entries per build)
    }
    u32 inc = len;
    inc = wave_scan_add(inc);
    if (lane == 63) s_w[wv] = inc;
    if (tid < 256) s_cnt[tid] = 0;
    __syncthreads();
    ST_T(48);
    u32 run = inc - len, m = 0;
    for (u32 w = 0; w < P2_THREADS / 64; ++w) { if (w < wv) run += s_w[w]; m += s_w[w]; }
    if (tid < ns) s_pre[tid] = run;
    if (tid == 0) s_pre[ns] = m;
    if (m > P2_TILE) {   // (keys far from uniform: the paged partition takes this build)
        if (tid == 0) __hip_atomic_store(ovf, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __syncthreads();
    ST_T(49);
    // gather straight into registers: entry i of the tile lies in the segment sg with s_pre[sg] <= i < s_pre[sg + 1]
    // (bisection in LDS; neighbouring threads read neighbouring entries of the same segment).  All loads of a thread
    // are in flight together: a loop over the segments would be one memory round trip per segment.
    const u32 nb2m1 = (1u << pb2) - 1u;
    const u64 mult = scal[PC_MULT];
    const u32 ident = reinterpret_cast<const u32*>(scal + PC_MODE)[0];
    u64 key[P2_EPT];
    V tag[P2_EPT];
    u32 rk[P2_EPT];
    u32 sg[P2_EPT];
    seg_of_entries<P2_EPT, P2_THREADS>(s_pre, ns, tid, sg);
    ST_T(53);
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        key[k] = 0; tag[k] = V(0);
        if (i < m) {
            const u32 lo = sg[k];
            key[k] = __builtin_nontemporal_load(keys + s_addr[lo] + (i - s_pre[lo]));
            const u32 src = s0 + lo;
            tag[k] = make_tag<V>(((src / TB) << 8) | (src % TB), 0u);
        }
    }
#ifdef KSP_FKTIME
    __builtin_amdgcn_s_waitcnt(0x0F70); ST_T(54);   // (timing build: the key loads alone)
#endif
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        rk[k] = part_bucket(key[k], mult, ident, nbm1) & nb2m1;
        if (i < m) rk[k] |= atomicAdd(&s_cnt[rk[k]], 1u) << 8;
    }
    ST_T(55);
    __syncthreads();
    ST_T(50);   // (also: every thread holds its entries in registers, the staging buffers are free)
    if (wv == 0) {   // where every bucket's run starts inside the tile
        u32 c[4], t = 0;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { c[i] = s_cnt[4 * lane + i]; t += c[i]; }
        u32 in2 = t;
        in2 = wave_scan_add(in2);
        u32 r2 = in2 - t;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { s_ls[4 * lane + i] = r2; r2 += c[i]; }
    } else if (tid - 64 < 256) {   // ... and among the bucket's places: one atomic per bucket present in the tile
        const u32 bin = tid - 64, c = s_cnt[bin];
        if (c) {
            const u32 bkt = (B << pb2) + bin;
            const u32 at = atomicAdd(&gcur[bkt], c);
            if (at + c > cap) __hip_atomic_store(ovf, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the stores below stay inside the arrays: see the host side)
            s_gb[bin] = bkt * cap + min(at, cap);
        }
    }
    __syncthreads();
    ST_T(51);
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = rk[k] & 0xFFu, slot = s_ls[bin] + (rk[k] >> 8);
            s_key[slot] = key[k];
            s_tag[slot] = tag[k];
            s_bin[slot] = (u8)bin;
        }
    }
    __syncthreads();
    ST_T(52);
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = s_bin[i], dst = s_gb[bin] + (i - s_ls[bin]);
            K2[dst] = s_key[i];
            T2[dst] = s_tag[i];
        }
    }
    ST_T(56);
    ST_END(48);
}

// Sets of more than 65 536 buckets (three levels): the same trick for THEIR level 1 — one workgroup per (group of
// sources, level-1 bucket) gathers its tile from the segments of the sorted runs and then is k_part_mid: the tile is
// ordered by the middle digit in LDS, every digit present reserves its run in the paged list of its (level-1 bucket,
// middle digit) and keys, tags and the final level's digit bytes are written out.  k_part1's copy of every entry
// (8 bytes read, 11 written, 11 read again by k_part_mid) becomes one boundary pass (8 read) and a gather.
template <class V>
__global__ __launch_bounds__(P2_THREADS) void k_seg_mid(const u64* __restrict__ keys, const u64* __restrict__ off,
                                                        const u32* __restrict__ bnd, const u32* __restrict__ groups,
                                                        u64* __restrict__ scal, const int pb2, const int pbm, const u32 nbm1,
                                                        const u32 nb1, const PartLists pm, u64* __restrict__ Km,
                                                        V* __restrict__ Tm, u8* __restrict__ Dm) {
    __shared__ u64 s_key[P2_TILE];
    __shared__ V s_tag[P2_TILE];
    __shared__ u8 s_dig[P2_TILE], s_bin[P2_TILE];
    __shared__ u32 s_ls[256], s_cnt[256], s_v[256], s_g0[256], s_g1[256];
    __shared__ u32 s_pre[SEG_SMAX + 1], s_addr[SEG_SMAX];
    __shared__ u32 s_w[P2_THREADS / 64];
    u32* const ovf = reinterpret_cast<u32*>(scal + PC_OVF);
    if (part_ovf_uniform(ovf)) return;
    u32 g, BA;
    seg_tile_of(nb1, g, BA);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 s0 = groups[g], ns = groups[g + 1] - s0;        // (<= SEG_SMAX by construction)
    u32 len = 0;
    if (tid < ns) {
        const u32* row = bnd + (size_t)(s0 + tid) * (nb1 + 1) + BA;
        const u32 lo = row[0], hi = row[1];
        len = hi - lo;
        s_addr[tid] = (u32)off[s0 + tid] + lo;
    }
    u32 inc = len;
    inc = wave_scan_add(inc);
    if (lane == 63) s_w[wv] = inc;
    if (tid < 256) s_cnt[tid] = 0;
    __syncthreads();
    u32 run = inc - len, m = 0;
    for (u32 w = 0; w < P2_THREADS / 64; ++w) { if (w < wv) run += s_w[w]; m += s_w[w]; }
    if (tid < ns) s_pre[tid] = run;
    if (tid == 0) s_pre[ns] = m;
    if (m > P2_TILE) {
        if (tid == 0) __hip_atomic_store(ovf, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __syncthreads();
    const u32 nmid = 1u << pbm, sub = blockIdx.x & (pm.subs - 1);
    const u64 mult = scal[PC_MULT];
    const u32 ident = reinterpret_cast<const u32*>(scal + PC_MODE)[0];
    u64 key[P2_EPT];
    V tag[P2_EPT];
    u32 rk[P2_EPT], dig[P2_EPT];
    u32 sg[P2_EPT];
    seg_of_entries<P2_EPT, P2_THREADS>(s_pre, ns, tid, sg);
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        key[k] = 0; tag[k] = V(0);
        if (i < m) {
            const u32 lo = sg[k];
            key[k] = __builtin_nontemporal_load(keys + s_addr[lo] + (i - s_pre[lo]));
            const u32 src = s0 + lo;
            tag[k] = make_tag<V>(((src / TB) << 8) | (src % TB), 0u);
        }
    }
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        const u32 b = part_bucket(key[k], mult, ident, nbm1);
        dig[k] = b & ((1u << pb2) - 1u);
        rk[k] = (b >> pb2) & (nmid - 1);
        if (i < m) rk[k] |= atomicAdd(&s_cnt[rk[k]], 1u) << 8;
    }
    __syncthreads();
    if (wv == 0) {
        u32 c[4], t = 0;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { c[i] = s_cnt[4 * lane + i]; t += c[i]; }
        u32 in2 = t;
        in2 = wave_scan_add(in2);
        u32 r2 = in2 - t;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) { s_ls[4 * lane + i] = r2; r2 += c[i]; }
    } else if (tid - 64 < 256) {   // one reservation per middle digit present in the tile
        const u32 bin = tid - 64, c = s_cnt[bin];
        u32 v = 0, g0 = ~0u, g1 = ~0u;
        if (c) part_reserve(pm, ((BA << pbm) | bin) * pm.subs + sub, sub, c, ovf, v, g0, g1);
        s_v[bin] = v; s_g0[bin] = g0; s_g1[bin] = g1;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = rk[k] & 0xFFu, slot = s_ls[bin] + (rk[k] >> 8);
            s_key[slot] = key[k];
            s_tag[slot] = tag[k];
            s_dig[slot] = (u8)dig[k];
            s_bin[slot] = (u8)bin;
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < P2_EPT; ++k) {
        const u32 i = k * P2_THREADS + tid;
        if (i < m) {
            const u32 bin = s_bin[i], base = s_v[bin], v = base + (i - s_ls[bin]);
            const u32 ph = (v >> P1_PLOG) == (base >> P1_PLOG) ? s_g0[bin] : s_g1[bin];
            if (ph != ~0u) {
                const size_t a = ((size_t)ph << P1_PLOG) | (v & (P1_PAGE - 1));
                Km[a] = s_key[i];
                Tm[a] = s_tag[i];
                Dm[a] = s_dig[i];
            }
        }
    }
}
