// Stage 2 of the engine: k_join and its device functions — rank-aligned cell search in a 4-ary LDS tree,
// LDS-counter accumulation (weighted input, single-source postings), bit-sliced accumulation (unweighted
// multi-source postings: diagonal tiles and collected matches), edge emission.
// Included by engine.hip inside namespace ksp (one translation unit).  Host side: ksp_engine_join.
#pragma once
// ------------------------------------------------------------------------------------
// stage 2: the join kernel
// ------------------------------------------------------------------------------------
struct JoinArgs {
    const u32* brk;     // block lists: distinct key ranks, ascending inside a block
    const u32* info;
    const u32* bw;      // NULL -> weight 1
    const uint4* bigmask; // 128-bit membership masks of the postings with > 4 sources
    const uint4* pmask;   // (or NULL) the membership mask of EVERY list word at its list position: the bit-sliced paths read
                          // it in place — one coalesced load instead of posting word -> mask index -> mask
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
    // match-list mode (work-list mode only; NULL: the lists are searched): stage 1 has already paired the list words
    // of every key that sits in two blocks — one record (posting word of the I side | posting word of the J side
    // << 32) per (key, block pair), sorted by tile; tile `i` of the work list owns mrec[mstart[i] .. mstart[i + 1])
    const u64* mrec;
    const u32* mstart;
#ifdef KSP_WGTIME
    unsigned long long* wgt;   // (-DKSP_WGTIME builds, tools/wg_times.py) per workgroup: start, end (100 MHz clock), I << 32 | J, sub << 32 | sp
#endif
};

#ifdef KSP_WGTIME
// (timing builds) thread 0 adds the time since its previous mark to slot k of its workgroup's row
#define WGT_ACC(a, k) do { if ((a).wgt && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); \
        unsigned long long* r_ = (a).wgt + 16 * (size_t)((a).wg0 + blockIdx.x); r_[k] += now_ - r_[15]; r_[15] = now_; } } while (0)
#else
#define WGT_ACC(a, k) do {} while (0)
#endif
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

// Output slots: ONE global atomic per workgroup.  Every wave first counts the non-zero values it is going to
// write (pass 1 over its registers / counters), the wave totals meet in LDS, one lane reserves the whole tile's
// range and every wave gets its start; pass 2 writes.  (One atomic per wave and emitted row — all on the same
// word, which the memory side serves one at a time — was most of the join on inputs with many small tiles:
// C5, 1 M sources: 21.4 -> 1.05 ms; C3: 10.4 -> 3.8 ms.)  `scratch`: JW + 2 words of LDS nobody else touches between the two barriers.
__device__ inline u64 emit_reserve(const JoinArgs& a, u32* scratch, const u32 wave_count, const int tid, const int lane,
                                   const int wv) {
    if (lane == 0) scratch[wv] = wave_count;
    __syncthreads();
    if (tid == 0) {
        u32 t = 0;
        for (int w = 0; w < JW; ++w) t += scratch[w];
        const unsigned long long b = t ? atomicAdd(a.out_count, (unsigned long long)t) : 0ull;
        scratch[JW] = (u32)b;
        scratch[JW + 1] = (u32)(b >> 32);
    }
    __syncthreads();
    u64 base = (u64)scratch[JW] | ((u64)scratch[JW + 1] << 32);
    for (int w = 0; w < wv; ++w) base += scratch[w];
    return base;
}
// one value per lane at the wave's running output position (ballot + popcount prefix inside the wave)
__device__ inline void emit_at(const JoinArgs& a, const u32 gi, const u32 gj, const u32 v, const int lane, u64& wpos) {
    const bool nz = v != 0;
    const unsigned long long mask = __ballot(nz);
    if (mask == 0) return;
    if (nz) {
        const u64 pos = wpos + __popcll(mask & ((1ull << lane) - 1ull));
        if (pos < a.cap) {
            const u32 o1 = a.inv[gi], o2 = a.inv[gj];   // back from the engine's source order to the caller's ids
            ksp_edge e;
            e.source_1 = min(o1, o2);
            e.source_2 = max(o1, o2);
            e.shared = v;
            a.out[pos] = e;
        }
    }
    wpos += (u64)__popcll(mask);
}
__device__ inline u32 count_nz(const u32 v) { return (u32)__popcll(__ballot(v != 0)); }

// Compact the non-zero counters of one tile into (source_1, source_2, shared) records.
template <class Get>
__device__ inline void emit_tile(const JoinArgs& a, const u32 I, const u32 J, const int tid, const int lane, u32* scratch,
                                 Get get) {
    const u32 gi0 = I * TB, gj0 = J * TB;
    const int wv = tid >> 6;
    u32 cnt = 0;
    for (int base = 0; base < TB * TB; base += JW * 64) cnt += count_nz(get(base + tid));
    u64 wpos = emit_reserve(a, scratch, cnt, tid, lane, wv);
    for (int base = 0; base < TB * TB; base += JW * 64) {
        const int idx = base + tid;
        emit_at(a, gi0 + (u32)(idx / TB), gj0 + (u32)(idx % TB), get(idx), lane, wpos);
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
// The partner's word of every butterfly step comes through the vector unit (gfx950: v_permlane32_swap / v_permlane16_swap
// across the halves and the rows of a wave, DPP row rotation / mirrors / quad permutations inside a row), not through
// __shfl_xor — which compiles to ds_bpermute_b32: twelve dependent round trips through the LDS pipe per transpose, in a
// kernel whose workgroups keep that pipe busy with column reads.
template <int J>
__device__ inline u64 transpose64_step(const u64 x, const int lane, const u64 m) {
    const u64 p = (u64)xor_lane<J>((u32)x, lane) | ((u64)xor_lane<J>((u32)(x >> 32), lane) << 32);
    return (lane & J) == 0 ? ((x & m) | ((p & m) << J)) : (((p >> J) & m) | (x & (m << J)));
}
__device__ inline u64 transpose64(u64 x, const int lane) {   // bit b of lane r  <->  bit r of lane b
    {   // step 32: lanes < 32 keep their low word and take the partner's low word as their high word, lanes >= 32 the other
        // way round — the upper half of the low words changes places with the lower half of the high words: ONE instruction
        const auto r = __builtin_amdgcn_permlane32_swap((u32)x, (u32)(x >> 32), false, false);
        x = (u64)r[0] | ((u64)r[1] << 32);
    }
    x = transpose64_step<16>(x, lane, 0x0000FFFF0000FFFFull);
    x = transpose64_step<8>(x, lane, 0x00FF00FF00FF00FFull);
    x = transpose64_step<4>(x, lane, 0x0F0F0F0F0F0F0F0Full);
    x = transpose64_step<2>(x, lane, 0x3333333333333333ull);
    x = transpose64_step<1>(x, lane, 0x5555555555555555ull);
    return x;
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
    WGT_ACC(a, 4);
    // the posting words of a chunk are requested while the previous chunk is being accumulated (they are the first
    // of two dependent loads: posting word -> mask), so a chunk starts with its mask loads
    constexpr u32 GW = (G + JW - 1) / JW, GB = 3;   // groups per wave and chunk, in batches of GB (register budget)
    u32 inf_n[GW];
    auto fetch_inf = [&](const u32 base) {
#pragma unroll
        for (u32 q = 0; q < GW; ++q) {
            const u32 g = (u32)wv + q * JW, k = base + 64u * g + (u32)lane;
            inf_n[q] = (base < klen && g < G && k < klen) ? a.info[kb0 + k] : 0xFFFFFFFFu;
        }
    };
    if (!a.pmask) fetch_inf(0);
    else {
#pragma unroll
        for (u32 q = 0; q < GW; ++q) inf_n[q] = 0xFFFFFFFFu;
    }
    for (u32 base = 0; base < klen; base += G * 64) {
        const u32 ng = min(G, (klen - base + 63u) / 64u);
        // transpose 64 masks into 128 column words; a wave takes groups wv, wv + JW, ...
        u32 inf_c[GW];
#pragma unroll
        for (u32 q = 0; q < GW; ++q) inf_c[q] = inf_n[q];
        for (u32 q0 = 0; q0 < GW; q0 += GB) {
            uint4 mk[GB];
#pragma unroll
            for (u32 q = 0; q < GB; ++q) {
                mk[q] = make_uint4(0, 0, 0, 0);
                if (a.pmask) {   // masks at their list positions: one coalesced load, no posting word needed
                    const u32 g = (u32)wv + (q0 + q) * JW, k = base + 64u * g + (u32)lane;
                    if (q0 + q < GW && g < ng && k < klen) mk[q] = a.pmask[kb0 + k];
                } else {
                    const u32 inf = q0 + q < GW ? inf_c[q0 + q] : 0xFFFFFFFFu;
                    if (inf != 0xFFFFFFFFu && inf >= BIG) mk[q] = a.bigmask[inf & ~BIG];
                }
            }
            if (q0 == 0 && !a.pmask) fetch_inf(base + G * 64);   // (the next chunk's posting words: in flight during this chunk's work)
#pragma unroll
            for (u32 q = 0; q < GB; ++q) {
                const u32 g = (u32)wv + (q0 + q) * JW;
                if (q0 + q < GW && g < ng) {
                    const u32 inf = inf_c[q0 + q];
                    uint4 m = mk[q];
                    if (!a.pmask && inf != 0xFFFFFFFFu && inf < BIG) m = posting_mask(inf, a.bigmask);   // inline ids -> mask
                    const u64 lo = transpose64((u64)m.x | ((u64)m.y << 32), lane);
                    const u64 hi = transpose64((u64)m.z | ((u64)m.w << 32), lane);
                    col[g * 128u + (u32)lane] = lo;
                    col[g * 128u + 64u + (u32)lane] = hi;
                }
            }
        }
        WGT_ACC(a, 5);
        __syncthreads();
        WGT_ACC(a, 6);
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
        WGT_ACC(a, 7);
        __syncthreads();
        WGT_ACC(a, 8);
    }
    if (dst) {   // one of several shares: partial counts into the tile's buffer
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (off && acc[4 * x + y]) atomicAdd(&dst[(4u * ti + (u32)x) * TB + 4u * tj + (u32)y], acc[4 * x + y]);
        if (dia && dacc) atomicAdd(&dst[s0 * TB + s1], dacc);
        WGT_ACC(a, 9);
        return;
    }
    const u32 g0 = I * TB;
    u32 cnt = count_nz(dia ? dacc : 0u);
#pragma unroll
    for (int i = 0; i < 16; ++i) cnt += count_nz(off ? acc[i] : 0u);
    u64 wpos = emit_reserve(a, reinterpret_cast<u32*>(smem), cnt, tid, lane, wv);   // (the column buffer is dead: last barrier above)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) emit_at(a, g0 + 4u * ti + (u32)x, g0 + 4u * tj + (u32)y, off ? acc[4 * x + y] : 0u, lane, wpos);
    emit_at(a, g0 + s0, g0 + s1, dia ? dacc : 0u, lane, wpos);
    WGT_ACC(a, 10);
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
    // the keys of the wave's NEXT step are requested as soon as the step is known (end of the previous one): they
    // arrive while the workgroup collects and accumulates — a round was two memory round trips before its search began
    uint4 An = make_uint4(INF_A, INF_A, INF_A, INF_A), Bn = make_uint4(PAD, PAD, PAD, PAD);
    if (have) { An = load_a(a, ca, a0, a1, lane); Bn = load_b(a, cb, b1, lane); }
    // the two 4 x 4 patches of this thread: rows 4 pi .. (block I), columns 4 pj .. (block J)
    const u32 pi0 = (u32)tid >> 5, pi1 = pi0 + 16u, pj = (u32)tid & 31u;
    constexpr int NA = C16 ? 8 : 16;   // C16: acc[k] = pairs (x, 2k) and (x, 2k + 1) ... see below
    u32 acc0[NA], acc1[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { acc0[i] = 0; acc1[i] = 0; }
    u32 s0 = 0, s1 = 0, s2 = 0;
    WGT_ACC(a, 4);
    while (true) {
        __syncthreads();   // (everyone has read the previous round's s_n / s_more)
        if (tid == 0) { s_n = 0; s_more = 0; }
        __syncthreads();
        if (have) {
            // one step: chunk [ca, ca + 256) of the cell's A keys against window [cb, cb + 256) of its B keys
            const uint4 A = An, B = Bn;
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
            if (have) { An = load_a(a, ca, a0, a1, lane); Bn = load_b(a, cb, b1, lane); }
            if (have && lane == 0) s_more = 1;
        }
        WGT_ACC(a, 5);
        __syncthreads();
        WGT_ACC(a, 6);
        const u32 n = s_n;
        const bool more = s_more != 0;
        for (u32 mb = 0; mb < n; mb += MSUB) {
            // masks of 64 matches per wave -> bit columns
            const u32 mi = mb + 64u * (u32)wv + (u32)lane;
            uint4 ma = make_uint4(0, 0, 0, 0), mbm = ma;
            if (mi < n) {
                const uint2 q = cl.match[mi];
                if (a.pmask) {   // masks at the list positions of the match: one round of memory latency
                    ma = a.pmask[q.x];
                    mbm = a.pmask[q.y];
                } else {
                    const u32 ia = a.info[q.x], ib = a.info[q.y];
                    ma = posting_mask(ia, a.bigmask);
                    mbm = posting_mask(ib, a.bigmask);
                }
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
            WGT_ACC(a, 7);
            __syncthreads();
            WGT_ACC(a, 8);
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
            WGT_ACC(a, 9);
            __syncthreads();
            WGT_ACC(a, 10);
        }
        if (!more) break;
    }
    WGT_ACC(a, 11);
    // results: partial counts into the tile's buffer (one of several shares) or straight to edges
    const u32 gi = I * TB, gj = J * TB;
    u64 wpos = 0;
    if (!dst) {   // (wave-uniform) one reservation for the whole tile
        u32 cnt = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                cnt += count_nz(C16 ? (acc0[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc0[C16 ? 0 : 4 * x + y]);
                cnt += count_nz(C16 ? (acc1[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc1[C16 ? 0 : 4 * x + y]);
            }
        __syncthreads();   // (nobody reads the collect buffers any more: their first words become the scratch)
        wpos = emit_reserve(a, reinterpret_cast<u32*>(smem), cnt, tid, lane, wv);
    }
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
                emit_at(a, gi + r0, gj + cc, v0, lane, wpos);
                emit_at(a, gi + r1, gj + cc, v1, lane, wpos);
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps hipcc from hoisting all 64 id look-ups: registers)
        }
    WGT_ACC(a, 12);
}

// ---- off-diagonal tile from its match records (no search) ---------------------------------------
// Inputs whose keys are shared by few sources each (metagenome bins: ~2 holders per list word) have long
// block lists of which a given block pair matches next to nothing: searching them costs two full lists per
// tile.  Stage 1 knows every key's blocks, so it can hand the join the matches themselves (k_match_emit):
// this share's slice of the tile's records is applied to the counter tile, one record per lane and step.
template <bool C16>
__device__ inline void join_matches_counters(const JoinArgs& a, u32* S, WaveLds& wl, const u32 r0, const u32 r1,
                                             const int tid, const int lane) {
    constexpr u32 UN = 4;   // records per lane and round: four loads in flight instead of one round trip per record
    for (u32 base = r0; base < r1; base += UN * JW * 64) {   // (uniform trip count: pending_apply is wave-wide)
        u64 rec[UN];
#pragma unroll
        for (u32 k = 0; k < UN; ++k) {
            const u32 i = base + k * JW * 64 + (u32)tid;
            rec[k] = i < r1 ? a.mrec[i] : ~0ull;
        }
#pragma unroll
        for (u32 k = 0; k < UN; ++k) {
            if (base + k * JW * 64 >= r1) break;   // (uniform)
            Pending q;
            q.valid = rec[k] != ~0ull;             // (no record is all ones: a posting word never is)
            q.ia = (u32)rec[k];
            q.ib = (u32)(rec[k] >> 32);
            q.w = 1;
            pending_apply<C16>(S, wl.lst, a.bigmask, q, lane);
        }
    }
}
// ... or accumulated bit-sliced, as join_cells_collect does with the matches it finds (multi-source postings)
template <int S_BYTES, bool C16>
__device__ inline void join_matches_collect(const JoinArgs& a, unsigned char* smem, const u32 I, const u32 J, const u32 r0,
                                            const u32 r1, u32* __restrict__ dst, const int tid, const int lane, const int wv) {
    CollectLds& cl = *reinterpret_cast<CollectLds*>(smem);
    const u32 pi0 = (u32)tid >> 5, pi1 = pi0 + 16u, pj = (u32)tid & 31u;
    constexpr int NA = C16 ? 8 : 16;
    u32 acc0[NA], acc1[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) { acc0[i] = 0; acc1[i] = 0; }
    for (u32 mb = r0; mb < r1; mb += MSUB) {
        const u32 mi = mb + 64u * (u32)wv + (u32)lane;
        uint4 ma = make_uint4(0, 0, 0, 0), mbm = ma;
        if (mi < r1) {
            const u64 rec = a.mrec[mi];
            ma = posting_mask((u32)rec, a.bigmask);
            mbm = posting_mask((u32)(rec >> 32), a.bigmask);
        }
        if (mb + 64u * (u32)wv < r1) {   // wave-uniform
            cl.col[0][wv][lane] = transpose64((u64)ma.x | ((u64)ma.y << 32), lane);
            __builtin_amdgcn_sched_barrier(0);
            cl.col[0][wv][64 + lane] = transpose64((u64)ma.z | ((u64)ma.w << 32), lane);
            __builtin_amdgcn_sched_barrier(0);
            cl.col[1][wv][lane] = transpose64((u64)mbm.x | ((u64)mbm.y << 32), lane);
            __builtin_amdgcn_sched_barrier(0);
            cl.col[1][wv][64 + lane] = transpose64((u64)mbm.z | ((u64)mbm.w << 32), lane);
        }
        __syncthreads();
        const u32 ng = min((u32)(MSUB / 64), (r1 - mb + 63u) / 64u);
#pragma unroll 1
        for (u32 g = 0; g < ng; ++g) {
            const ulonglong2* cc = reinterpret_cast<const ulonglong2*>(&cl.col[1][g][4u * pj]);
            const ulonglong2 c01 = cc[0], c23 = cc[1];
            const u64 cw[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const ulonglong2* rr = reinterpret_cast<const ulonglong2*>(&cl.col[0][g][4u * (h ? pi1 : pi0)]);
                const ulonglong2 x01 = rr[0], x23 = rr[1];
                const u64 rw[4] = {x01.x, x01.y, x23.x, x23.y};
                u32* acc = h ? acc1 : acc0;
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; y += 2) {
                        const u32 v0 = (u32)__popcll(rw[x] & cw[y]), v1 = (u32)__popcll(rw[x] & cw[y + 1]);
                        if (C16) acc[2 * x + y / 2] += v0 | (v1 << 16);
                        else { acc[4 * x + y] += v0; acc[4 * x + y + 1] += v1; }
                    }
            }
        }
        __syncthreads();
    }
    const u32 gi = I * TB, gj = J * TB;
    u64 wpos = 0;
    if (!dst) {
        u32 cnt = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                cnt += count_nz(C16 ? (acc0[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc0[C16 ? 0 : 4 * x + y]);
                cnt += count_nz(C16 ? (acc1[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc1[C16 ? 0 : 4 * x + y]);
            }
        __syncthreads();
        wpos = emit_reserve(a, reinterpret_cast<u32*>(smem), cnt, tid, lane, wv);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const u32 r0c = 4u * pi0 + (u32)x, r1c = 4u * pi1 + (u32)x, cc = 4u * pj + (u32)y;
            const u32 v0 = C16 ? (acc0[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc0[C16 ? 0 : 4 * x + y];
            const u32 v1 = C16 ? (acc1[2 * x + y / 2] >> (16 * (y & 1))) & 0xFFFFu : acc1[C16 ? 0 : 4 * x + y];
            if (dst) {
                if (v0) atomicAdd(&dst[r0c * TB + cc], v0);
                if (v1) atomicAdd(&dst[r1c * TB + cc], v1);
            } else {
                emit_at(a, gi + r0c, gj + cc, v0, lane, wpos);
                emit_at(a, gi + r1c, gj + cc, v1, lane, wpos);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <bool W, bool C16, bool CELLS>
__device__ __forceinline__ void join_workgroup(const JoinArgs& a, u32& I, u32& J, u32& sub, u32& sp) {
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
    u32 tail_id = 0xFFFFFFFFu;
    sub = 0; sp = 1;
    u32 mr0 = 0, mr1 = 0;   // match-list mode: this share's records
    if (a.sched) {
        const u32 wg = a.wg0 + blockIdx.x;
        const u32 ai = a.sched[wg];
        const u32* t = a.act + 4 * (size_t)ai;
        if (a.mrec) { mr0 = a.mstart[ai]; mr1 = a.mstart[ai + 1]; }
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
    const bool matches = !W && a.mrec != nullptr && I != J;
    if (matches) {   // this share's slice of the tile's records
        const u32 n = mr1 - mr0;
        mr1 = mr0 + (u32)(((u64)n * (sub + 1)) / sp);
        mr0 = mr0 + (u32)(((u64)n * sub) / sp);
    }
    // 16-bit counters are exact when no pair can reach 2^16 in THIS SHARE: one of the two blocks has no source with
    // >= 2^16 k-mers (shared <= min(n_a, n_b)) — or, unweighted (every key counts 1), the share itself offers fewer keys:
    // a share of an off-diagonal tile of the match-list join holds one record per key both blocks have, a share of a
    // diagonal tile a range of the block's keys, any other tile cannot count more keys than the shorter of its two lists
    // holds.  The shares of a tile add their counters into the tile's 32-bit buffer, so only a share has to stay below
    // 2^16, not the tile (the host cuts the shares of big blocks accordingly).  (Metagenome bins: a block of 128 bins
    // nearly always holds one of >= 2^16 hashes, so half the tiles took the 32-bit instantiation — 64 KB of LDS, two
    // workgroups per CU, 9.2 of 13.5 ms — although a pair of unrelated bins shares a few hundred keys.)
    {
        u32 lim = min(a.blk_max[I], a.blk_max[J]);
        if (!W) {
            const u32 kI = a.blk_raw[I + 1] - a.blk_raw[I], kJ = a.blk_raw[J + 1] - a.blk_raw[J];
            const u32 mine = matches ? mr1 - mr0
                           : I == J  ? (u32)(((u64)kI * (sub + 1)) / sp) - (u32)(((u64)kI * sub) / sp)
                                     : min(kI, kJ);
            lim = min(lim, mine);
        }
        if ((lim < 65536u) != C16) return;
    }

    const bool popc = !W && a.collect && (I == J || CELLS);
    if (popc) {
        // unweighted tiles: bit-sliced accumulation in registers (no counter tile, no LDS atomics)
        u32* dst = tail_id != 0xFFFFFFFFu ? a.tailbuf + (size_t)tail_id * (TB * TB) : nullptr;
        if (I == J) self_tile_popc<SMEM_BYTES>(a, smem, I, sub, sp, dst, tid, lane, wv);
        else if (matches) join_matches_collect<S_BYTES, C16>(a, smem, I, J, mr0, mr1, dst, tid, lane, wv);
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
    } else if (matches) {
        join_matches_counters<C16>(a, S, wlds[wv], mr0, mr1, tid, lane);
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
        WGT_ACC(a, 13);
        if (!s_last) return;
        __threadfence();
        emit_tile(a, I, J, tid, lane, reinterpret_cast<u32*>(wlds),
                  [&](int idx) { return __hip_atomic_load(&dst[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
        WGT_ACC(a, 14);
        return;
    }
    // flush: compact the non-zero counters of the tile into edges (the per-wave LDS is free by now: scratch)
    emit_tile(a, I, J, tid, lane, reinterpret_cast<u32*>(wlds),
              [&](int idx) { return C16 ? ((S[idx >> 1] >> ((idx & 1) * 16)) & 0xFFFFu) : S[idx]; });
}

template <bool W, bool C16, bool CELLS>
__global__ __launch_bounds__(JW * 64, 6) void k_join(JoinArgs a) {   // (6 waves per SIMD = three workgroups per CU: caps the unweighted variant at 80 VGPRs)
    u32 I = 0, J = 0, sub = 0, sp = 1;
#ifdef KSP_WGTIME
    const unsigned long long t0 = wall_clock64();
    if (a.wgt && threadIdx.x == 0) a.wgt[16 * (size_t)(a.wg0 + blockIdx.x) + 15] = t0;
#endif
    join_workgroup<W, C16, CELLS>(a, I, J, sub, sp);
#ifdef KSP_WGTIME
    if (a.wgt && threadIdx.x == 0) {
        unsigned long long* r = a.wgt + 16 * (size_t)(a.wg0 + blockIdx.x);
        r[0] = t0; r[1] = wall_clock64(); r[2] = ((unsigned long long)I << 32) | J; r[3] = ((unsigned long long)sub << 32) | sp;
    }
#endif
}

// one workgroup per tail tile: its summed counters -> edges
__global__ __launch_bounds__(JW * 64) void k_tail_emit(JoinArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    u32 I, J;
    tile_decode(a.tile_begin + a.n_normal + blockIdx.x, a.nb, I, J);
    const u32* src = a.tailbuf + (size_t)blockIdx.x * (TB * TB);
    __shared__ u32 scratch[JW + 2];
    emit_tile(a, I, J, tid, lane, scratch, [&](int idx) { return src[idx]; });
}

