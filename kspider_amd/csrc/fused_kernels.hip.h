// Stage 1, middle part, bucket-resident (round 3).  Included by engine.hip inside namespace ksp, after
// stage1_kernels.hip.h (constants, tag helpers, BucketBounds, hb_slot).
//
// What it replaces: k_bucket_group -> scan -> k_bucket_emit -> k_label -> [order] -> k_key_groups -> scan ->
// k_move_groups -> k_ms_hist: five passes that re-read what the pass before wrote, because a bucket's place in the
// dense output (tags in key order, first[], group records in rank order) was only known after a scan over all buckets.
// Here a bucket never leaves its own region (what DESIGN.md section 9.2 of round 2 asked for):
//
//   k_fgroup   one bucket at a time in LDS (the hash grouping of k_bucket_group), and while it is there: the kept
//              tags leave in key order into the bucket's OWN places of a second tag array, the start of every kept key
//              into the bucket's own places of kst[], the bucket's totals into bsum[] — and the label pass runs on the
//              bucket in LDS (min source id over the holders of a sampled key, one atomicMin per holder).  No record per
//              entry, no emit pass, no first[] / crank[], no label kernel.
//   (scan of bsum: ranks and record places of every bucket; source order from the labels — unchanged)
//   k_fkeys    one workgroup per chunk of consecutive buckets: stages a bucket's kept holders as new source indices,
//              walks every key (the (block, key) groups of k_key_groups) and writes the group records STRAIGHT to their
//              final order: a key's rank is chunk base + the order in which it was committed, its records take the next
//              places of the chunk's region (one packed LDS atomic gives both, so record order = rank order), the region
//              starts at the chunk's first kept entry (groups <= kept entries: regions never overlap).  Also counted
//              here: the chunk's records per block (the histogram row k_ms_hist made) and the diagonal work.
//   k_fms_place  k_ms_place over a chunk's region (any number of records: rounds of 2 048), and the tile flags of
//              k_tile_flags read off the same records.
//
// src/pairwise.cpp:194-225 equivalent: still "bring the holders of every key together", nothing else.
#pragma once

constexpr u32 FK_THREADS = 512;
constexpr u32 FK_CAP = HB_CAP;          // kept holders of one bucket staged at a time
constexpr u32 FK_KEYS = HB_CAP / 2;     // kept keys of one bucket
constexpr u32 FK_NB_MAX = 1024;         // blocks the LDS tables of k_fkeys hold (= MS_MAXB: the split's tables)

// places of bucket b's key starts in kst[] (a kept key has >= 2 entries: at most size / 2 keys, + 1 sentinel)
__device__ __host__ inline u32 fk_kst_first(const u32 first_entry, const u32 b) { return first_entry / 2 + b; }

// ---- grouping + emit + labels, one bucket at a time --------------------------------------------------------------
// Persistent workgroups, buckets b = blockIdx.x, += gridDim.x (as k_bucket_group: the next bucket's keys and tags are
// in flight while this one is grouped).  out_tags / kst / bsum as described above; label == nullptr: no label pass.
template <class V>
__global__ __launch_bounds__(HB_THREADS, 6) void k_fgroup(const u64* __restrict__ keys, const V* __restrict__ tags, const BucketBounds bb,
                                                          const u32 nbuckets, V* __restrict__ out_tags, u32* __restrict__ kst,
                                                          u64* __restrict__ bsum, u32* __restrict__ overflow, u32* __restrict__ big_list,
                                                          u32* __restrict__ label, const int lshift, const u32 skip, const u32 max_holders) {
    constexpr u32 NT = HB_THREADS, NWV = NT / 64;
    __shared__ unsigned long long tkey[HB_SLOTS + 1];
    __shared__ u32 tcnt2[HB_SLOTS / 2 + 1];   // entries per key, two 16-bit counters per word
    __shared__ unsigned short eslot[HB_CAP];
    __shared__ u32 wpart[NWV];
    // after the inserts the keys are dead; their storage then holds per slot the first place of the key's entries |
    // entries placed so far << 16, the bucket's kept tags in key order, and the start of every kept key
    u32* const tplace = (u32*)tkey;                                           // HB_SLOTS + 1 words
    V* const o_tag = (V*)(tplace + (HB_SLOTS + 2));                           // HB_CAP tags
    unsigned short* const kst_l = (unsigned short*)(o_tag + HB_CAP);          // FK_KEYS + 2
    static_assert((HB_SLOTS + 2) * 4 + HB_CAP * sizeof(V) + (FK_KEYS + 2) * 2 <= (HB_SLOTS + 1) * 8, "aliases fit the key table");
    constexpr unsigned long long EMPTY = ~0ull;
    constexpr u32 EPT = HB_CAP / NT;
    constexpr u32 PER = (HB_SLOTS + 1 + NT - 1) / NT;
    const u32 tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // (the bounds through restrict pointers: nothing this kernel writes overlaps them, so the next bucket's bounds are scalar
    //  loads — as vector loads each was waited for on the spot, three round trips at the head of every pass)
    const u32* __restrict__ const bstart = bb.start;
    const u32* __restrict__ const bcnt = bb.cnt;
    const u32 bcap = bb.cap;
    auto bfirst = [&](const u32 q) { return bcap ? q * bcap : bstart[q]; };
    auto bsize = [&](const u32 q) { return bcap ? min(bcnt[q], bcap) : bstart[q + 1] - bstart[q]; };
    // bounds two buckets ahead: the loads of bucket b + 2 x grid are issued at the head of pass b and first looked at when
    // pass b ends (next bucket's keys are fetched from bounds that arrived a whole pass earlier)
    u32 vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    u32 b = blockIdx.x;
    u32 b0 = 0, raw = 0, n0 = 0, nraw = 0;
    if (b < nbuckets) { b0 = bfirst(b); raw = bsize(b); }
    if (b + gridDim.x < nbuckets) { n0 = bfirst(b + gridDim.x); nraw = bsize(b + gridDim.x); }
    u32 size = raw > HB_CAP ? 0 : raw;   // a bucket that does not fit is listed (overflow[1] counts them): the caller falls back
    unsigned long long mykey[EPT], nkey[EPT];
    V mytag[EPT], ntag[EPT];
#pragma unroll
    for (u32 j = 0; j < EPT; ++j) {
        const bool in = tid + j * NT < size;
        mykey[j] = in ? keys[b0 + tid + j * NT] : 0;
        mytag[j] = in ? tags[b0 + tid + j * NT] : V(0);
    }
    // (the first bucket's loads have landed before the loop is entered: otherwise the compiler's wait-count bookkeeping carries
    //  "a load may still be writing these registers" into EVERY pass of the loop and drains the memory pipeline — s_waitcnt
    //  vmcnt(0), the prefetch of the next bucket and every store included — wherever one of them is touched)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    while (b < nbuckets) {
        if (raw > HB_CAP && tid == 0) {
            const u32 q = atomicAdd(&overflow[1], 1u);
            big_list[q] = b;   // (one slot per bucket: cannot overflow)
        }
        const u32 bn = b + gridDim.x, bnn = bn + gridDim.x;
        // bounds of the bucket after the next, loaded through a lane-private zero: the compiler cannot tell that the address is
        // uniform, so the values stay in vector registers (no readfirstlane, no wait) until the pass ends
        u32 la = 0, lb = 0;   // (the words as loaded; first / size are worked out when the pass ends)
        if (bnn < nbuckets) {
            if (bcap) la = bcnt[bnn + vzero];
            else { la = bstart[bnn + vzero]; lb = bstart[bnn + 1 + vzero]; }
        }
        const u32 slots = size <= 416 ? 512u : size <= 832 ? 1024u : size <= 1664 ? 2048u : HB_SLOTS;
        for (u32 i = tid; i <= slots; i += NT) tkey[i] = EMPTY;
        for (u32 i = tid; i <= slots / 2; i += NT) tcnt2[i] = 0;
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            const u32 i = tid + j * NT;
            if (i >= size) break;
            const unsigned long long key = mykey[j];
            u32 h;
            if (key == EMPTY) {
                h = slots;   // the one key that looks like an empty slot has a slot of its own
            } else {
                h = hb_slot(key) & (slots - 1);
                while (true) {
                    const unsigned long long prev = atomicCAS(&tkey[h], EMPTY, key);
                    if (prev == EMPTY || prev == key) break;
                    h = (h + 1) & (slots - 1);
                }
            }
            eslot[i] = (unsigned short)h;
            atomicAdd(&tcnt2[h >> 1], 1u << (16 * (h & 1)));
        }
        // the next bucket's keys and tags: in flight during the scan, the placement and the output
        const u32 nsize = nraw > HB_CAP ? 0 : nraw;
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            const bool in = tid + j * NT < nsize;
            nkey[j] = in ? keys[n0 + tid + j * NT] : 0;
            ntag[j] = in ? tags[n0 + tid + j * NT] : V(0);
        }
        __syncthreads();   // (all inserts done: the key table is dead, its storage is reused below)
        // exclusive scan over the slots of (kept entries | kept keys << 16): up to 9 slots per thread
        const u32 per = slots / NT + 1;
        u32 cnt[PER];
        u32 mine = 0;
#pragma unroll
        for (u32 j = 0; j < PER; ++j) {
            const u32 sl = tid * per + j;
            cnt[j] = j < per && sl <= slots ? (tcnt2[sl >> 1] >> (16 * (sl & 1))) & 0xFFFFu : 0;
            if (cnt[j] >= 2) mine += cnt[j] | (1u << 16);
        }
        u32 inc = mine;
        inc = wave_scan_add(inc);
        if (lane == 63) wpart[wv] = inc;
        __syncthreads();
        u32 run = inc - mine;
        for (int w = 0; w < wv; ++w) run += wpart[w];
        u32 tot = 0;
        for (u32 w = 0; w < NWV; ++w) tot += wpart[w];
        const u32 kept = tot & 0xFFFFu, nkeys = tot >> 16;
        if (tid == 0) {
            bsum[b] = (u64)kept | ((u64)nkeys << 32);
            kst_l[nkeys] = (unsigned short)kept;   // sentinel
        }
#pragma unroll
        for (u32 j = 0; j < PER; ++j) {
            const u32 sl = tid * per + j;
            if (cnt[j] >= 2) {
                tplace[sl] = run & 0xFFFFu;                          // first place of the key's entries, nothing placed yet
                kst_l[run >> 16] = (unsigned short)(run & 0xFFFFu);  // the key's rank inside the bucket = kept keys in the slots before it
                run += cnt[j] | (1u << 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            const u32 i = tid + j * NT;
            if (i >= size) break;
            const u32 sl = eslot[i];
            if (((tcnt2[sl >> 1] >> (16 * (sl & 1))) & 0xFFFFu) >= 2) {
                const u32 old = atomicAdd(&tplace[sl], 1u << 16);
                o_tag[(old & 0xFFFFu) + (old >> 16)] = mytag[j];
            }
        }
        __syncthreads();
        // the bucket's kept tags, in key order, to the bucket's own places; key starts as positions in that array
        for (u32 i = tid; i < kept; i += NT) out_tags[b0 + i] = o_tag[i];
        const u32 kb0 = fk_kst_first(b0, b);
        for (u32 r = tid; r <= nkeys; r += NT) kst[kb0 + r] = b0 + kst_l[r];
        if (label) {
            // label pass on the bucket in LDS: one key in (skip + 1), one thread per sampled key (see k_label).  No look at
            // the label first: a load would put a memory round trip per holder on the bucket's critical path — the atomics
            // are sent without waiting for anything (nothing is returned) and land while the next bucket is grouped.
            for (u32 r = tid * (skip + 1); r < nkeys; r += NT * (skip + 1)) {
                const u32 f0 = kst_l[r], f1 = kst_l[r + 1];
                if (f1 - f0 > max_holders) continue;
                u32 mn = ~0u;
                for (u32 x = f0; x < f1; ++x) mn = min(mn, src_of_tag(tag_of(o_tag[x])));
                for (u32 x = f0; x < f1; ++x) {
                    const u32 s = src_of_tag(tag_of(o_tag[x]));
                    if (s != mn) __hip_atomic_fetch_min(&label[(size_t)s << lshift], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        b = bn; b0 = n0; raw = nraw; size = nsize;
        {
            const u32 ua = __builtin_amdgcn_readfirstlane(la), ub = __builtin_amdgcn_readfirstlane(lb);
            n0 = bnn < nbuckets ? (bcap ? bnn * bcap : ua) : 0u;
            nraw = bnn < nbuckets ? (bcap ? min(ua, bcap) : ub - ua) : 0u;
        }
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) { mykey[j] = nkey[j]; mytag[j] = ntag[j]; }
        __syncthreads();   // the table is rebuilt from here on
    }
}

// totals of the bucket scan (bbase = exclusive scan of bsum): scal[6] = kept entries, scal[2] = kept distinct keys
__global__ void k_ftotals(const u64* __restrict__ bsum, const u64* __restrict__ bbase, const u32 nbuckets, u64* __restrict__ scal) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const u64 t = bbase[nbuckets - 1] + bsum[nbuckets - 1];
        scal[6] = (u32)t;
        scal[2] = (u32)(t >> 32);
    }
}

// ---- (block, key) groups of a chunk of buckets, records straight to rank order -----------------------------------
// rec_blk / rec_val: region of chunk c starts at place (u32)bbase[first bucket of c]; nrec[c] records are used.
// hist: records of chunk c per block, row c (NBT words).  work: diagonal work per block, holders in slot nb.
// A bucket holds at most HB_CAP entries here (the caller only runs this path when no bucket was oversize).
//
// One bucket at a time per workgroup, its successor's tags (-> new source indices) and key starts already in flight:
//   A  the holders' new indices and the key starts go to LDS; every key marks its first place in a bitmap;
//   B  prefix popcounts of the bitmap: entry i belongs to key (heads at or below i) - 1;
//   C  ENTRY-parallel: every holder ORs its bit into its key's 128-bit mask in LDS and compares its block with its left
//      neighbour's (same key, different block -> the key is marked "several blocks").  No walk, no divergence: a wave
//      handles 64 holders whatever the sizes of their keys (a thread per key walked ~25 steps for the longest of its
//      wave's 64 keys and ~3 for the average one);
//   D  KEY-parallel: a key whose holders share one block — nearly all after the source reordering — is one record: block
//      of its first holder, the mask as it stands.  Keys in several blocks: up to FK_SMALL holders in their thread's
//      registers, above that a whole wave walks the key block by block as k_key_groups does.
// A key's rank is chunk base + the order in which it is committed; its records take the next places of the chunk's
// region — one packed LDS atomic hands out both, so record order = rank order.
struct FkOut {
    u32* rec_blk;
    u64* rec_val;
    u32* nrec;
    u32* hist;      // chunks x NBT
    uint4* bigmask;
    unsigned long long* work;
};
// the grouping gave up, the partition overflowed or a bucket was oversize: the host repeats the build, and the kernels
// queued behind the grouping must not walk tables nobody wrote (known before they start: one decision per launch)
__device__ inline bool fk_abandoned(const u64* __restrict__ scal) {
    return ((u32)scal[9] | (u32)scal[PC_OVF] | reinterpret_cast<const u32*>(scal + 9)[1]) != 0;
}
// (timing build, make fktime: thread 0 of every workgroup adds the shader-clock time of every stage to fk_time[stage])
#ifdef KSP_FKTIME
__device__ unsigned long long fk_time[16];
#define FK_T(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&fk_time[k], now_ - t_last_); t_last_ = now_; } } while (0)
#else
#define FK_T(k) do { } while (0)
#endif
constexpr u32 FK_SMALL = 8;     // holders of a several-block key its thread takes in registers
constexpr u32 FK_KEYB = 1024;   // keys whose masks are in LDS at a time (a batch with more takes another round)
constexpr u32 FK_BE = 8192;     // kept holders of a batch of consecutive buckets (staged in LDS together)
constexpr u32 FK_BK = 4096;     // ... and its kept keys
constexpr u32 FK_MAXBQ = 4;     // buckets of a batch at most (their loads are unrolled)
constexpr u32 FK_GBMAX = 32;    // buckets per chunk at most (LDS table of their totals)
template <class V, u32 NBT>
__global__ __launch_bounds__(FK_THREADS) void k_fkeys(const V* __restrict__ tags, const u32* __restrict__ kst, const BucketBounds bb,
                                                      const u64* __restrict__ bsum, const u64* __restrict__ bbase, const u32 nbuckets,
                                                      const u32 gb, const u32* __restrict__ newidx, const u32 n_sources, const u32 nb,
                                                      const FkOut out, const u64* __restrict__ scal) {
    constexpr u32 NT = FK_THREADS, NWV = NT / 64, HW = FK_BE / 32;
    typedef typename std::conditional<(NBT * TB <= 65536u && sizeof(V) <= 2), unsigned short, u32>::type IT;   // a tag, then a new source index
    __shared__ IT s_idx[FK_BE + 4];                 // the batch's holders: tags first, then (in place) new source indices
    __shared__ unsigned short s_kst[FK_BK + 2];     // first place of every key of the batch, + sentinel
    __shared__ u32 s_head[HW + 1], s_hpre[HW + 1];
    __shared__ u32 s_mask[FK_KEYB * 4];
    __shared__ u32 s_multi[FK_KEYB / 32];
    __shared__ unsigned short s_big[FK_KEYB];
    __shared__ u32 s_nbig;
    __shared__ unsigned long long s_cur;                    // records committed so far | keys committed so far << 32
    __shared__ u32 s_hist[NBT];
    __shared__ unsigned long long s_work[NBT + 1];
    __shared__ u32 s_bkept[FK_GBMAX], s_bkeys[FK_GBMAX];
    static_assert(sizeof(V) <= sizeof(IT), "a tag fits the staging word");
    if (fk_abandoned(scal)) return;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 bk0 = blockIdx.x * gb, bk1 = min(nbuckets, bk0 + gb);
    const u64 base = bbase[bk0];
    const u32 rbase = (u32)base, kbase = (u32)(base >> 32);
    if (tid == 0) { s_cur = 0; s_nbig = 0; }
    if (tid < bk1 - bk0) {
        const u64 t = bsum[bk0 + tid];
        s_bkept[tid] = min((u32)t, FK_CAP);
        s_bkeys[tid] = min((u32)(t >> 32), FK_KEYS);
    }
    for (u32 i = tid; i < NBT; i += NT) s_hist[i] = 0;
    for (u32 i = tid; i <= nb; i += NT) s_work[i] = 0;
    for (u32 i = tid; i <= HW; i += NT) s_head[i] = 0;
    unsigned long long holders = 0;
    // posting word of one group (block `cur`, members lo | hi); a mask of its own for more than INLINE_MAX members
    auto posting = [&](const u32 fa, const u32 cur, const unsigned long long lo, const unsigned long long hi, u32& bigs, const bool store) -> u32 {
        const u32 cnt = __popcll(lo) + __popcll(hi);
        u32 inf;
        if (cnt <= INLINE_MAX) {
            inf = (cnt - 1) << 29;
            unsigned long long a = lo, bq = hi;
            for (u32 j = 0; j < cnt; ++j) {   // local ids, ascending, 7 bits each
                u32 id;
                if (a) { id = __ffsll((long long)a) - 1; a &= a - 1; }
                else { id = 64 + __ffsll((long long)bq) - 1; bq &= bq - 1; }
                inf |= id << (7 * j);
            }
        } else {
            // a group with a mask takes at least INLINE_MAX + 1 of the key's entries: the masks of the key whose entries begin
            // at kept entry fa have the places fa / 5, fa / 5 + 1, ... to themselves
            const u32 slot = fa / (INLINE_MAX + 1) + bigs;
            if (store) out.bigmask[slot] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
            inf = BIG | slot;
            ++bigs;
        }
        if (store) {
            atomicAdd(&s_hist[cur & (NBT - 1)], 1u);
            holders += cnt;
            if (cnt > 1) atomicAdd(&s_work[cur], (unsigned long long)cnt * (cnt - 1) / 2);
        }
        return inf;
    };
    __syncthreads();
#ifdef KSP_FKTIME
    unsigned long long t_last_ = clock64();
#endif
    u32 eseen = 0;   // kept entries of this chunk's buckets before the batch
    for (u32 bq0 = bk0; bq0 < bk1;) {
        // the batch: up to FK_MAXBQ consecutive buckets while their holders and keys fit the staging (one bucket always does)
        u32 bq1 = bq0, ne = 0, nk = 0;
        while (bq1 < bk1 && bq1 - bq0 < FK_MAXBQ && (bq1 == bq0 || (ne + s_bkept[bq1 - bk0] <= FK_BE && nk + s_bkeys[bq1 - bk0] <= FK_BK))) {
            ne += s_bkept[bq1 - bk0];
            nk += s_bkeys[bq1 - bk0];
            ++bq1;
        }
        // ---- A: tags and key starts of every bucket of the batch to LDS.  Straight-line code, every load unconditional at a
        // clamped (always valid) address: ALL loads of a thread are in flight together — one memory round trip per batch, then
        // one more (L2: the table is small) for the new source indices.  (Loads inside branches or loops were waited for one
        // by one: s_waitcnt vmcnt(0) behind each, 24 round trips per batch, 40 % of the kernel.)
        {
            constexpr u32 EPB = FK_CAP / NT, KPB = (FK_KEYS + 1 + NT - 1) / NT;
            const u32 nbq = bq1 - bq0;
            V tg[FK_MAXBQ][EPB];
            u32 ks[FK_MAXBQ][KPB], ni[FK_MAXBQ][EPB];
            u32 kp[FK_MAXBQ], kk[FK_MAXBQ], fb[FK_MAXBQ];
#pragma unroll
            for (u32 q = 0; q < FK_MAXBQ; ++q) {
                const u32 bq = bq0 + (q < nbq ? q : 0u);
                kp[q] = q < nbq ? s_bkept[bq - bk0] : 0u;
                kk[q] = q < nbq ? s_bkeys[bq - bk0] : 0u;
                fb[q] = bb.first(bq);
            }
#pragma unroll
            for (u32 q = 0; q < FK_MAXBQ; ++q) {
                const u32 kb = fk_kst_first(fb[q], bq0 + (q < nbq ? q : 0u));
#pragma unroll
                for (u32 j = 0; j < EPB; ++j) tg[q][j] = tags[fb[q] + min(tid + j * NT, max(kp[q], 1u) - 1u)];
#pragma unroll
                for (u32 j = 0; j < KPB; ++j) ks[q][j] = kst[kb + min(tid + j * NT, kk[q])];   // (kk + 1 words: the last is the sentinel)
            }
#pragma unroll
            for (u32 q = 0; q < FK_MAXBQ; ++q)
#pragma unroll
                for (u32 j = 0; j < EPB; ++j) ni[q][j] = newidx[min(src_of_tag(tag_of(tg[q][j])), n_sources - 1u)];   // (clamped: an unused place holds anything)
            u32 eo = 0, ko = 0;
#pragma unroll
            for (u32 q = 0; q < FK_MAXBQ; ++q) {
#pragma unroll
                for (u32 j = 0; j < KPB; ++j) {
                    const u32 r = tid + j * NT;
                    if (r < kk[q]) {
                        const u32 off = eo + (ks[q][j] - fb[q]);
                        s_kst[ko + r] = (unsigned short)off;
                        atomicOr(&s_head[off >> 5], 1u << (off & 31));
                    }
                }
#pragma unroll
                for (u32 j = 0; j < EPB; ++j) {
                    const u32 i = tid + j * NT;
                    if (i < kp[q]) s_idx[eo + i] = (IT)ni[q][j];
                }
                eo += kp[q];
                ko += kk[q];
            }
            if (tid == 0) s_kst[nk] = (unsigned short)ne;
        }
        __syncthreads();
        FK_T(2);
        // ---- B: heads in front of every word of the bitmap (256 words: four per lane of one wave)
        if (wv == 0) {
            static_assert(HW == 256, "four bitmap words per lane");
            u32 c[4], t = 0;
#pragma unroll
            for (u32 q = 0; q < 4; ++q) { c[q] = (u32)__popc(s_head[4 * lane + q]); t += c[q]; }
            u32 inc = t;
            inc = wave_scan_add(inc);
            u32 run = inc - t;
#pragma unroll
            for (u32 q = 0; q < 4; ++q) { s_hpre[4 * lane + q] = run; run += c[q]; }
        }
        const u32 ebatch = rbase + eseen;   // kept-entry number of the batch's first holder
        FK_T(3);
        for (u32 kb = 0; kb < max(nk, 1u); kb += FK_KEYB) {   // (one round for nearly every batch)
            const u32 kn = min(FK_KEYB, nk - kb);
            if (kb) __syncthreads();   // (the previous round's masks have been read)
            for (u32 i = tid; i < kn * 4; i += NT) s_mask[i] = 0;
            for (u32 i = tid; i < (kn + 31) / 32; i += NT) s_multi[i] = 0;
            __syncthreads();
            FK_T(4);
            // ---- C: every holder ORs its bit into its key's mask; a holder in another block than its left neighbour marks the key
            for (u32 i0 = tid; i0 < ne; i0 += 4 * NT) {   // (four holders per pass: their LDS reads are independent)
                u32 w[4], hp[4], t[4], tl[4];
#pragma unroll
                for (u32 u = 0; u < 4; ++u) {
                    const u32 i = i0 + u * NT, ic = min(i, ne - 1);
                    w[u] = s_head[ic >> 5]; hp[u] = s_hpre[ic >> 5]; t[u] = s_idx[ic]; tl[u] = ic ? (u32)s_idx[ic - 1] : 0u;
                }
#pragma unroll
                for (u32 u = 0; u < 4; ++u) {
                    const u32 i = i0 + u * NT;
                    if (i >= ne) break;
                    const u32 k = hp[u] + (u32)__popc(w[u] & (0xFFFFFFFFu >> (31 - (i & 31)))) - 1 - kb;   // (heads at or below place i) - 1
                    if (k < kn) {
                        if (!((w[u] >> (i & 31)) & 1u) && tl[u] / TB != t[u] / TB) atomicOr(&s_multi[k >> 5], 1u << (k & 31));
                        atomicOr(&s_mask[k * 4 + ((t[u] % TB) >> 5)], 1u << (t[u] & 31));
                    }
                }
            }
            __syncthreads();
            FK_T(5);
            // ---- D: a key per thread
            for (u32 k0 = 0; k0 < kn; k0 += NT) {   // (uniform trip count: the waves reserve together)
                const u32 kk = k0 + tid;
                const bool have = kk < kn;
                const u32 r = kb + kk;
                const u32 f0 = have ? s_kst[r] : 0u, c = have ? s_kst[r + 1] - f0 : 0u;
                const u32 fa = ebatch + f0;
                const bool multi = have && ((s_multi[kk >> 5] >> (kk & 31)) & 1u);
                // keys with all holders in one block — one record each: the wave reserves their places and ranks with ONE atomic
                const bool single = have && !multi;
                const unsigned long long sm = __ballot(single);
                unsigned long long old = 0;
                if (sm) {
                    const u32 ns = (u32)__popcll(sm);
                    if (lane == (u32)__ffsll((long long)sm) - 1) old = atomicAdd(&s_cur, (unsigned long long)ns | ((unsigned long long)ns << 32));
                    old = __shfl(old, __ffsll((long long)sm) - 1);
                }
                if (single) {
                    const u32 mine = (u32)__popcll(sm & ((1ull << lane) - 1ull));
                    const u32 bfirst = (u32)s_idx[f0] / TB;
                    const unsigned long long lo = (unsigned long long)s_mask[kk * 4] | ((unsigned long long)s_mask[kk * 4 + 1] << 32);
                    const unsigned long long hi = (unsigned long long)s_mask[kk * 4 + 2] | ((unsigned long long)s_mask[kk * 4 + 3] << 32);
                    const u32 pos = rbase + (u32)old + mine, rank = kbase + (u32)(old >> 32) + mine;
                    u32 bigs = 0;
                    out.rec_blk[pos] = bfirst;
                    out.rec_val[pos] = ((u64)rank << 32) | posting(fa, bfirst, lo, hi, bigs, true);
                }
                if (!multi) continue;
                if (c > FK_SMALL) { s_big[atomicAdd(&s_nbig, 1u)] = (unsigned short)r; continue; }   // several blocks, more than a handful of holders: a whole wave (below)
                // several blocks, a handful of holders (the common case of unrelated sources): all of them in registers, one
                // record per distinct block, in order of first appearance
                u32 t[FK_SMALL];
#pragma unroll
                for (u32 q = 0; q < FK_SMALL; ++q) t[q] = q < c ? (u32)s_idx[f0 + q] : ~0u;
                u32 groups = 0;
#pragma unroll
                for (u32 q = 0; q < FK_SMALL; ++q) {
                    bool fresh = q < c;
#pragma unroll
                    for (u32 p2 = 0; p2 < q; ++p2) fresh = fresh && (t[p2] / TB != t[q] / TB);
                    groups += fresh ? 1u : 0u;
                }
                const unsigned long long o2 = atomicAdd(&s_cur, (unsigned long long)groups | (1ull << 32));
                const u32 pos = rbase + (u32)o2, rank = kbase + (u32)(o2 >> 32);
                u32 g = 0, bigs = 0;
#pragma unroll
                for (u32 q = 0; q < FK_SMALL; ++q) {
                    bool fresh = q < c;
#pragma unroll
                    for (u32 p2 = 0; p2 < q; ++p2) fresh = fresh && (t[p2] / TB != t[q] / TB);
                    if (!fresh) continue;
                    const u32 cur = t[q] / TB;
                    unsigned long long lo = 0, hi = 0;
#pragma unroll
                    for (u32 p2 = q; p2 < FK_SMALL; ++p2) {
                        if (p2 < c && t[p2] / TB == cur) {
                            const u32 l = t[p2] % TB;
                            if (l < 64) lo |= 1ull << l; else hi |= 1ull << (l - 64);
                        }
                    }
                    out.rec_blk[pos + g] = cur;
                    out.rec_val[pos + g] = ((u64)rank << 32) | posting(fa, cur, lo, hi, bigs, true);
                    ++g;
                }
            }
            // keys in several blocks with more holders, one wave each: the lanes share the walk (the first holder's block first,
            // then the others ascending, as k_key_groups), masks and next block are reduced
            __syncthreads();
            FK_T(6);
            const u32 nbig = s_nbig;
            for (u32 q = wv; q < nbig; q += NWV) {
                const u32 r = s_big[q];
                const u32 f0 = s_kst[r], c = s_kst[r + 1] - f0;
                const u32 bfirst = (u32)s_idx[f0] / TB;
                u32 groups = 0;
                {
                    u32 cur = bfirst;
                    while (cur != ~0u) {
                        ++groups;
                        u32 n2 = ~0u;
                        const u32 floor_b = groups == 1 ? 0u : cur + 1;
                        for (u32 i = lane; i < c; i += 64) {
                            const u32 bq = (u32)s_idx[f0 + i] / TB;
                            if (bq != cur && bq >= floor_b && bq != bfirst && bq < n2) n2 = bq;
                        }
                        for (int o = 32; o; o >>= 1) n2 = min(n2, (u32)__shfl_xor(n2, o));
                        cur = n2;
                    }
                }
                unsigned long long old = 0;
                if (lane == 0) old = atomicAdd(&s_cur, (unsigned long long)groups | (1ull << 32));
                old = __shfl(old, 0);
                const u32 pos = rbase + (u32)old, rank = kbase + (u32)(old >> 32);
                const u32 fa = ebatch + f0;
                u32 cur = bfirst, g = 0, bigs = 0;
                while (cur != ~0u) {
                    u32 n2 = ~0u;
                    unsigned long long lo = 0, hi = 0;
                    const u32 floor_b = g == 0 ? 0u : cur + 1;
                    for (u32 i = lane; i < c; i += 64) {
                        const u32 t = s_idx[f0 + i], bq = t / TB;
                        if (bq == cur) {
                            const u32 l = t % TB;
                            if (l < 64) lo |= 1ull << l; else hi |= 1ull << (l - 64);
                        } else if (bq >= floor_b && bq != bfirst && bq < n2) n2 = bq;
                    }
                    for (int o = 32; o; o >>= 1) {
                        lo |= __shfl_xor(lo, o);
                        hi |= __shfl_xor(hi, o);
                        n2 = min(n2, (u32)__shfl_xor(n2, o));
                    }
                    const u32 inf = posting(fa, cur, lo, hi, bigs, lane == 0);
                    if (lane == 0) {
                        out.rec_blk[pos + g] = cur;
                        out.rec_val[pos + g] = ((u64)rank << 32) | inf;
                    }
                    ++g;
                    cur = n2;
                }
            }
            if (nbig) {
                __syncthreads();
                if (tid == 0) s_nbig = 0;
            }
        }
        for (u32 i = tid; i <= HW; i += NT) s_head[i] = 0;   // (the next batch's bitmap: last read in C)
        __syncthreads();   // (the batch's tables are rebuilt from here on)
        FK_T(7);
        eseen += ne;
        bq0 = bq1;
    }
    if (tid == 0) out.nrec[blockIdx.x] = (u32)s_cur;
    for (u32 i = tid; i < NBT; i += NT) out.hist[(size_t)blockIdx.x * NBT + i] = s_hist[i];
    for (int o = 32; o > 0; o >>= 1) holders += __shfl_down(holders, o);
    if (lane == 0 && holders) atomicAdd(&s_work[nb], holders);
    __syncthreads();
    for (u32 i = tid; i <= nb; i += NT)
        if (s_work[i]) atomicAdd(&out.work[i], s_work[i]);
    FK_T(8);
}

// ---- the chunk regions to the padded block lists -------------------------------------------------------------------
// k_ms_scan over the rows k_fkeys wrote (per chunk, not per 2 048 records): same kernel, `chunks` from the host.
// hist[c][b] -> records of block b in the chunks before c; the workgroup that finishes last lays out the block tables
// and leaves the number of list words in scal[1].
__global__ __launch_bounds__(256) void k_fms_scan(u32* __restrict__ hist, const u32 stride, const u32 chunks, u64* __restrict__ scal,
                                                  u32* __restrict__ tot, u32* __restrict__ blk_raw, u32* __restrict__ blk_pos, const u32 nb) {
    __shared__ u32 s_w[4], s_last;
    if (fk_abandoned(scal)) return;
    const u32 per = (chunks + 255u) / 256u;
    const u32 b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 c0 = min(chunks, tid * per), c1 = min(chunks, c0 + per);
    u32 sum = 0;
    for (u32 cb = c0; cb < c1; cb += MS_PER) {
        u32 h[MS_PER];
#pragma unroll
        for (u32 i = 0; i < MS_PER; ++i) h[i] = cb + i < c1 ? hist[(size_t)(cb + i) * stride + b] : 0u;
#pragma unroll
        for (u32 i = 0; i < MS_PER; ++i) sum += h[i];
    }
    u32 inc = sum;
    inc = wave_scan_add(inc);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    u32 run = inc - sum, total = 0;
    for (u32 w = 0; w < 4; ++w) { if (w < wv) run += s_w[w]; total += s_w[w]; }
    for (u32 cb = c0; cb < c1; cb += MS_PER) {
        u32 h[MS_PER];
#pragma unroll
        for (u32 i = 0; i < MS_PER; ++i) h[i] = cb + i < c1 ? hist[(size_t)(cb + i) * stride + b] : 0u;
#pragma unroll
        for (u32 i = 0; i < MS_PER; ++i) {
            if (cb + i < c1) hist[(size_t)(cb + i) * stride + b] = run;
            run += h[i];
        }
    }
    if (tid == 0) {
        tot[b] = total;
        __threadfence();
        s_last = atomicAdd(reinterpret_cast<u32*>(scal + 15), 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    __shared__ u32 s_t[MS_MAXB];
    for (u32 i = tid; i < nb; i += 256) s_t[i] = __hip_atomic_load(&tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid != 0) return;
    u32 raw = 0, pos = 0;
    for (u32 bb = 0; bb < nb; ++bb) {
        const u32 t = s_t[bb];
        blk_raw[bb] = raw;
        blk_pos[bb] = pos;
        pos = ((pos + t + 3u) & ~3u) + WIN;
        raw += t;
    }
    blk_raw[nb] = raw;
    blk_pos[nb] = pos;
    scal[3] = pos;
    scal[1] = raw;   // list words
    reinterpret_cast<u32*>(scal + 15)[0] = 0;   // (the next build's counter)
}

// one workgroup per chunk: its records (rank order) to the padded lists — ranks, posting words, positional masks — in
// rounds of MS_CHUNK records, and one flag byte per tile two records of the same key name (k_tile_flags)
template <u32 MB>
__global__ __launch_bounds__(MS_THREADS) void k_fms_place(const u32* __restrict__ rec_blk, const u64* __restrict__ rec_val,
                                                           const u32* __restrict__ nrec, const u64* __restrict__ bbase, const u32 gb,
                                                           const u32* __restrict__ base, const u32* __restrict__ blk_pos, const u32 nb,
                                                           u32* __restrict__ brk, u32* __restrict__ info, const uint4* __restrict__ bigmask,
                                                           uint4* __restrict__ pmask, unsigned char* __restrict__ flags,
                                                           const u64* __restrict__ scal) {
    constexpr u32 NWV = MS_THREADS / 64, NSL = MS_ROUNDS * NWV, BITS = MB == 256 ? 8u : 10u;
    static_assert(MB == 256 || MB == 1024, "two table sizes");
    __shared__ unsigned short s_cnt[NSL][MB];   // records of block b in (round, wave) slot (<= 64); then: records before the slot (< 2 048)
    __shared__ u32 s_dst[MB];                   // next place of this chunk's records of block b in the padded list
    if (fk_abandoned(scal)) return;
    const u32 n = nrec[blockIdx.x];
    if (!n) return;
    const u32 r0 = (u32)bbase[(size_t)blockIdx.x * gb];   // the chunk's region
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (u32 i = tid; i < MB; i += MS_THREADS) s_dst[i] = (i < nb ? blk_pos[i] : 0u) + base[(size_t)blockIdx.x * MB + i];
    for (u32 g0 = 0; g0 < n; g0 += MS_CHUNK) {
        __syncthreads();   // (the previous round's tables are no longer read)
        for (u32 i = tid; i < NSL * MB / 2; i += MS_THREADS) reinterpret_cast<u32*>(&s_cnt[0][0])[i] = 0;
        u32 blk[MS_ROUNDS], rk[MS_ROUNDS];
        u64 val[MS_ROUNDS];
#pragma unroll
        for (u32 k = 0; k < MS_ROUNDS; ++k) {
            const u32 g = g0 + k * MS_THREADS + tid;
            blk[k] = g < n ? (rec_blk[r0 + g] & (MB - 1)) : ~0u;
            val[k] = g < n ? rec_val[r0 + g] : 0;
        }
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < MS_ROUNDS; ++k) {
            unsigned long long m = __ballot(blk[k] != ~0u);
#pragma unroll
            for (u32 bit = 0; bit < BITS; ++bit) {
                const unsigned long long bal = __ballot((blk[k] >> bit) & 1u);
                m &= ((blk[k] >> bit) & 1u) ? bal : ~bal;
            }
            const unsigned long long below = m & ((1ull << lane) - 1ull);
            rk[k] = (u32)__popcll(below);
            if (blk[k] != ~0u && below == 0) s_cnt[k * NWV + wv][blk[k]] = (unsigned short)__popcll(m);
        }
        __syncthreads();
        u32 tot_b[(MB + MS_THREADS - 1) / MS_THREADS];
        for (u32 bq = tid, z = 0; bq < MB; bq += MS_THREADS, ++z) {
            u32 run = 0;
            for (u32 sl = 0; sl < NSL; ++sl) {
                const u32 c = s_cnt[sl][bq];
                s_cnt[sl][bq] = (unsigned short)run;
                run += c;
            }
            tot_b[z] = run;
        }
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < MS_ROUNDS; ++k) {
            if (blk[k] == ~0u) continue;
            const u32 dst = s_dst[blk[k]] + s_cnt[k * NWV + wv][blk[k]] + rk[k];
            brk[dst] = (u32)(val[k] >> 32);
            info[dst] = (u32)val[k];
            if (pmask) {
                const u32 inf = (u32)val[k];
                uint4 m;
                if (inf >= PM_BIG) m = bigmask[inf & ~PM_BIG];
                else {
                    u32 w4[4] = {0, 0, 0, 0};
                    const u32 cnt = (inf >> 29) + 1;
                    for (u32 x = 0; x < cnt; ++x) {
                        const u32 id = (inf >> (7 * x)) & 127u;
#pragma unroll
                        for (int z = 0; z < 4; ++z) w4[z] |= (id >> 5) == (u32)z ? (1u << (id & 31)) : 0u;
                    }
                    m = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
                pmask[dst] = m;
            }
            if (flags) {   // the records of a key are adjacent, blocks in the order they were walked: every pair is an active tile
                const u32 g = g0 + k * MS_THREADS + tid, r = (u32)(val[k] >> 32), I = blk[k];
                for (u32 j = g + 1; j < n && (u32)(rec_val[r0 + j] >> 32) == r; ++j) {
                    const u32 J = rec_blk[r0 + j] & (MB - 1), A = min(I, J), B = max(I, J);
                    const u64 t = tile_row_start_dev(A, nb) + (B - A);
                    if (!flags[t]) flags[t] = 1;
                }
            }
        }
        __syncthreads();   // (s_dst is read above, advanced below)
        for (u32 bq = tid, z = 0; bq < MB; bq += MS_THREADS, ++z) s_dst[bq] += tot_b[z];
    }
}
