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

constexpr u32 FK_THREADS = 256;
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
    u32 b = blockIdx.x;
    u32 b0 = 0, raw = 0;
    if (b < nbuckets) { b0 = bb.first(b); raw = bb.size(b); }
    u32 size = raw > HB_CAP ? 0 : raw;   // a bucket that does not fit is listed (overflow[1] counts them): the caller falls back
    unsigned long long mykey[EPT], nkey[EPT];
    V mytag[EPT], ntag[EPT];
#pragma unroll
    for (u32 j = 0; j < EPT; ++j) {
        const bool in = tid + j * NT < size;
        mykey[j] = in ? keys[b0 + tid + j * NT] : 0;
        mytag[j] = in ? tags[b0 + tid + j * NT] : V(0);
    }
    while (b < nbuckets) {
        if (raw > HB_CAP && tid == 0) {
            const u32 q = atomicAdd(&overflow[1], 1u);
            big_list[q] = b;   // (one slot per bucket: cannot overflow)
        }
        const u32 bn = b + gridDim.x;
        u32 n0 = 0, nraw = 0;           // bounds of the next bucket
        if (bn < nbuckets) { n0 = bb.first(bn); nraw = bb.size(bn); }
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
        for (int o = 1; o < 64; o <<= 1) { const u32 up = __shfl_up(inc, o); if (lane >= o) inc += up; }
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
            // label pass on the bucket in LDS: one key in (skip + 1), one thread per sampled key (see k_label)
            for (u32 r = tid * (skip + 1); r < nkeys; r += NT * (skip + 1)) {
                const u32 f0 = kst_l[r], f1 = kst_l[r + 1];
                if (f1 - f0 > max_holders) continue;
                u32 mn = ~0u;
                for (u32 x = f0; x < f1; ++x) mn = min(mn, src_of_tag(tag_of(o_tag[x])));
                for (u32 x = f0; x < f1; ++x) {
                    const u32 s = src_of_tag(tag_of(o_tag[x]));
                    if (mn < label[(size_t)s << lshift]) atomicMin(&label[(size_t)s << lshift], mn);
                }
            }
        }
        b = bn; b0 = n0; raw = nraw; size = nsize;
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
// hist (MB > 0): records of chunk c per block, row c.  work (nb <= KG_WORK): diagonal work per block, holders in slot nb.
// ovf[0] is raised when a key cannot be taken (more than FK_CAP holders cannot happen here: a bucket holds at most HB_CAP
// entries — the caller only runs this kernel when no bucket was oversize).
// the grouping gave up, the partition overflowed or a bucket was oversize: the host repeats the build, and the kernels
// queued behind the grouping must not walk tables nobody wrote (known before they start: one decision per launch)
__device__ inline bool fk_abandoned(const u64* __restrict__ scal) {
    return ((u32)scal[9] | (u32)scal[PC_OVF] | reinterpret_cast<const u32*>(scal + 9)[1]) != 0;
}
struct FkOut {
    u32* rec_blk;
    u64* rec_val;
    u32* nrec;
    u32* hist;      // chunks x mb, or nullptr
    u32 mb;
    uint4* bigmask;
    unsigned long long* work;
};
template <class V>
__global__ __launch_bounds__(FK_THREADS) void k_fkeys(const V* __restrict__ tags, const u32* __restrict__ kst, const BucketBounds bb,
                                                      const u64* __restrict__ bsum, const u64* __restrict__ bbase, const u32 nbuckets,
                                                      const u32 gb, const u32* __restrict__ newidx, const u32 nb, const FkOut out,
                                                      const u32 coop, const u64* __restrict__ scal) {
    constexpr u32 NT = FK_THREADS, EPT = FK_CAP / NT;
    __shared__ u32 s_idx[FK_CAP + 4];
    __shared__ u32 s_kst[FK_KEYS + 2];
    __shared__ u32 s_big[FK_KEYS / (KG_COOP / 2) + 8], s_nbig;
    __shared__ unsigned long long s_cur;                    // records committed so far | keys committed so far << 32
    __shared__ u32 s_hist[FK_NB_MAX];
    __shared__ unsigned long long s_work[FK_NB_MAX + 1];
    if (fk_abandoned(scal)) return;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 bk0 = blockIdx.x * gb, bk1 = min(nbuckets, bk0 + gb);
    const u64 base = bbase[bk0];
    const u32 rbase = (u32)base, kbase = (u32)(base >> 32);
    const bool do_work = out.work != nullptr;
    if (tid == 0) { s_cur = 0; s_nbig = 0; }
    if (out.hist) for (u32 i = tid; i < out.mb; i += NT) s_hist[i] = 0;
    if (do_work) for (u32 i = tid; i <= nb; i += NT) s_work[i] = 0;
    unsigned long long holders = 0;
    // this bucket's holders as new source indices (registers), the next bucket's while this one is walked
    u32 kept = 0, nkeys = 0, b0 = 0;
    u32 pidx[EPT];
    auto fetch = [&](const u32 b, u32& kept_o, u32& nkeys_o, u32& b0_o) {
        kept_o = nkeys_o = b0_o = 0;
        if (b < bk1) {
            const u64 s = bsum[b];
            kept_o = min((u32)s, FK_CAP); nkeys_o = min((u32)(s >> 32), FK_KEYS); b0_o = bb.first(b);
        }
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            const u32 i = tid + j * NT;
            pidx[j] = i < kept_o ? newidx[src_of_tag(tag_of(tags[b0_o + i]))] : 0u;
        }
    };
    fetch(bk0, kept, nkeys, b0);
    // one group of a key: block `cur`, members lo | hi -> posting word (a mask of its own for more than INLINE_MAX members)
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
            // a group with a mask takes at least INLINE_MAX + 1 of the key's entries: the masks of the key that starts at
            // entry fa have the places fa / 5, fa / 5 + 1, ... to themselves
            const u32 slot = fa / (INLINE_MAX + 1) + bigs;
            if (store) out.bigmask[slot] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
            inf = BIG | slot;
            ++bigs;
        }
        if (store) {
            if (out.hist) atomicAdd(&s_hist[cur & (out.mb - 1)], 1u);
            if (do_work) {
                holders += cnt;
                if (cnt > 1) atomicAdd(&s_work[cur], (unsigned long long)cnt * (cnt - 1) / 2);
            }
        }
        return inf;
    };
    for (u32 b = bk0; b < bk1; ++b) {
        __syncthreads();   // (the previous bucket's staging is no longer read; first round: the tables are zeroed)
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            const u32 i = tid + j * NT;
            if (i < kept) s_idx[i] = pidx[j];
        }
        const u32 kb0 = fk_kst_first(b0, b);
        for (u32 r = tid; r <= nkeys; r += NT) s_kst[r] = kst[kb0 + r] - b0;
        const u32 c_kept = kept, c_keys = nkeys, c_b0 = b0;
        (void)c_kept;
        fetch(b + 1, kept, nkeys, b0);   // (loads in flight during the walk)
        __syncthreads();
        for (u32 r = tid; r < c_keys; r += NT) {
            const u32 f0 = s_kst[r], c = s_kst[r + 1] - f0;
            if (c > coop) { s_big[atomicAdd(&s_nbig, 1u)] = r; continue; }   // many holders: a whole wave walks this key (below)
            // first walk: the block of the first holder (for most keys the only block), and how many other blocks there are
            const u32 bfirst = s_idx[f0] / TB;
            unsigned long long lo = 0, hi = 0;
            u32 others = 0, nxt = ~0u;
            for (u32 i = 0; i < c; i += 4) {
                u32 t4[4];
#pragma unroll
                for (u32 q = 0; q < 4; ++q) t4[q] = s_idx[f0 + i + q];   // (the staging area has 4 words of slack)
#pragma unroll
                for (u32 q = 0; q < 4; ++q) {
                    if (i + q >= c) break;
                    const u32 t = t4[q], bq = t / TB;
                    if (bq == bfirst) {
                        const u32 l = t % TB;
                        if (l < 64) lo |= 1ull << l; else hi |= 1ull << (l - 64);
                    } else { ++others; if (bq < nxt) nxt = bq; }
                }
            }
            // groups of this key: 1 + distinct other blocks (counted by walking them, ascending: rare)
            u32 groups = 1;
            if (others) {
                u32 cur = nxt;
                while (cur != ~0u) {
                    ++groups;
                    u32 n2 = ~0u;
                    for (u32 i = 0; i < c; ++i) {
                        const u32 bq = s_idx[f0 + i] / TB;
                        if (bq > cur && bq != bfirst && bq < n2) n2 = bq;
                    }
                    cur = n2;
                }
            }
            const unsigned long long old = atomicAdd(&s_cur, (unsigned long long)groups | (1ull << 32));
            const u32 pos = rbase + (u32)old, rank = kbase + (u32)(old >> 32);
            const u32 fa = c_b0 + f0;
            u32 bigs = 0;
            out.rec_blk[pos] = bfirst;
            out.rec_val[pos] = ((u64)rank << 32) | posting(fa, bfirst, lo, hi, bigs, true);
            if (others) {
                u32 cur = nxt, g = 1;
                while (cur != ~0u) {
                    u32 n2 = ~0u;
                    lo = hi = 0;
                    for (u32 i = 0; i < c; ++i) {
                        const u32 t = s_idx[f0 + i], bq = t / TB;
                        if (bq == cur) {
                            const u32 l = t % TB;
                            if (l < 64) lo |= 1ull << l; else hi |= 1ull << (l - 64);
                        } else if (bq > cur && bq != bfirst && bq < n2) n2 = bq;
                    }
                    out.rec_blk[pos + g] = cur;
                    out.rec_val[pos + g] = ((u64)rank << 32) | posting(fa, cur, lo, hi, bigs, true);
                    ++g;
                    cur = n2;
                }
            }
        }
        __syncthreads();
        // keys with many holders, one wave each: the lanes share the walk, the masks and the next block are reduced
        const u32 nbig = s_nbig;
        for (u32 q = wv; q < nbig; q += NT / 64) {
            const u32 r = s_big[q];
            const u32 f0 = s_kst[r], c = s_kst[r + 1] - f0;
            const u32 bfirst = s_idx[f0] / TB;
            // count the groups: distinct blocks among the holders (first block first, then ascending)
            u32 groups = 0;
            {
                u32 cur = bfirst;
                while (cur != ~0u) {
                    ++groups;
                    u32 n2 = ~0u;
                    const u32 floor_b = groups == 1 ? 0u : cur + 1;
                    for (u32 i = lane; i < c; i += 64) {
                        const u32 bq = s_idx[f0 + i] / TB;
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
            const u32 fa = c_b0 + f0;
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
        __syncthreads();
        if (tid == 0) s_nbig = 0;
    }
    __syncthreads();
    if (tid == 0) out.nrec[blockIdx.x] = (u32)s_cur;
    if (out.hist) for (u32 i = tid; i < out.mb; i += NT) out.hist[(size_t)blockIdx.x * out.mb + i] = s_hist[i];
    if (do_work) {
        for (int o = 32; o > 0; o >>= 1) holders += __shfl_down(holders, o);
        if (lane == 0 && holders) atomicAdd(&s_work[nb], holders);
        __syncthreads();
        for (u32 i = tid; i <= nb; i += NT)
            if (s_work[i]) atomicAdd(&out.work[i], s_work[i]);
    }
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
    for (int o = 1; o < 64; o <<= 1) { const u32 up = __shfl_up(inc, o); if ((int)lane >= o) inc += up; }
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
