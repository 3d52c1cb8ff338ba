// Stage 1 of the engine: device kernels that turn the sketches (sorted uint64 runs) — or an inverted
// index — into per-block posting lists, the source order and the join's tile bitmap.
// Included by engine.hip inside namespace ksp (one translation unit: the kernels are templates over the
// tag type and share the constants defined there).  Host orchestration: build_impl in engine.hip.
#pragma once

// Entry tags.  Canonical form: (block << 8) | local id in 32 bits; weighted input carries the key's weight in
// the high half of a 64-bit tag.  Unweighted sets of up to 65 536 sources use the compact form — the source
// index itself in 16 bits — which takes 2 bytes per entry out of every pass of the two radix sorts.
typedef unsigned short u16;
template <class V> __host__ __device__ inline u32 tag_of(V v) { return (u32)v; }
template <> __host__ __device__ inline u32 tag_of<u16>(u16 v) { return (((u32)v >> 7) << 8) | ((u32)v & 127u); }
template <class V> __host__ __device__ inline V make_tag(u32 canon, u32 w) { (void)w; return (V)canon; }
template <> __host__ __device__ inline u64 make_tag<u64>(u32 canon, u32 w) { return ((u64)w << 32) | canon; }
template <> __host__ __device__ inline u16 make_tag<u16>(u32 canon, u32 w) { (void)w; return (u16)(((canon >> 8) << 7) | (canon & 127u)); }
template <class V> __host__ __device__ inline u32 weight_of(V v) { (void)v; return 0u; }
template <> __host__ __device__ inline u32 weight_of<u64>(u64 v) { return (u32)(v >> 32); }
static_assert(TB == 128, "compact tags: 7-bit local ids");
// Inclusive prefix sum over the 64 lanes of a wave on the vector unit alone: row_shr 1 / 2 / 4 / 8 inside the rows of 16
// lanes, then the last lane of a row broadcast to the next row (row_bcast:15 on rows 1 and 3) and lane 31 to the upper
// half (row_bcast:31).  The shuffle formulation (__shfl_up, six steps) compiles to six ds_bpermute_b32 — six dependent
// round trips through the LDS pipe, which the workgroups of stage 1 keep busy with their own traffic.
__device__ inline u32 wave_scan_add(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return x;
}
// The word of lane ^ J through the vector unit (gfx950: v_permlane32_swap / v_permlane16_swap across the halves and the
// rows of a wave, DPP row rotation / mirrors / quad permutations inside a row) — __shfl_xor compiles to ds_bpermute_b32,
// a round trip through the LDS pipe.  All lanes of the wave must be active.
template <int J>
__device__ inline u32 xor_lane(const u32 d, const int lane) {   // d of lane ^ J
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "a power of two below 64");
    if (J == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)d, 0xB1, 0xf, 0xf, false);    // quad_perm:[1,0,3,2]
    if (J == 2) return (u32)__builtin_amdgcn_update_dpp(0, (int)d, 0x4E, 0xf, 0xf, false);    // quad_perm:[2,3,0,1]
    if (J == 4) {   // lane ^ 7 (row_half_mirror), then lane ^ 3 (quad_perm:[3,2,1,0])
        const u32 t = (u32)__builtin_amdgcn_update_dpp(0, (int)d, 0x141, 0xf, 0xf, false);
        return (u32)__builtin_amdgcn_update_dpp(0, (int)t, 0x1B, 0xf, 0xf, false);
    }
    if (J == 8) return (u32)__builtin_amdgcn_update_dpp(0, (int)d, 0x128, 0xf, 0xf, false);   // row_ror:8
    if (J == 16) {   // the odd rows of the first operand change places with the even rows of the second; both are d
        const auto r = __builtin_amdgcn_permlane16_swap(d, d, false, false);   // {[r0 r0 r2 r2], [r1 r1 r3 r3]}
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(d, d, false, false);   // {[lower lower], [upper upper]}
    return (lane & 32) ? r[0] : r[1];
}
// all-lanes reductions over a wave (butterfly; every lane ends with the result)
template <class Op>
__device__ inline u32 wave_all(u32 x, const int lane, Op op) {
    x = op(x, xor_lane<1>(x, lane));
    x = op(x, xor_lane<2>(x, lane));
    x = op(x, xor_lane<4>(x, lane));
    x = op(x, xor_lane<8>(x, lane));
    x = op(x, xor_lane<16>(x, lane));
    x = op(x, xor_lane<32>(x, lane));
    return x;
}
__device__ inline u32 wave_all_min(const u32 x, const int lane) { return wave_all(x, lane, [](u32 a, u32 b) { return min(a, b); }); }
__device__ inline u32 wave_all_max(const u32 x, const int lane) { return wave_all(x, lane, [](u32 a, u32 b) { return max(a, b); }); }
__device__ inline u64 wave_all_or(const u64 x, const int lane) {
    const u32 lo = wave_all((u32)x, lane, [](u32 a, u32 b) { return a | b; }), hi = wave_all((u32)(x >> 32), lane, [](u32 a, u32 b) { return a | b; });
    return (u64)lo | ((u64)hi << 32);
}

// A read-back that rides along with the next kernel instead of taking a dispatch of its own (~4.5 us each): the first wave
// of the kernel's first workgroup copies a few words of the scalar block to pinned host memory and writes a sequence
// number behind them, before it does anything else; the host polls the number (engine.hip: wait_readback).
struct Rider {
    const u64* src;
    u64* dst_host;
    unsigned long long* flag_host;
    unsigned long long seq;
    u32 words;   // 0: nothing rides along
};
__device__ inline void rider_run(const Rider& r) {
    if (r.words && blockIdx.x == 0 && threadIdx.x < 64) {
        if (threadIdx.x < r.words) __hip_atomic_store(r.dst_host + threadIdx.x, r.src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store(r.flag_host, r.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ------------------------------------------------------------------------------------
// stage 1 kernels
// ------------------------------------------------------------------------------------
// (timing build, make fktime: thread 0 of every workgroup adds the shader-clock time since its previous mark to
//  st_time[slot]; tools/fk_times.py prints the shares.  Slots: 16.. k_key_groups, 32.. k_bucket_group, 48.. k_seg_scatter)
#ifdef KSP_FKTIME
__device__ unsigned long long st_time[64];
// (sums in registers, one atomic per stage when the workgroup ends: a global atomic per mark is itself memory traffic
//  that the kernel's next s_waitcnt vmcnt(0) waits for — timers built that way moved half of k_bucket_group's time into
//  the stage that held the wait)
#define ST_BEGIN() unsigned long long st_last_ = clock64(), st_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define ST_T(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); st_acc_[(k) & 15] += now_ - st_last_; st_last_ = now_; } } while (0)
#define ST_END(base) do { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 16; ++q_) if (st_acc_[q_]) atomicAdd(&st_time[(base) + q_], st_acc_[q_]); } } while (0)
#else
#define ST_BEGIN() do { } while (0)
#define ST_T(k) do { } while (0)
#define ST_END(base) do { } while (0)
#endif

// One workgroup per source: tag each entry with (block << 8 | local id) [and weight].
// Weighted mode also records the source's weight sum (the bound of any pair counter that source
// takes part in; unweighted: k_src_size).
template <class V, bool W>
__global__ void k_tag(const u64* __restrict__ off, const u32* __restrict__ wts, V* __restrict__ vals,
                      u32* __restrict__ src_bound) {
    __shared__ unsigned long long acc;
    const u32 s = blockIdx.x;
    const u64 b = off[s], e = off[s + 1];
    const u32 tag = ((s / TB) << 8) | (s % TB);
    if (W) { if (threadIdx.x == 0) acc = 0; __syncthreads(); }
    unsigned long long part = 0;
    for (u64 i = b + threadIdx.x; i < e; i += blockDim.x) {
        if (W) { const u32 w = wts[i]; part += w; vals[i] = make_tag<V>(tag, w); }
        else vals[i] = make_tag<V>(tag, 0u);
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

// everything a build needs per source before the partition, in one launch: the largest key (= largest last element
// of the sorted runs), the bound of a source's pair counters (its size) and the four identity maps
__global__ void k_prep_sources(const u64* __restrict__ keys, const u64* __restrict__ off, unsigned long long* __restrict__ max_out,
                               u32* __restrict__ src_bound, u32* __restrict__ p0, u32* __restrict__ p1, u32* __restrict__ p2,
                               u32* __restrict__ p3, u32 n_sources) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (s < n_sources) {
        const u64 b = off[s], e = off[s + 1];
        if (e > b) v = keys[e - 1];
        const u64 c = e - b;
        src_bound[s] = (u32)(c > 0xFFFFFFFFull ? 0xFFFFFFFFull : c);
        p0[s] = s; p1[s] = s; p2[s] = s; p3[s] = s;
    }
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned long long)__shfl_down(v, o));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(max_out, v);
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
// the four identity maps of a build in one launch (iota, source order, new index, label)
__global__ void k_iota4(u32* __restrict__ p0, u32* __restrict__ p1, u32* __restrict__ p2, u32* __restrict__ p3, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { p0[i] = i; p1[i] = i; p2[i] = i; p3[i] = i; }
}
__device__ inline u32 src_of_tag(u32 t) { return (t >> 8) * TB + (t & 0xFFu); }
template <class V>
__global__ void k_label(const V* __restrict__ vals, const u32* __restrict__ first, u32* __restrict__ label,
                        const int lshift, const u32 skip, const u32 max_holders, u32 n_keys, const u64* __restrict__ scal,
                        const Rider rider) {
    rider_run(rider);
    // scal != NULL: launched before the host has read the grouping's results back — the key count comes from the
    // device (n_keys is an upper bound) and nothing is touched when the grouping gave up (the build is repeated)
    if (scal) {
        if ((u32)scal[9] | (u32)scal[14]) return;
        n_keys = min(n_keys, (u32)scal[2]);
    }
    const u32 r = (blockIdx.x * blockDim.x + threadIdx.x) * (skip + 1);   // one thread per sampled key (skip = 2^k - 1)
    if (r >= n_keys) return;
    const u32 f0 = first[r], f1 = first[r + 1];   // (first[] has a sentinel: first[U] = number of entries)
    // a key held by very many sources says nothing about who is related to whom — it would only pull
    // unrelated clusters under one label
    if (f1 - f0 > max_holders) return;
    u32 mn = ~0u;   // (the entries of a key are in no particular order after the bucket grouping)
    for (u32 e = f0; e < f1; ++e) mn = min(mn, src_of_tag(tag_of(vals[e])));
    for (u32 e = f0; e < f1; ++e) {
        const u32 s = src_of_tag(tag_of(vals[e]));
        if (mn < label[(size_t)s << lshift]) atomicMin(&label[(size_t)s << lshift], mn);
    }
}
// While they are being lowered the labels sit on memory lines of their own (label s at index s << lshift):
// atomics on one 128-byte line are served one at a time, and a few thousand sources share a few hundred lines.
__global__ void k_label_spread(u32* __restrict__ wide, const int lshift, u32 n) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) wide[(size_t)s << lshift] = s;
}
__global__ void k_label_gather(const u32* __restrict__ wide, const int lshift, u32* __restrict__ label, u32 n) {
    const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) label[s] = wide[(size_t)s << lshift];
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
        vals[e] = make_tag<V>(tag, W ? w : 0u);
        rk[e] = r;
        if (w) atomicAdd(&src_bound[s], w);   // (the caller guarantees sums below 2^32)
    }
}
// order[i] = i-th source in (label, id) order  ->  newidx[order[i]] = i; order itself is the inverse map
__global__ void k_perm(const u32* __restrict__ order, u32* __restrict__ newidx, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) newidx[order[i]] = i;
}
// both at once: newidx[order[i]] = i and the largest per-source bound of every block of the new order
__global__ void k_perm_bound(const u32* __restrict__ order, u32* __restrict__ newidx, const u32* __restrict__ src_bound,
                             u32* __restrict__ blk_max, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 s = order[i];
    newidx[s] = i;
    atomicMax(&blk_max[i / TB], src_bound[s]);
}
template <class V>
__global__ void k_retag(V* __restrict__ vals, const u32* __restrict__ newidx, u64 n) {
    // 16 bytes of tags per thread (2-byte tags one at a time leave most of every memory transaction unused)
    constexpr u32 VEC = 16 / sizeof(V);
    union Pack { uint4 q; V v[VEC]; };
    const u64 e0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (e0 >= n) return;
    if (e0 + VEC <= n) {
        Pack p;
        p.q = *reinterpret_cast<const uint4*>(vals + e0);
#pragma unroll
        for (u32 j = 0; j < VEC; ++j) {
            const u32 ni = newidx[src_of_tag(tag_of(p.v[j]))];
            p.v[j] = make_tag<V>(((ni / TB) << 8) | (ni % TB), weight_of(p.v[j]));
        }
        *reinterpret_cast<uint4*>(vals + e0) = p.q;
    } else {
        for (u64 e = e0; e < n; ++e) {
            const V v = vals[e];
            const u32 ni = newidx[src_of_tag(tag_of(v))];
            vals[e] = make_tag<V>(((ni / TB) << 8) | (ni % TB), weight_of(v));
        }
    }
}
// ---- block boundaries that respect the clusters ------------------------------------------------------------
// Sources with the same label (a cluster) are contiguous after the (label, id) sort; cutting that order every 128
// sources lets most clusters straddle a block boundary: their keys then sit in two list words instead of one and the
// off-diagonal tile of the two blocks is as heavy as a diagonal one (C2: clusters of up to 100 sources, 253 active
// off-diagonal tiles, the longest workgroups of the join).  So a cluster of at most 128 sources that would cross the
// end of a block starts the next block instead, and the slots it leaves behind stay empty ("holes": engine indices
// no source maps to).  The number of blocks is fixed by the host before the labels exist, so the holes have a budget
// (blocks x 128 - sources); when it is spent the remaining boundaries fall where they fall.  Exactness never depends
// on where the boundaries are.
//
// k_pack_blocks, ONE workgroup: the distance of every sorted position to the head of its cluster goes to LDS (one
// byte, capped); from it every position gets, as if a block started there, where the next block would start (one byte:
// 128 or less); then one lane walks the chain of block starts — one LDS read per block — spending the budget of holes.
// bstart[b] = sorted position where block b begins (= sources in the blocks before it), bstart[nb] = n.
constexpr u32 PACK_MAX = 65536;   // sources (two byte tables in LDS); larger sets keep plain cuts (blocks_for)
__global__ __launch_bounds__(1024) void k_pack_blocks(const u32* __restrict__ labs, const u32 n, const u32 nb,
                                                      u32* __restrict__ bstart) {
    __shared__ unsigned char s_d[PACK_MAX], s_nx[PACK_MAX];
    __shared__ u32 s_head[1024];   // last cluster head in or before the thread's chunk, + 1 (0: none yet)
    __shared__ u32 s_used;
    const u32 tid = threadIdx.x;
    // head flags, read in full lines (a thread's chunk below is contiguous: it would read the labels one word per line)
    for (u32 p = tid; p < n; p += 1024) s_d[p] = (p == 0 || labs[p] != labs[p - 1]) ? 1 : 0;
    __syncthreads();
    const u32 per = (n + 1023u) / 1024u;
    const u32 p0 = min(n, tid * per), p1 = min(n, p0 + per);
    u32 last = 0;
    for (u32 p = p0; p < p1; ++p)
        if (s_d[p]) last = p + 1;
    // inclusive max-scan over the threads (heads ascend with the position): inside a wave on the vector unit, the sixteen
    // wave maxima through LDS — two barriers instead of the twenty of a shared-memory scan
    {
        u32 x = last;
        x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));   // row_shr:1
        x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));   // row_shr:2
        x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));   // row_shr:4
        x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));   // row_shr:8
        x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
        x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
        if ((tid & 63) == 63) s_head[tid >> 6] = x;   // (s_head[0..15]: the waves' maxima)
        __syncthreads();
        for (u32 w = 0; w < (tid >> 6); ++w) x = max(x, s_head[w]);
        __syncthreads();
        s_head[tid] = x;
    }
    __syncthreads();
    {
        u32 head = tid ? s_head[tid - 1] : 0u;   // (+ 1; position 0 is always a head, so 0 never survives the first store)
        for (u32 p = p0; p < p1; ++p) {
            if (s_d[p]) head = p + 1;
            s_d[p] = (unsigned char)min(p - (head - 1), 255u);
        }
    }
    __syncthreads();
    // a block that starts at b ends at x = b + 128 — or, when a cluster of at most 128 sources that began inside the block
    // would straddle x, in front of that cluster: s_nx[b] = holes that costs (0: the block is full)
    for (u32 b = tid; b < n; b += 1024) {
        const u32 x = b + (u32)TB;
        u32 holes = 0;
        if (x < n) {
            const u32 d = s_d[x];
            if (d != 0 && d < (u32)TB) {
                const u32 hx = x - d;
                const bool big = hx + (u32)TB < n && s_d[hx + (u32)TB] >= (u32)TB;   // more than 128 members: no block holds it
                if (!big) holes = d;
            }
        }
        s_nx[b] = (unsigned char)holes;
    }
    __syncthreads();
    if (tid == 0) {
        const u32 budget = nb * (u32)TB - n;
        u32 b = 0, k = 0, spent = 0;
        while (true) {
            bstart[k] = b;
            if (b + (u32)TB >= n) break;
            const u32 h = s_nx[b];
            const bool take = h && spent + h <= budget;
            if (take) spent += h;
            b += (u32)TB - (take ? h : 0u);
            ++k;
        }
        s_used = k + 1;   // (<= nb: every block but the last covers 128 slots, sources + holes <= nb x 128)
    }
    __syncthreads();
    for (u32 j = s_used + tid; j <= nb; j += 1024) bstart[j] = n;   // the blocks the holes did not need stay empty
}
// slot -> source (inv, ~0 for a hole), source -> slot (newidx) and the largest per-source bound of every block
__global__ void k_place_sources(const u32* __restrict__ sorted_src, const u32* __restrict__ bstart, u32* __restrict__ newidx,
                                u32* __restrict__ inv, const u32* __restrict__ src_bound, u32* __restrict__ blk_max, const u32 nb) {
    const u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nb * (u32)TB) return;
    const u32 k = slot / (u32)TB, j = slot % (u32)TB;
    const u32 b0 = bstart[k], b1 = bstart[k + 1];
    u32 s = ~0u, bound = 0;
    if (j < b1 - b0) {
        s = sorted_src[b0 + j];
        newidx[s] = slot;
        bound = src_bound[s];
    }
    inv[slot] = s;
    // (a wave's 64 slots lie in one block: one atomic per wave — 128 on the same word are served one at a time)
    for (int o = 32; o > 0; o >>= 1) bound = max(bound, (u32)__shfl_xor(bound, o));
    if ((threadIdx.x & 63) == 0 && bound) atomicMax(&blk_max[k], bound);
}
// per block (of the new order): the largest per-source bound
__global__ void k_blk_bound(const u32* __restrict__ src_bound, const u32* __restrict__ newidx, u32* __restrict__ blk_max,
                            u32 n_sources) {
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_sources) atomicMax(&blk_max[newidx[s] / TB], src_bound[s]);
}


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

// ---- grouping by hash bucket: two partition passes instead of a full sort ------------------------
// The engine does not need the keys in order — only equal keys together, a consistent dense rank per
// key, and the keys held by one source dropped.  So the entries are only *partitioned* by their top
// `pb` key bits (two 8-bit radix passes; sketch hashes are uniform), and one workgroup at a time
// groups a bucket's entries in an LDS hash table keyed by the full 64-bit key: count per key, drop the
// singletons, number the kept keys (slot order), and give every kept entry its place inside the
// bucket — the entries of a key contiguous.  One small scan over the bucket totals and a streaming
// kernel then move tags and ranks to their final places.  This replaces the other two radix passes,
// the mixed-run fix-up and the prune scan.  A bucket that does not fit (skewed keys) sends the build
// back to the sort path.
// (A single-kernel variant — bucket offsets by decoupled look-back, tickets for the order — was
//  measured at 1.1 ms against 0.41 ms for the same kernel without the look-back: the persistent
//  workgroups move in step, so every look-back walks hundreds of predecessors.  Two kernels it is.)
// The key-by-key list build works on chunks of 4 096 kept entries and needs, per chunk, the rank of its first entry:
// crank[c] = rank of entry c x 4 096.  Nothing else reads a rank per entry on the default path, so the grouping writes
// these 1-in-4 096 ranks instead of 4 bytes for every kept entry (135 MB per C2 build); the sort-by-block fallback
// expands first[] into the per-entry array when it needs it (k_rank_fill).
constexpr u32 CR_CHUNK = 4096;
constexpr u32 HB_CAP = 3072;     // entries per bucket (one 16-bit slot index each in LDS)
constexpr u32 HB_SLOTS = 4096;   // hash slots per bucket (power of two; more than HB_CAP: never full)
constexpr u32 HB_THREADS = 512;
// The host picks the bucket count for at most this mean size over [0, 2^key_bits); real hash ranges are
// not powers of two (2^64 / scaled is 0.51 x 2^55 for scaled = 1000), so the mean can be twice that.
constexpr u32 HB_MEAN = 800;
// The hand-written partition spreads the keys evenly over any number of buckets: the same mean the power-of-two
// partition ends up with on real hash ranges (2^64 / scaled covers just over half of its power of two)
constexpr u32 HB_HAND_MEAN = 2000, HB_HAND_MEAN_MAX = 2400;
constexpr u32 HB_KEPT = 1u << 15, HB_FIRST = 1u << 14;   // per-entry record (16 bits): kept | first of its key | place inside the
                                                         // bucket's kept entries (< 4 096); the key's rank is the number of
                                                         // "first" places below — counted by k_bucket_emit, not stored
constexpr u32 HB_EMIT = 8;       // buckets per workgroup of the emit kernel
constexpr u32 HB_BIG_DISTINCT = 3072;   // k_bucket_big: distinct keys per oversize bucket (any number of such buckets: the list holds one slot per bucket)

// Where a bucket's entries sit: dense (bucket b = [start[b], start[b + 1]), the paged / library partitions) or one
// fixed range of `cap` places per bucket of which cnt[b] are used (the segment partition, k_seg_scatter).
struct BucketBounds {
    const u32* start;
    const u32* cnt;
    u32 cap;
    __device__ u32 first(const u32 b) const { return cap ? b * cap : start[b]; }
    __device__ u32 size(const u32 b) const { return cap ? min(cnt[b], cap) : start[b + 1] - start[b]; }
};

// slot of a key in a bucket's table (12 bits; the callers mask to their table size): the halves of the key folded, one
// 32-bit multiply (a 64 x 64 multiply is six quarter-rate vector multiplies per entry — a tenth of k_bucket_group).  The
// keys of a bucket agree in their top bits only, so the fold keeps what distinguishes them.
__device__ inline u32 hb_slot(const unsigned long long key) { return (((u32)key ^ (u32)(key >> 32)) * 0x9E3779B1u) >> 20; }

__global__ void k_bucket_bounds(const u64* __restrict__ keys, u64 n, int shiftb, u32 nbuckets, u32* __restrict__ bstart) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbuckets) return;
    u64 lo = 0, hi = n;
    if (b == nbuckets) lo = n;
    while (lo < hi) {
        const u64 mid = lo + ((hi - lo) >> 1);
        if (((keys[mid] >> shiftb) & (u64)(nbuckets - 1)) < (u64)b) lo = mid + 1; else hi = mid;   // (the partition's digit)
    }
    bstart[b] = (u32)lo;
}

// Persistent workgroups, buckets b = blockIdx.x, += gridDim.x: while a bucket is grouped, the bounds
// and then the keys of the workgroup's next bucket are already on their way (the kernel is a chain
// of memory round trips otherwise).  bsum[] is zero at launch (trailing empty buckets are not visited).
__global__ __launch_bounds__(HB_THREADS, 6) void k_bucket_group(const u64* __restrict__ keys, const BucketBounds bb,
                                                             u32 nbuckets, u32 nw, unsigned short* __restrict__ rec,
                                                             u64* __restrict__ bsum, u32* __restrict__ overflow,
                                                             u32* __restrict__ big_list) {
    constexpr u32 NT = HB_THREADS, NWV = NT / 64;
    __shared__ unsigned long long tkey[HB_SLOTS + 1];
    __shared__ u32 tcnt2[HB_SLOTS / 2 + 1];   // entries per key, two 16-bit counters per word
    __shared__ unsigned short eslot[HB_CAP];
    __shared__ u32 wpart[NWV];
    // after the inserts the keys are dead and their storage holds, per slot, the first place of the key's
    // entries | its rank << 16, and the fill cursor
    u32* toff = (u32*)tkey;
    u32* tfill = toff + (HB_SLOTS + 1);
    constexpr unsigned long long EMPTY = ~0ull;
    constexpr u32 EPT = HB_CAP / NT;
    constexpr u32 PER = (HB_SLOTS + 1 + NT - 1) / NT;
    const u32 tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // bounds two buckets ahead (round 3): the words of bucket b + 2 x grid are loaded at the head of pass b through a
    // lane-private zero — the compiler cannot tell that the address is uniform, so they stay in vector registers, nobody
    // waits for them — and first looked at when the pass ends.  Before, the next bucket's bounds were two or three scalar
    // loads at the head of every pass, each waited for on the spot (s_waitcnt lgkmcnt(0)): memory round trips in front of
    // the table initialisation, in the stage that was half of every pass (tools/st_times.py).
    u32 vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    u32 b = blockIdx.x;
    u32 b0 = 0, raw = 0, n0 = 0, nraw = 0;
    if (b < nbuckets) { b0 = bb.first(b); raw = bb.size(b); }
    if (b + gridDim.x < nbuckets) { n0 = bb.first(b + gridDim.x); nraw = bb.size(b + gridDim.x); }
    u32 size = raw > HB_CAP ? 0 : raw;   // a bucket that does not fit is skipped here: k_bucket_big takes it
    unsigned long long mykey[EPT], nkey[EPT];
#pragma unroll
    for (u32 j = 0; j < EPT; ++j) mykey[j] = tid + j * NT < size ? keys[b0 + tid + j * NT] : 0;
    ST_BEGIN();
#ifdef KSP_FKTIME
    const bool st_probe = st_time[63] != 0;   // (ksp_debug_sttime(out, 2) switches the probe waits on)
#endif
    while (b < nbuckets) {
        if (raw > HB_CAP && tid == 0) {   // left to k_bucket_big (overflow[1] counts them)
            const u32 q = atomicAdd(&overflow[1], 1u);
            big_list[q] = b;   // (one slot per bucket: cannot overflow)
        }
        const u32 bn = b + gridDim.x, bnn = bn + gridDim.x;
        u32 la = 0, lb = 0;   // (the words as loaded; first / size are worked out when the pass ends)
        if (bnn < nbuckets) {
            if (bb.cap) la = bb.cnt[bnn + vzero];
            else { la = bb.start[bnn + vzero]; lb = bb.start[bnn + 1 + vzero]; }
        }
        // the table is as large as the bucket needs (load at most 13/16 even if no two keys are equal)
        const u32 slots = size <= 416 ? 512u : size <= 832 ? 1024u : size <= 1664 ? 2048u : HB_SLOTS;
        for (u32 i = tid; i <= slots; i += NT) tkey[i] = EMPTY;
        for (u32 i = tid; i <= slots / 2; i += NT) tcnt2[i] = 0;
        __syncthreads();
        ST_T(32);
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
        // the next bucket's keys: in flight during the scan and the placement
        const u32 nsize = nraw > HB_CAP ? 0 : nraw;
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) nkey[j] = tid + j * NT < nsize ? keys[n0 + tid + j * NT] : 0;
        __syncthreads();
        ST_T(33);
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
        ST_T(34);
        u32 run = inc - mine;
        for (int w = 0; w < wv; ++w) run += wpart[w];
        if (tid == NT - 1) {   // the last thread's inclusive sum: the bucket's total
            const u32 tot = run + mine;
            bsum[b] = (u64)(tot & 0xFFFFu) | ((u64)(tot >> 16) << 32);
        }
#pragma unroll
        for (u32 j = 0; j < PER; ++j) {
            const u32 sl = tid * per + j;
            if (j < per && sl <= slots) { toff[sl] = run; tfill[sl] = 0; }
            if (cnt[j] >= 2) run += cnt[j] | (1u << 16);
        }
        __syncthreads();
        ST_T(35);
#ifdef KSP_FKTIME
        if (st_probe) { __builtin_amdgcn_s_waitcnt(0x0F70); ST_T(37); }   // (probe: what the next keys still need here)
#endif
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            const u32 i = tid + j * NT;
            if (i >= size) break;
            const u32 sl = eslot[i];
            u32 r = 0;
            if (((tcnt2[sl >> 1] >> (16 * (sl & 1))) & 0xFFFFu) >= 2) {
                const u32 fill = atomicAdd(&tfill[sl], 1u);
                const u32 t = toff[sl];
                r = HB_KEPT | (fill == 0 ? HB_FIRST : 0u) | ((t & 0xFFFFu) + fill);
            }
            rec[b0 + i] = (unsigned short)r;
        }
#ifdef KSP_FKTIME
        if (st_probe) { ST_T(38); __builtin_amdgcn_s_waitcnt(0x0F70); ST_T(39); }   // (probe: the acknowledgement of the record stores)
#endif
        b = bn; b0 = n0; raw = nraw; size = nsize;
        {
            const u32 ua = __builtin_amdgcn_readfirstlane(la), ub = __builtin_amdgcn_readfirstlane(lb);
            n0 = bnn < nbuckets ? (bb.cap ? bnn * bb.cap : ua) : 0u;
            nraw = bnn < nbuckets ? (bb.cap ? min(ua, bb.cap) : ub - ua) : 0u;
        }
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) mykey[j] = nkey[j];
        __syncthreads();
        ST_T(36);   // the table is rebuilt from here on
    }
    ST_END(32);
}

// Buckets above HB_CAP entries — a key held by thousands of sources lands in one — one workgroup each, with no
// per-entry state in LDS: the bucket's keys are streamed from global memory twice (count, then place).  PART 0
// leaves the bucket's totals in bsum[] before the scan over the buckets; PART 1, after it, groups again and
// writes tags, ranks and first[] straight to their final places.  More than HB_BIG_DISTINCT distinct keys in
// such a bucket (thousands of keys under one prefix) is the one case left for the sort path (*overflow).
template <class V, int PART>
__global__ __launch_bounds__(HB_THREADS) void k_bucket_big(const u64* __restrict__ keys, const V* __restrict__ vals,
                                                           const BucketBounds bb, const u32* __restrict__ big_list,
                                                           u32* __restrict__ overflow, u64* __restrict__ bsum,
                                                           const u64* __restrict__ bbase, V* __restrict__ vals2,
                                                           u32* __restrict__ rank2, u32* __restrict__ first, u32* __restrict__ crank) {
    constexpr u32 NT = HB_THREADS, NWV = NT / 64;
    constexpr unsigned long long EMPTY = ~0ull;
    __shared__ unsigned long long tkey[HB_SLOTS + 1];
    __shared__ u32 tcnt[HB_SLOTS + 1], toff[HB_SLOTS + 1], trank[HB_SLOTS + 1], tfill[HB_SLOTS + 1];
    __shared__ u32 wpe[NWV], wpk[NWV], s_distinct;
    const u32 n_big = overflow[1];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (u32 q = blockIdx.x; q < n_big; q += gridDim.x) {   // (a few workgroups walk the list: it is empty for most inputs)
    const u32 b = big_list[q];
    const u32 b0 = bb.first(b), size = bb.size(b);
    __syncthreads();   // (the table of the previous bucket is no longer read)
    for (u32 i = tid; i <= HB_SLOTS; i += NT) { tkey[i] = EMPTY; tcnt[i] = 0; tfill[i] = 0; }
    if (tid == 0) s_distinct = 0;
    __syncthreads();
    auto slot_of = [&](const unsigned long long key, const bool insert) -> u32 {
        if (key == EMPTY) return HB_SLOTS;
        u32 h = hb_slot(key) & (HB_SLOTS - 1);
        while (true) {
            if (insert) {
                const unsigned long long prev = atomicCAS(&tkey[h], EMPTY, key);
                if (prev == EMPTY) { atomicAdd(&s_distinct, 1u); return h; }
                if (prev == key) return h;
                if (s_distinct > HB_BIG_DISTINCT) return HB_SLOTS;   // (the table is filling up: given up below)
            } else if (tkey[h] == key) return h;
            h = (h + 1) & (HB_SLOTS - 1);
        }
    };
    for (u32 i = tid; i < size; i += NT) atomicAdd(&tcnt[slot_of(keys[b0 + i], true)], 1u);
    __syncthreads();
    if (s_distinct > HB_BIG_DISTINCT) {   // too many distinct keys for the table: the sort path
        if (tid == 0) *overflow = 1;
        continue;
    }
    // exclusive scan over the slots: kept entries, kept keys
    constexpr u32 PER = (HB_SLOTS + 1 + NT - 1) / NT;
    u32 me = 0, mk = 0;
    for (u32 j = 0; j < PER; ++j) {
        const u32 sl = tid * PER + j;
        const u32 c = sl <= HB_SLOTS ? tcnt[sl] : 0;
        if (c >= 2) { me += c; ++mk; }
    }
    const u32 ie = wave_scan_add(me), ik = wave_scan_add(mk);
    if (lane == 63) { wpe[wv] = ie; wpk[wv] = ik; }
    __syncthreads();
    u32 re = ie - me, rk = ik - mk, te = 0, tk = 0;
    for (u32 w = 0; w < NWV; ++w) { if (w < (u32)wv) { re += wpe[w]; rk += wpk[w]; } te += wpe[w]; tk += wpk[w]; }
    if (PART == 0) {
        if (tid == 0) bsum[b] = (u64)te | ((u64)tk << 32);
        continue;
    }
    for (u32 j = 0; j < PER; ++j) {
        const u32 sl = tid * PER + j;
        if (sl > HB_SLOTS) break;
        const u32 c = tcnt[sl];
        toff[sl] = re; trank[sl] = rk;
        if (c >= 2) { re += c; ++rk; }
    }
    __syncthreads();
    const u64 base = bbase[b];
    const u32 ebase = (u32)base, kbase = (u32)(base >> 32);
    for (u32 i = tid; i < size; i += NT) {
        const u32 sl = slot_of(keys[b0 + i], false);
        if (tcnt[sl] >= 2) {
            const u32 fill = atomicAdd(&tfill[sl], 1u);
            const u32 p = ebase + toff[sl] + fill, r = kbase + trank[sl];
            vals2[p] = vals[b0 + i];
            if (rank2) rank2[p] = r;
            if (p % CR_CHUNK == 0) crank[p / CR_CHUNK] = r;
            if (fill == 0) first[r] = p;
        }
    }
    }
}

// kept entries to their final places: bucket base (exclusive scan over bsum) + place inside the bucket.
// The places inside a bucket are a random permutation (slot order), so a bucket's output is put in
// order in LDS and leaves in full lines.  One workgroup handles HB_EMIT consecutive buckets.
template <class V>
__global__ __launch_bounds__(HB_THREADS) void k_bucket_emit(const unsigned short* __restrict__ rec, const V* __restrict__ vals,
                                                            const BucketBounds bb, const u64* __restrict__ bbase,
                                                            const u64* __restrict__ bsum, u32 nbuckets,
                                                            V* __restrict__ vals2, u32* __restrict__ rank2,
                                                            u32* __restrict__ first, u64* __restrict__ scal, u32* __restrict__ crank) {
    constexpr u32 NT = HB_THREADS, EPT = HB_CAP / NT, FW = HB_CAP / 32;
    __shared__ u32 s_start[HB_EMIT], s_size[HB_EMIT];
    __shared__ u64 s_base[HB_EMIT], s_sum[HB_EMIT];
    __shared__ V o_tag[HB_CAP];
    __shared__ u32 s_fb2[2][FW], s_fpre[FW];   // bit p: a key's entries begin at place p (two bitmaps: the next bucket's is
                                               // cleared while this one's is read); keys below word w
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 g0 = blockIdx.x * HB_EMIT;
    if (tid < HB_EMIT) {
        const bool in = g0 + tid < nbuckets;
        s_start[tid] = in ? bb.first(g0 + tid) : 0;
        s_size[tid] = in ? bb.size(g0 + tid) : 0;
        s_base[tid] = in ? bbase[g0 + tid] : 0;
        s_sum[tid] = in ? bsum[g0 + tid] : 0;
    }
    if (tid < 2 * FW) (&s_fb2[0][0])[tid] = 0;
    __syncthreads();
    u32 flip = 0;
    for (u32 q = 0; q < HB_EMIT; ++q) {
        const u32 b0 = s_start[q], size = s_size[q];
        const u64 sum = s_sum[q];
        if (sum == 0 || size > HB_CAP) continue;   // nothing kept (or a bucket the group kernel skipped)
        u32* const s_fbits = s_fb2[flip];
        u32* const s_fnext = s_fb2[flip ^ 1];
        flip ^= 1;
        u32 r[EPT];
        V t[EPT];
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            const bool in = tid + j * NT < size;
            r[j] = in ? rec[b0 + tid + j * NT] : 0;
            t[j] = in ? vals[b0 + tid + j * NT] : V(0);
        }
#pragma unroll
        for (u32 j = 0; j < EPT; ++j) {
            if (r[j] & HB_KEPT) {
                const u32 place = r[j] & 0xFFFu;
                o_tag[place] = t[j];
                if (r[j] & HB_FIRST) atomicOr(&s_fbits[place >> 5], 1u << (place & 31));
            }
        }
        __syncthreads();
        if (tid < 64) {   // keys in front of every word of the bitmap (96 words: two per lane of one wave)
            static_assert(FW <= 128, "two bitmap words per lane");
            const u32 c0 = lane < FW ? (u32)__popc(s_fbits[lane]) : 0u, c1 = 64 + lane < FW ? (u32)__popc(s_fbits[64 + lane]) : 0u;
            const u32 i0 = wave_scan_add(c0), i1 = wave_scan_add(c1);
            const u32 tot0 = __shfl(i0, 63);
            if (lane < FW) s_fpre[lane] = i0 - c0;
            if (64 + lane < FW) s_fpre[64 + lane] = tot0 + i1 - c1;
        }
        __syncthreads();
        const u64 base = s_base[q];
        const u32 ebase = (u32)base, kbase = (u32)(base >> 32), ke = (u32)sum;
        for (u32 i = tid; i < ke; i += NT) {
            vals2[ebase + i] = o_tag[i];
            const u32 w = s_fbits[i >> 5], below = s_fpre[i >> 5] + (u32)__popc(w & ((1u << (i & 31)) - 1u));   // keys that begin below place i
            const bool head = (w >> (i & 31)) & 1u;
            const u32 rk = kbase + below - (head ? 0u : 1u);   // rank of the key place i belongs to
            if (head) first[kbase + below] = ebase + i;
            if (rank2) rank2[ebase + i] = rk;
            if ((ebase + i) % CR_CHUNK == 0) crank[(ebase + i) / CR_CHUNK] = rk;
        }
        if (tid < FW) s_fnext[tid] = 0;   // (the bitmap of the bucket after this one: last read two barriers ago)
        __syncthreads();
    }
    if (g0 + HB_EMIT >= nbuckets && tid == 0) {
        const u64 tot = bbase[nbuckets - 1] + bsum[nbuckets - 1];
        scal[6] = (u32)tot;           // kept entries
        scal[2] = (u32)(tot >> 32);   // kept distinct keys (U)
        first[(u32)(tot >> 32)] = (u32)tot;   // sentinel
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
                c.first[hi] = lo; // sentinel: one past the last kept entry
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

// ---- (block, key) groups straight from the keys --------------------------------------------------
// After the grouping by key the entries of a key are contiguous (first[]), and after the source
// reordering nearly all of a key's holders sit in one block: the (block, key) groups — the words of
// the block lists — can be read off key by key, and only the *groups* (an order of magnitude fewer than
// the entries) have to be brought into block order.  A workgroup stages a chunk of entries in LDS as
// new source indices; one thread per key walks its holders block by block (ascending), building the
// 128-bit membership mask of each group.  A key with c holders has at most c groups and c/5 groups
// with more than INLINE_MAX members, so its records can be parked without knowing any offsets yet: the
// first group of key r (for most keys the only one) in per-key arrays at index r, further groups at the
// key's own entry positions (block and posting word at first[r] + j, masks at first[r]/4 + j).
// gsum[r] = groups | masks << 32; after the scan k_move_groups packs the records in rank order.
// Keys with more than KG_MAXC holders do not fit the staging: they are listed for k_key_groups_huge; only
// when that cannot take them (more than KG_HUGE_NB blocks, or more than KG_HUGE_CAP such keys) *ovf is
// raised and the build takes the sort-by-block path instead.
constexpr u32 KG_CHUNK = 4096, KG_MAXC = 2048, KG_THREADS = 256, KG_COOP = 64, KG_COOP_MIN = 16;
constexpr u32 KG_HUGE_CAP = 4096, KG_HUGE_NB = 2048;   // keys with more than KG_MAXC holders per build / blocks their LDS table holds

template <class V, bool W>
__global__ __launch_bounds__(KG_THREADS) void k_key_groups(const V* __restrict__ vals, const u32* __restrict__ rank,
                                                           const u32* __restrict__ first, const u32* __restrict__ newidx,
                                                           u32 m, u32 n_keys, u64* __restrict__ gsum,
                                                           u32* __restrict__ blk0, u32* __restrict__ info0,
                                                           uint4* __restrict__ mask0, u32* __restrict__ tmp_blk,
                                                           u32* __restrict__ tmp_info, uint4* __restrict__ tmp_mask,
                                                           u32* __restrict__ wkey, u32* __restrict__ ovf, const u32 coop,
                                                           u32* __restrict__ huge_list) {
    __shared__ u32 s_idx[KG_CHUNK + KG_MAXC + 4];
    ST_BEGIN();
    const u32 E0 = blockIdx.x * KG_CHUNK, E1 = min(m, E0 + KG_CHUNK);
    // the keys that start inside [E0, E1)
    static_assert(KG_CHUNK == CR_CHUNK, "crank[] holds the rank of every KG_CHUNK-th entry");
    u32 r_lo = rank[blockIdx.x];   // (crank: the rank of entry E0)
    if (first[r_lo] != E0) ++r_lo;
    u32 r_hi = n_keys;
    if (E1 < m) { r_hi = rank[blockIdx.x + 1]; if (first[r_hi] != E1) ++r_hi; }
    if (r_lo >= r_hi) return;
    const u32 Eend = min(first[r_hi], E0 + KG_CHUNK + KG_MAXC);
    ST_T(16);
    // (all of a thread's tag loads and look-ups issued together instead of this loop — the round-3 cure for k_fkeys — made
    //  this kernel SLOWER: C2 build 1.13 -> 1.18 ms, 100 000 genomes 18.4 -> 19.9; six workgroups per CU hide the loop's waits)
    for (u32 i = threadIdx.x; E0 + i < Eend; i += KG_THREADS) s_idx[i] = newidx[src_of_tag(tag_of(vals[E0 + i]))];
    __syncthreads();
    ST_T(17);
    // keys left to a whole wave (below): rank - r_lo, and (filled by the thread that lists the key, which has both words in
    // registers) where its entries start in the staging and how many there are — the wave-per-key pass then needs no memory
    // load per key (first[r], first[r + 1] were a round trip at the head of every key, behind the previous key's stores)
    constexpr u32 KG_NBIG = (KG_CHUNK + KG_MAXC) / KG_COOP_MIN + 8;
    __shared__ unsigned short s_big[KG_NBIG], s_bigf[KG_NBIG], s_bigc[KG_NBIG];
    __shared__ u32 s_nbig;
    if (threadIdx.x == 0) s_nbig = 0;
    // one group of key r: block `cur`, members lo | hi.  The key's first group goes to the per-key arrays, the
    // others are parked at the key's entry positions (every lane of a cooperating wave counts, one stores).
    auto emit = [&](const u32 r, const u32 fa, const u32 cur, const unsigned long long lo, const unsigned long long hi,
                    u32& groups, u32& bigs, u32& parked, const bool store) {
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
            const uint4 mask = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
            if (groups == 0) { if (store) mask0[r] = mask; inf = BIG; }
            else { const u32 slot = fa / 4 + parked; if (store) tmp_mask[slot] = mask; inf = BIG | slot; ++parked; }
            ++bigs;
        }
        if (store) {
            if (groups == 0) { blk0[r] = cur; info0[r] = inf; }
            else { tmp_blk[fa + groups] = cur; tmp_info[fa + groups] = inf; }
        }
        ++groups;
    };
    __syncthreads();
    for (u32 r = r_lo + threadIdx.x; r < r_hi; r += KG_THREADS) {
        const u32 fa = first[r], c = first[r + 1] - fa, f0 = fa - E0;
        if (c > KG_MAXC) {   // does not fit the staging: left to k_key_groups_huge (ovf[1] counts them)
            const u32 q = huge_list ? atomicAdd(&ovf[1], 1u) : KG_HUGE_CAP;
            if (q < KG_HUGE_CAP) huge_list[q] = r; else ovf[0] = 1;
            continue;
        }
        if (c > coop) {   // many holders: a whole wave walks this key (below)
            const u32 slot = atomicAdd(&s_nbig, 1u);
            s_big[slot] = (unsigned short)(r - r_lo);
            s_bigf[slot] = (unsigned short)f0;
            s_bigc[slot] = (unsigned short)c;
            continue;
        }
        if (W && c) wkey[r] = weight_of(vals[fa]);
        // the block of the first holder first (for most keys the only one: a single walk), then the others ascending
        const u32 b0 = c ? s_idx[f0] / TB : ~0u;
        u32 cur = b0;
        u32 groups = 0, bigs = 0, parked = 0;
        while (cur != ~0u) {
            u32 nxt = ~0u;
            unsigned long long lo = 0, hi = 0;
            const u32 floor_b = groups == 0 ? 0u : cur + 1;   // (first walk: the smallest other block; later: the next one up)
            for (u32 i = 0; i < c; i += 4) {   // four holders per step: the LDS reads of a step are independent
                u32 t4[4];
#pragma unroll
                for (u32 q = 0; q < 4; ++q) t4[q] = s_idx[f0 + i + q];   // (the staging area has 4 words of slack)
#pragma unroll
                for (u32 q = 0; q < 4; ++q) {
                    if (i + q >= c) break;
                    const u32 t = t4[q], b = t / TB;
                    if (b == cur) {
                        const u32 l = t % TB;
                        if (l < 64) lo |= 1ull << l; else hi |= 1ull << (l - 64);
                    } else if (b >= floor_b && b != b0 && b < nxt) nxt = b;
                }
            }
            emit(r, fa, cur, lo, hi, groups, bigs, parked, true);
            cur = nxt;
        }
        gsum[r] = (u64)groups | ((u64)bigs << 32);
    }
    ST_T(18);
    __syncthreads();
    ST_T(19);
    // keys with many holders, one wave each.  Round 3: two strided passes over the holders whatever the number of blocks —
    // the smallest block (wave minimum), then every holder ORs its bit into a table of 32 blocks x 128 bits in LDS (related
    // sources are neighbours after the reordering: a key's blocks are a short range); the 32 table rows are then read by 32
    // lanes at once, one group each.  (Before: one pass over ALL holders per block of the key, every pass ending in a
    // shuffle reduction of mask and next block — 100 000 genomes, 152 holders per key in up to 8 blocks: 4.1 ms.)  A key whose
    // blocks span more than 32 takes the block-by-block walk as before.
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ u32 s_wtab[KG_THREADS / 64][32 * 4];
    u32* const tab = s_wtab[wv];
    for (u32 q = wv; q < s_nbig; q += KG_THREADS / 64) {
        const u32 r = r_lo + s_big[q];
        const u32 f0 = s_bigf[q], c = s_bigc[q], fa = E0 + f0;
        if (W && lane == 0) wkey[r] = weight_of(vals[fa]);
        u32 bmin = ~0u, bmax = 0;
        for (u32 i = lane; i < c; i += 64) { const u32 b = s_idx[f0 + i] / TB; bmin = min(bmin, b); bmax = max(bmax, b); }
        bmin = wave_all_min(bmin, (int)lane);
        bmax = wave_all_max(bmax, (int)lane);
        if (bmax - bmin < 32) {
            tab[lane] = 0; tab[64 + lane] = 0;
            __builtin_amdgcn_wave_barrier();
            for (u32 i = lane; i < c; i += 64) {
                const u32 t = s_idx[f0 + i], l = t % TB;
                atomicOr(&tab[(t / TB - bmin) * 4 + (l >> 5)], 1u << (l & 31));
            }
            __builtin_amdgcn_wave_barrier();
            // lane j < 32 owns block bmin + j
            unsigned long long lo = 0, hi = 0;
            if (lane < 32) {
                lo = (unsigned long long)tab[lane * 4] | ((unsigned long long)tab[lane * 4 + 1] << 32);
                hi = (unsigned long long)tab[lane * 4 + 2] | ((unsigned long long)tab[lane * 4 + 3] << 32);
            }
            const u32 cnt = __popcll(lo) + __popcll(hi);
            const unsigned long long have = __ballot(cnt != 0), bigm = __ballot(cnt > INLINE_MAX);
            const unsigned long long below = (1ull << lane) - 1ull;
            const u32 gi = (u32)__popcll(have & below);                       // my group among the key's groups (blocks ascending)
            const u32 g0big = (bigm >> (__ffsll((long long)have) - 1)) & 1ull ? 1u : 0u;   // the key's first group has a mask of its own
            if (cnt) {
                u32 inf;
                if (cnt <= INLINE_MAX) {
                    inf = (cnt - 1) << 29;
                    unsigned long long a = lo, bq = hi;
                    for (u32 j = 0; j < cnt; ++j) {
                        u32 id;
                        if (a) { id = __ffsll((long long)a) - 1; a &= a - 1; }
                        else { id = 64 + __ffsll((long long)bq) - 1; bq &= bq - 1; }
                        inf |= id << (7 * j);
                    }
                } else {
                    const uint4 mask = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
                    if (gi == 0) { mask0[r] = mask; inf = BIG; }
                    else { const u32 slot = fa / 4 + ((u32)__popcll(bigm & below) - g0big); tmp_mask[slot] = mask; inf = BIG | slot; }
                }
                if (gi == 0) { blk0[r] = bmin + lane; info0[r] = inf; }
                else { tmp_blk[fa + gi] = bmin + lane; tmp_info[fa + gi] = inf; }
            }
            if (lane == 0) gsum[r] = (u64)__popcll(have) | ((u64)__popcll(bigm) << 32);
            __builtin_amdgcn_wave_barrier();   // (the table is zeroed again for the next key)
            continue;
        }
        const u32 b0 = s_idx[f0] / TB;
        u32 cur = b0;
        u32 groups = 0, bigs = 0, parked = 0;
        while (cur != ~0u) {
            u32 nxt = ~0u;
            unsigned long long lo = 0, hi = 0;
            const u32 floor_b = groups == 0 ? 0u : cur + 1;
            for (u32 i = lane; i < c; i += 64) {
                const u32 t = s_idx[f0 + i], b = t / TB;
                if (b == cur) {
                    const u32 l = t % TB;
                    if (l < 64) lo |= 1ull << l; else hi |= 1ull << (l - 64);
                } else if (b >= floor_b && b != b0 && b < nxt) nxt = b;
            }
            lo = wave_all_or(lo, (int)lane);
            hi = wave_all_or(hi, (int)lane);
            nxt = wave_all_min(nxt, (int)lane);
            emit(r, fa, cur, lo, hi, groups, bigs, parked, lane == 0);
            cur = nxt;
        }
        if (lane == 0) gsum[r] = (u64)groups | ((u64)bigs << 32);
    }
    ST_T(20);
    ST_END(16);
}
// crank[] from a per-entry rank array (the sort path and the postings input produce one) / the per-entry array from
// first[] (the sort-by-block fallback after a grouping that only wrote crank[])
__global__ void k_crank_from_rank(const u32* __restrict__ rank, u32* __restrict__ crank, const u32 m) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if ((u64)c * CR_CHUNK < m) crank[c] = rank[(size_t)c * CR_CHUNK];
}
__global__ void k_rank_fill(const u32* __restrict__ first, const u32 n_keys, u32* __restrict__ rank) {
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (u32 r = wave; r < n_keys; r += nwaves)
        for (u32 e = first[r] + lane; e < first[r + 1]; e += 64) rank[e] = r;
}
// Keys with more than KG_MAXC holders (conserved k-mers of a large same-species collection), one workgroup
// each: the holders are read from global memory once and OR-ed into an LDS table of 128-bit masks, one per
// block (up to KG_HUGE_NB blocks); the non-empty blocks, ascending, are the key's groups.  Same parked
// record format as k_key_groups.
template <class V, bool W>
__global__ __launch_bounds__(256) void k_key_groups_huge(const V* __restrict__ vals, const u32* __restrict__ first,
                                                         const u32* __restrict__ newidx, const u32 nb,
                                                         u64* __restrict__ gsum, u32* __restrict__ blk0,
                                                         u32* __restrict__ info0, uint4* __restrict__ mask0,
                                                         u32* __restrict__ tmp_blk, u32* __restrict__ tmp_info,
                                                         uint4* __restrict__ tmp_mask, u32* __restrict__ wkey,
                                                         const u32* __restrict__ ovf, const u32* __restrict__ huge_list) {
    __shared__ u32 tab[KG_HUGE_NB * 4];
    __shared__ u32 s_part[4], s_g0big;
    const u32 n_huge = min(ovf[1], KG_HUGE_CAP);
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (u32 q = blockIdx.x; q < n_huge; q += gridDim.x) {   // (a few workgroups walk the list: it is empty for most inputs)
    const u32 r = huge_list[q];
    const u32 fa = first[r], c = first[r + 1] - fa;
    __syncthreads();   // (the table of the previous key is no longer read)
    for (u32 i = tid; i < nb * 4; i += 256) tab[i] = 0;
    if (tid == 0) s_g0big = 0;
    __syncthreads();
    for (u32 i = tid; i < c; i += 256) {
        const u32 t = newidx[src_of_tag(tag_of(vals[fa + i]))];
        atomicOr(&tab[(t / TB) * 4 + ((t % TB) >> 5)], 1u << (t & 31));
    }
    if (W && tid == 0) wkey[r] = weight_of(vals[fa]);
    __syncthreads();
    // every thread owns a contiguous range of blocks (ascending order is kept); groups | bigs << 16, scanned
    const u32 per = (nb + 255) / 256, b_lo = tid * per, b_hi = min(nb, b_lo + per);
    u32 mine = 0;
    for (u32 b = b_lo; b < b_hi; ++b) {
        const u32 cnt = __popc(tab[b * 4]) + __popc(tab[b * 4 + 1]) + __popc(tab[b * 4 + 2]) + __popc(tab[b * 4 + 3]);
        if (cnt) mine += 1u + ((cnt > INLINE_MAX) << 16);
    }
    u32 inc = mine;
    inc = wave_scan_add(inc);
    if (lane == 63) s_part[wv] = inc;
    __syncthreads();
    u32 run = inc - mine;
    for (u32 w = 0; w < wv; ++w) run += s_part[w];
    // is the key's first group one with a mask?  (its mask goes to the per-key array, not to a parked slot)
    if ((run & 0xFFFFu) == 0 && (mine & 0xFFFFu)) {
        for (u32 b = b_lo; b < b_hi; ++b) {
            const u32 cnt = __popc(tab[b * 4]) + __popc(tab[b * 4 + 1]) + __popc(tab[b * 4 + 2]) + __popc(tab[b * 4 + 3]);
            if (cnt) { s_g0big = cnt > INLINE_MAX; break; }
        }
    }
    __syncthreads();
    const u32 g0big = s_g0big;
    u32 j = run & 0xFFFFu, bigs = run >> 16;
    for (u32 b = b_lo; b < b_hi; ++b) {
        const u32 m0 = tab[b * 4], m1 = tab[b * 4 + 1], m2 = tab[b * 4 + 2], m3 = tab[b * 4 + 3];
        const u32 cnt = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
        if (!cnt) continue;
        u32 inf;
        if (cnt <= INLINE_MAX) {
            inf = (cnt - 1) << 29;
            unsigned long long a = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
            unsigned long long bq = (unsigned long long)m2 | ((unsigned long long)m3 << 32);
            for (u32 q = 0; q < cnt; ++q) {
                u32 id;
                if (a) { id = __ffsll((long long)a) - 1; a &= a - 1; }
                else { id = 64 + __ffsll((long long)bq) - 1; bq &= bq - 1; }
                inf |= id << (7 * q);
            }
        } else {
            const uint4 mask = make_uint4(m0, m1, m2, m3);
            if (j == 0) { mask0[r] = mask; inf = BIG; }
            else { const u32 slot = fa / 4 + (bigs - g0big); tmp_mask[slot] = mask; inf = BIG | slot; }
            ++bigs;
        }
        if (j == 0) { blk0[r] = b; info0[r] = inf; }
        else { tmp_blk[fa + j] = b; tmp_info[fa + j] = inf; }
        ++j;
    }
    if (tid == 255) gsum[r] = (u64)j | ((u64)bigs << 32);   // (the last thread's running totals are the key's totals)
    }
}
// the parked records to their places in rank order (goff = exclusive scan of gsum): block, rank << 32 |
// posting word, rank; masks to their final index.  With `work` set (at most KG_WORK blocks) the kernel also
// sums what the join's schedule needs — per block the pair updates of its diagonal tile, C(holders, 2) per
// group, and in slot nb the holders of all groups — in LDS per workgroup (a few hundred persistent ones).
constexpr u32 KG_WORK = 2048;
__global__ __launch_bounds__(256) void k_move_groups(const u64* __restrict__ gsum, const u64* __restrict__ goff,
                                                     const u32* __restrict__ first, const u32* __restrict__ blk0,
                                                     const u32* __restrict__ info0, const uint4* __restrict__ mask0,
                                                     const u32* __restrict__ tmp_blk, const u32* __restrict__ tmp_info,
                                                     const uint4* __restrict__ tmp_mask, u32* __restrict__ rec_blk,
                                                     u64* __restrict__ rec_val, u32* __restrict__ rec_rank,
                                                     uint4* __restrict__ bigmask, u32 n_keys,
                                                     unsigned long long* __restrict__ work, u32 nb, const u32* __restrict__ ovf,
                                                     const Rider rider) {
    __shared__ unsigned long long s_work[KG_WORK + 1];
    rider_run(rider);
    if (*ovf) return;   // (queued before the host knew: a key with too many holders — the build sorts the entries by block instead)
    if (work) {
        for (u32 i = threadIdx.x; i <= nb; i += blockDim.x) s_work[i] = 0;
        __syncthreads();
    }
    unsigned long long holders = 0;
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < n_keys; r += gridDim.x * blockDim.x) {
        const u32 k = (u32)gsum[r];
        if (!k) continue;
        const u64 base = goff[r];
        u32 o = (u32)(base >> 32);
        u32 g = (u32)base;
        u32 fa = 0;
        for (u32 j = 0; j < k; ++j, ++g) {
            u32 inf, blk, cnt;
            if (j == 0) { inf = info0[r]; blk = blk0[r]; }   // the key's first group: per-key arrays
            else {
                if (j == 1) fa = first[r];
                inf = tmp_info[fa + j]; blk = tmp_blk[fa + j];
            }
            if (inf >= BIG) {
                const uint4 mk = j == 0 ? mask0[r] : tmp_mask[inf & ~BIG];
                cnt = __popc(mk.x) + __popc(mk.y) + __popc(mk.z) + __popc(mk.w);
                bigmask[o] = mk;
                inf = BIG | o;
                ++o;
            } else cnt = (inf >> 29) + 1;
            rec_blk[g] = blk;
            rec_val[g] = ((u64)r << 32) | inf;
            rec_rank[g] = r;
            if (work) {
                holders += cnt;
                if (cnt > 1) atomicAdd(&s_work[blk], (unsigned long long)cnt * (cnt - 1) / 2);
            }
        }
    }
    if (work) {
        for (int o = 32; o > 0; o >>= 1) holders += __shfl_down(holders, o);
        if ((threadIdx.x & 63) == 0 && holders) atomicAdd(&s_work[nb], holders);
        __syncthreads();
        for (u32 i = threadIdx.x; i <= nb; i += blockDim.x)
            if (s_work[i]) atomicAdd(&work[i], s_work[i]);
    }
}
// totals of the group scan: scal[1] = list words (groups), scal[7] = masks
// Exclusive prefix sums over words that pack two 32-bit counters (groups | masks << 32: neither total reaches 2^32), in two
// dispatches — the sums of 2 048-word tiles, then every tile scanned behind the sum of the tile sums in front of it (a
// few hundred words out of L2) — instead of the library scan's initialisation + look-back passes and a totals kernel
// (C2: 30 us for 1.3 M keys).  The last tile leaves the totals in scal[1] (low counter) and scal[7] (high counter).
constexpr u32 PS_THREADS = 256, PS_PER = 8, PS_TILE = PS_THREADS * PS_PER;
__global__ __launch_bounds__(PS_THREADS) void k_pair_tile_sums(const u64* __restrict__ in, const u32 n, u64* __restrict__ tsum) {
    __shared__ u32 s_lo[PS_THREADS / 64], s_hi[PS_THREADS / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, base = blockIdx.x * PS_TILE;
    u32 lo = 0, hi = 0;
#pragma unroll
    for (u32 k = 0; k < PS_PER; ++k) {
        const u32 i = base + k * PS_THREADS + tid;
        const u64 v = i < n ? in[i] : 0;
        lo += (u32)v; hi += (u32)(v >> 32);
    }
    lo = wave_scan_add(lo); hi = wave_scan_add(hi);
    if (lane == 63) { s_lo[wv] = lo; s_hi[wv] = hi; }
    __syncthreads();
    if (tid == 0) {
        u32 a = 0, b = 0;
        for (u32 w = 0; w < PS_THREADS / 64; ++w) { a += s_lo[w]; b += s_hi[w]; }
        tsum[blockIdx.x] = (u64)a | ((u64)b << 32);
    }
}
__global__ __launch_bounds__(PS_THREADS) void k_pair_scan_tiles(const u64* __restrict__ in, u64* __restrict__ out, const u32 n,
                                                                const u64* __restrict__ tsum, u64* __restrict__ scal) {
    __shared__ u32 s_lo[PS_THREADS / 64], s_hi[PS_THREADS / 64], s_olo[PS_THREADS / 64], s_ohi[PS_THREADS / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, t = blockIdx.x, base = t * PS_TILE;
    // the words of this thread: PS_PER consecutive ones (a wave reads 4 KB in a row)
    u64 v[PS_PER];
    u32 lo = 0, hi = 0;
#pragma unroll
    for (u32 k = 0; k < PS_PER; ++k) {
        const u32 i = base + tid * PS_PER + k;
        v[k] = i < n ? in[i] : 0;
        lo += (u32)v[k]; hi += (u32)(v[k] >> 32);
    }
    // the tiles in front of this one
    u32 olo = 0, ohi = 0;
    for (u32 j = tid; j < t; j += PS_THREADS) { const u64 x = tsum[j]; olo += (u32)x; ohi += (u32)(x >> 32); }
    olo = wave_scan_add(olo); ohi = wave_scan_add(ohi);
    const u32 ilo = wave_scan_add(lo), ihi = wave_scan_add(hi);
    if (lane == 63) { s_lo[wv] = ilo; s_hi[wv] = ihi; s_olo[wv] = olo; s_ohi[wv] = ohi; }
    __syncthreads();
    u32 rlo = ilo - lo, rhi = ihi - hi;   // exclusive, inside the wave
    for (u32 w = 0; w < PS_THREADS / 64; ++w) {
        rlo += s_olo[w]; rhi += s_ohi[w];
        if (w < wv) { rlo += s_lo[w]; rhi += s_hi[w]; }
    }
#pragma unroll
    for (u32 k = 0; k < PS_PER; ++k) {
        const u32 i = base + tid * PS_PER + k;
        if (i < n) out[i] = (u64)rlo | ((u64)rhi << 32);
        rlo += (u32)v[k]; rhi += (u32)(v[k] >> 32);
        if (i == n - 1 && scal) { scal[1] = rlo; scal[7] = rhi; }   // (k_group_totals: list words, masks)
    }
}
__global__ void k_group_totals(const u64* __restrict__ gsum, const u64* __restrict__ goff, u64* __restrict__ scal, u32 n_keys) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const u64 t = goff[n_keys - 1] + gsum[n_keys - 1];
        scal[1] = (u32)t;
        scal[7] = t >> 32;
    }
}
// first group of every block in the block-sorted group list
__global__ void k_blk_raw_groups(const u32* __restrict__ sblk, u32 n_groups, u32* __restrict__ blk_raw, u32 nb) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    u32 lo = 0, hi = n_groups;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (sblk[mid] < b) lo = mid + 1; else hi = mid;
    }
    blk_raw[b] = lo;
}
// k_blk_raw_groups + k_blk_pos in one workgroup (up to 1024 x 8 blocks): bisect, then the padded starts
constexpr u32 BRP_MAX = 8192;
__global__ __launch_bounds__(1024) void k_blk_raw_pos(const u32* __restrict__ sblk, u32 n_groups, u32* __restrict__ blk_raw,
                                                      u32* __restrict__ blk_pos, u64* __restrict__ scal, u32 nb) {
    __shared__ u32 s_raw[BRP_MAX + 1];
    for (u32 b = threadIdx.x; b <= nb; b += blockDim.x) {
        u32 lo = 0, hi = n_groups;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (sblk[mid] < b) lo = mid + 1; else hi = mid;
        }
        s_raw[b] = lo;
        blk_raw[b] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 pos = 0;
        for (u32 b = 0; b < nb; ++b) {
            blk_pos[b] = pos;
            pos = ((pos + (s_raw[b + 1] - s_raw[b]) + 3u) & ~3u) + WIN;
        }
        blk_pos[nb] = pos;
        scal[3] = pos;
    }
}
// groups (sorted by block, ranks ascending inside a block) to the padded lists
template <bool W>
__global__ void k_place_groups(const u32* __restrict__ sblk, const u64* __restrict__ sval, const u32* __restrict__ blk_raw,
                               const u32* __restrict__ blk_pos, const u32* __restrict__ wkey, u32* __restrict__ brk,
                               u32* __restrict__ info, u32* __restrict__ bw, u32 n_groups) {
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_groups) return;
    const u32 b = sblk[d];
    const u64 v = sval[d];
    const u32 dst = blk_pos[b] + (d - blk_raw[b]);
    brk[dst] = (u32)(v >> 32);
    info[dst] = (u32)v;
    if (W) bw[dst] = wkey[(u32)(v >> 32)];
}

// ---- the groups to their blocks without a library sort (up to 256 blocks) ------------------------------------------
// The group records leave k_move_groups in rank order; the block lists want them by block, ranks ascending: a stable
// split on the block id.  The library's radix sort took six dispatches for it (fills, histogram, scan, one pass) and
// then two more kernels found the block bounds and placed the sorted records; here a chunk of 2 048 consecutive
// records is counted per block (k_ms_hist), one workgroup turns the counts into every chunk's first place inside every
// block and into the block tables (k_ms_scan: what k_blk_raw_pos searched for), and k_ms_place ranks the records of its
// chunk — equal blocks inside a wave by eight ballots, waves and rounds through a small LDS table — and writes ranks
// and posting words straight into the padded lists.
constexpr u32 MS_CHUNK = 2048, MS_THREADS = 256, MS_ROUNDS = MS_CHUNK / MS_THREADS;
constexpr u32 MS_MAXB = 1024;         // blocks the split takes (two instantiations: tables of 256 or 1 024 blocks)
constexpr u32 PM_BIG = 0xE0000000u;   // (= BIG: posting word with a mask index)
template <u32 MB>
__global__ __launch_bounds__(MS_THREADS) void k_ms_hist(const u32* __restrict__ rec_blk, u64* __restrict__ scal,
                                                         u32* __restrict__ hist) {
    __shared__ u32 s_h[MB];
    const u32 n = (u32)scal[1];   // groups (k_group_totals)
    const u32 g0 = blockIdx.x * MS_CHUNK;
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<u32*>(scal + 15)[0] = 0;   // k_ms_scan's "workgroups done"
    if (g0 >= n) return;
    for (u32 i = threadIdx.x; i < MB; i += MS_THREADS) s_h[i] = 0;
    __syncthreads();
    for (u32 k = 0; k < MS_ROUNDS; ++k) {
        const u32 g = g0 + k * MS_THREADS + threadIdx.x;
        if (g < n) atomicAdd(&s_h[rec_blk[g] & (MB - 1)], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < MB; i += MS_THREADS) hist[(size_t)blockIdx.x * MB + i] = s_h[i];
}
// hist[c][b] -> groups of block b in the chunks before c; blk_raw / blk_pos as k_blk_raw_pos leaves them.  One
// workgroup per block: every thread takes a run of consecutive chunks of the block's column, 16 loads in flight at a
// time (sum, then — after the workgroup's scan of the thread sums — running sums written back); the workgroup that
// finishes last (a counter in the scalar block, zeroed by k_ms_hist) lays out the block tables.
constexpr u32 MS_PER = 16;
__global__ __launch_bounds__(256) void k_ms_scan(u32* __restrict__ hist, const u32 stride, u64* __restrict__ scal,
                                                 u32* __restrict__ tot, u32* __restrict__ blk_raw, u32* __restrict__ blk_pos,
                                                 const u32 nb) {
    __shared__ u32 s_w[4], s_last;
    const u32 n = (u32)scal[1];
    const u32 chunks = (n + MS_CHUNK - 1) / MS_CHUNK, per = (chunks + 255u) / 256u;
    const u32 b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 c0 = tid * per, c1 = min(chunks, c0 + per);
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
    for (u32 i = tid; i < nb; i += 256) s_t[i] = __hip_atomic_load(&tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (one round trip, not one per block)
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
}
template <bool W, u32 MB>
__global__ __launch_bounds__(MS_THREADS) void k_ms_place(const u32* __restrict__ rec_blk, const u64* __restrict__ rec_val,
                                                          const u64* __restrict__ scal, const u32* __restrict__ base,
                                                          const u32* __restrict__ blk_pos, const u32 nb, const u32* __restrict__ wkey,
                                                          u32* __restrict__ brk, u32* __restrict__ info, u32* __restrict__ bw,
                                                          const uint4* __restrict__ bigmask, uint4* __restrict__ pmask,
                                                          const u32* __restrict__ blk_raw, const u32 padv) {
    constexpr u32 NWV = MS_THREADS / 64, NSL = MS_ROUNDS * NWV, BITS = MB == 256 ? 8u : 10u;
    static_assert(MB == 256 || MB == 1024, "two table sizes");
    __shared__ unsigned short s_cnt[NSL][MB];   // records of block b in (round, wave) slot (<= 64); then: records before the slot (< 2 048)
    __shared__ u32 s_dst[MB];                   // first place of this chunk's records of block b in the padded list
    const u32 n = (u32)scal[1];
    const u32 g0 = blockIdx.x * MS_CHUNK;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (k_pad's work — the padding words behind every block's list and behind the last block — rides along: a dispatch less)
    for (u32 b = blockIdx.x; b <= nb; b += gridDim.x) {
        const u32 lo = b < nb ? blk_pos[b] + (blk_raw[b + 1] - blk_raw[b]) : blk_pos[nb];
        const u32 hi = b < nb ? blk_pos[b + 1] : blk_pos[nb] + 4u * WIN;
        for (u32 i = lo + tid; i < hi; i += MS_THREADS) brk[i] = padv;
    }
    if (g0 >= n) return;
    for (u32 i = tid; i < NSL * MB / 2; i += MS_THREADS) reinterpret_cast<u32*>(&s_cnt[0][0])[i] = 0;
    for (u32 i = tid; i < MB; i += MS_THREADS) s_dst[i] = (i < nb ? blk_pos[i] : 0u) + base[(size_t)blockIdx.x * MB + i];
    u32 blk[MS_ROUNDS], rk[MS_ROUNDS];
    u64 val[MS_ROUNDS];
#pragma unroll
    for (u32 k = 0; k < MS_ROUNDS; ++k) {
        const u32 g = g0 + k * MS_THREADS + tid;
        blk[k] = g < n ? (rec_blk[g] & (MB - 1)) : ~0u;
        val[k] = g < n ? rec_val[g] : 0;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < MS_ROUNDS; ++k) {
        // lanes of this wave with the same block: one ballot per bit of the block id
        unsigned long long m = __ballot(blk[k] != ~0u);
#pragma unroll
        for (u32 bit = 0; bit < BITS; ++bit) {
            const unsigned long long bal = __ballot((blk[k] >> bit) & 1u);
            m &= ((blk[k] >> bit) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        rk[k] = (u32)__popcll(below);
        if (blk[k] != ~0u && below == 0) s_cnt[k * NWV + wv][blk[k]] = (unsigned short)__popcll(m);   // (the first lane of every block present)
    }
    __syncthreads();
    for (u32 bq = tid; bq < MB; bq += MS_THREADS) {   // per block: records in the slots before each (round, wave) slot
        u32 run = 0;
        for (u32 sl = 0; sl < NSL; ++sl) {
            const u32 c = s_cnt[sl][bq];
            s_cnt[sl][bq] = (unsigned short)run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < MS_ROUNDS; ++k) {
        if (blk[k] == ~0u) continue;
        const u32 dst = s_dst[blk[k]] + s_cnt[k * NWV + wv][blk[k]] + rk[k];
        brk[dst] = (u32)(val[k] >> 32);
        info[dst] = (u32)val[k];
        if (W) bw[dst] = wkey[(u32)(val[k] >> 32)];
        if (pmask) {   // the word's membership mask at its list position (inline ids expanded): the join reads it in place
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
    }
}

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

// Output "iterator" of the inclusive scan over the head flags: the store of element e records, for a head,
// where its (block, key) group starts and the group's rank — the prefix sums themselves are never written.
// The last element leaves scal[1] = Ktot (distinct (block, key) groups) and estart[Ktot] = n.
template <class V>
struct HeadScatterIt {
    using iterator_category = std::random_access_iterator_tag;
    using value_type = u32;
    using difference_type = std::ptrdiff_t;
    using pointer = void;
    struct Ctx {
        HeadFn<V> head;
        u32* estart;
        u32* grank;
        u64* scal;
        u64 n;
    };
    struct Ref {
        Ctx c;
        u64 e;
        __device__ const Ref& operator=(const u32 cur) const {
            if (c.head(e)) {
                c.estart[cur - 1] = (u32)e;
                c.grank[cur - 1] = c.head.rk[e];
            }
            if (e == c.n - 1) {
                c.scal[1] = cur;
                c.estart[cur] = (u32)c.n;
            }
            return *this;
        }
    };
    using reference = Ref;
    Ctx c;
    u64 base;
    __host__ __device__ HeadScatterIt operator+(const std::ptrdiff_t d) const { return HeadScatterIt{c, base + (u64)d}; }
    __host__ __device__ HeadScatterIt& operator+=(const std::ptrdiff_t d) { base += (u64)d; return *this; }
    __device__ Ref operator[](const std::ptrdiff_t i) const { return Ref{c, base + (u64)i}; }
    __device__ Ref operator*() const { return Ref{c, base}; }
};

// raw (unpadded) first distinct-key index of each block: entries are sorted by block, so the
// first entry of block b is found by bisection over the tags.
template <class V>
__global__ void k_blk_raw(const V* __restrict__ vals, const u32* __restrict__ estart, const u64* __restrict__ scal,
                          u32* __restrict__ blk_raw, u32 nb, u64 n) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    u64 lo = 0, hi = n;
    while (lo < hi) {
        u64 mid = lo + ((hi - lo) >> 1);
        if ((tag_of(vals[mid]) >> 8) < b) lo = mid + 1; else hi = mid;
    }
    // the block's first group: the first one that starts at or after the block's first entry
    u32 glo = 0, ghi = (u32)scal[1];
    while (glo < ghi) {
        const u32 mid = glo + ((ghi - glo) >> 1);
        if (estart[mid] < lo) glo = mid + 1; else ghi = mid;
    }
    blk_raw[b] = glo;
}

// Padded layout of the block lists: every list starts at a multiple of 4 entries and is
// followed by >= WIN pad entries (rank PAD = +inf), so that any 16-byte-aligned window of
// WIN entries that starts inside a list is sorted and never runs into the next list.
// padded start of every block list: list b takes its words rounded up to 4, + one window of pads — a prefix sum (every start
// is a multiple of 4).  One workgroup of 1 024 threads (launched as <<<1, 64>>> by older call sites: any block size works),
// every thread a contiguous run of blocks; one thread walking all blocks took 0.3 ms for the 7 813 blocks of 1 M sources.
__global__ void k_blk_pos(const u32* __restrict__ blk_raw, u32* __restrict__ blk_pos, u64* __restrict__ scal, u32 nb) {
    __shared__ u32 s_part[1024];
    if (blockIdx.x != 0) return;
    const u32 nt = blockDim.x, tid = threadIdx.x;
    const u32 per = (nb + nt - 1) / nt, b0 = min(nb, tid * per), b1 = min(nb, b0 + per);
    u32 sum = 0;
    for (u32 b = b0; b < b1; ++b) sum += ((blk_raw[b + 1] - blk_raw[b] + 3u) & ~3u) + WIN;
    s_part[tid] = sum;
    __syncthreads();
    for (u32 o = 1; o < nt; o <<= 1) {   // inclusive scan of the per-thread sums
        const u32 v = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    u32 pos = s_part[tid] - sum;
    for (u32 b = b0; b < b1; ++b) {
        blk_pos[b] = pos;
        pos += ((blk_raw[b + 1] - blk_raw[b] + 3u) & ~3u) + WIN;
    }
    if (tid == nt - 1) { blk_pos[nb] = s_part[nt - 1]; scal[3] = s_part[nt - 1]; }
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

// distinct ranks of every block, from the dense group list to the padded layout.
template <class V>
__global__ void k_emit_keys(const u32* __restrict__ grank, const V* __restrict__ vals, const u32* __restrict__ estart,
                            const u32* __restrict__ blk_raw, const u32* __restrict__ blk_pos, u32* __restrict__ brk,
                            u64 n_groups) {
    const u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_groups) return;
    const u32 b = tag_of(vals[estart[d]]) >> 8;
    brk[blk_pos[b] + ((u32)d - blk_raw[b])] = grank[d];
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
    if (W) bw[dst] = weight_of(v0);
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
        if (pr) {   // (the key-by-key build already has its pairs in rank order)
            pr[dst + i] = brk[src + i];
            pb[dst + i] = b;
        }
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
        const u32 J = pb[j], A = min(I, J), B = max(I, J);
        const u64 t = tile_row_start_dev(A, nb) + (B - A);
        if (!flags[t]) flags[t] = 1;
    }
}
__global__ void k_pack_flags(const unsigned char* __restrict__ flags, u64 n, u32* __restrict__ bits) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;   // blockDim is a multiple of 64
    const unsigned long long m = __ballot(t < n && flags[t] != 0);
    // (only waves that hold tiles write: the grid is rounded up to whole workgroups, and the words behind the
    //  bitmap are the flags of the first tiles — a trailing wave used to zero them while the first wave read them)
    if ((threadIdx.x & 63) == 0 && t < n) { bits[t >> 5] = (u32)m; bits[(t >> 5) + 1] = (u32)(m >> 32); }
}

// ---- match records (see join_matches_counters in join_kernels.hip.h) -----------------------------
// The group records of the key-by-key build are in rank order: the list words of one key are adjacent.  Every
// pair of them is one match of the join — (tile of the two blocks, posting word of the lower block, posting
// word of the higher block).  Counted per leading word, scanned, emitted, sorted by tile (host side).
__global__ void k_match_count(const u32* __restrict__ rank, u32 n_groups, u32* __restrict__ cnt) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const u32 r = rank[g];
    u32 c = 0;
    for (u32 j = g + 1; j < n_groups && rank[j] == r; ++j) ++c;
    cnt[g] = c;
}
__global__ void k_match_emit(const u32* __restrict__ rank, const u32* __restrict__ blk, const u64* __restrict__ val,
                             u32 n_groups, const u64* __restrict__ off, u32 nb, u32* __restrict__ mt, u64* __restrict__ mr) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const u32 r = rank[g], bg = blk[g], ig = (u32)val[g];
    u64 o = off[g];
    for (u32 j = g + 1; j < n_groups && rank[j] == r; ++j, ++o) {   // (a key's first word is its first holder's block, the others ascend)
        const u32 bj = blk[j], ij = (u32)val[j];
        const u32 I = min(bg, bj), J = max(bg, bj);
        mt[o] = (u32)(tile_row_start_dev(I, nb) + (J - I));
        mr[o] = bg < bj ? ((u64)ig | ((u64)ij << 32)) : ((u64)ij | ((u64)ig << 32));
    }
}
// first record of every tile of the work list (tid: the tile ids, ascending)
__global__ void k_match_bounds(const u32* __restrict__ tid, u32 n_act, const u32* __restrict__ mt, u32 n_rec,
                               u32* __restrict__ mstart) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_act) return;
    if (i == n_act) { mstart[i] = n_rec; return; }
    const u32 t = tid[i];
    u32 lo = 0, hi = n_rec;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (mt[mid] < t) lo = mid + 1; else hi = mid;
    }
    mstart[i] = lo;
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
        ftags[dst + i] = make_tag<V>(tag, W ? wts[src + i] : 0u);
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

