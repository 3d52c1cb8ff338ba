// kspider_amd engine — hand-written HIP for gfx950 (MI355X, CDNA4).
//
// What it replaces: the accumulate region of kSpider::pairwise()
// (/root/reference/src/pairwise.cpp:194-237): for every colour/k-mer, all C(m,2)
// source pairs are pushed through a 4096-shard mutex-protected hash map
// (PAIRS_COUNTER, :22-27).  Here the per-source hash sets live in HBM as sorted
// uint64 runs and the N x N shared-k-mer matrix is produced tile by tile:
//
//   stage 1 (build_blocks / build_postings; kernels in stage1_kernels.hip.h)  equal hashes are brought
//            together (partition by the top key bits + LDS hash buckets; or a prefix sort) and every hash
//            held by at least two sources is replaced by a dense 32-bit rank (exact: equal hash <=> equal
//            rank); sources that share keys are moved next to each other (label pass); the holders of
//            every key are cut into blocks of TB = 128 sources, which gives ONE sorted list of distinct
//            ranks per block with postings (which of the 128 sources hold the key); a bitmap says which
//            block pairs share a key at all.
//   stage 2 (k_join; join_kernels.hip.h)  a work list of the active tiles, cut into shares by estimated
//            work.  A share searches rank-aligned cells of the two lists (4-ary tree in LDS) and
//            accumulates the matches — in LDS pair counters (weighted input, single-source postings) or
//            bit-sliced in registers (multi-source postings: membership masks transposed into bit columns,
//            popcount(a & b) per 4 x 4 patch of source pairs) — and emits (source_1, source_2, shared).
//
// Integer set intersection: no MFMA.  See DESIGN.md for the data layout, the measurements and the history.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <set>
#include <thread>
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
#include "host_sync.h"

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

#include "stage1_kernels.hip.h"
#include "partition_kernels.hip.h"
#include "fused_kernels.hip.h"
#include "join_kernels.hip.h"

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
    bool padded = false;          // nb holds spare blocks: block boundaries respect the clusters, some slots are holes (k_pack_blocks)
    std::vector<u32> h_blk_src;   // nb + 1: sources in the blocks before block b (= b x 128 without holes)
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
    bool post_slice = false;      // ... of a postings input (ksp_engine_build_postings_slice): post_off stays valid until the finish
    u64 slice_hdr[4] = {0, 0, 0, 0};   // padded length, distinct keys (U), big postings, block keys
    u32 ncell = ksp::NP;          // fine rank cells per block (power of two)
    bool use_cells = true;        // rank-aligned cell join (KSP_JOIN=window selects the sliding-window merge)
    bool full_sort = false;       // keys defeat the 32-bit prefix sort: use all bits
    bool hash_off = false;        // keys defeat the bucket grouping (a bucket overflowed): use the sort path
    bool pre_zeroed_work = false, pre_zeroed_bits = false;   // dwork / tbits were zeroed by the build's one zeroing launch
    bool part_off = false;        // the hand-written partition gave up on these keys (page tables full): rocPRIM partition
    u32 part_min = 4096;          // entries from which the hand-written partition is used (KSP_PART_MIN)
    ksp::Buf PK, PT, PD, parena;  // level-1 pages of the partition: keys, tags, digit bytes; pools, cursors, page tables
    bool seg_off = false;         // the segment partition gave up on these keys (a tile or a bucket overflowed): paged levels
    ksp::Buf seg_tbl, seg_grp, seg_chk;   // segment partition: per source the first entry of every range; first source of every group; chunks of k_seg_bounds
    std::vector<uint4> seg_chunks;
    std::vector<u32> seg_groups;  // host copy of the groups (valid while the offsets and the range count are unchanged)
    u32 seg_groups_nb1 = 0;
    bool seg_groups_ok = false;
    ksp::Buf PK2, PT2, PD2;       // pages of the middle level (more than 65 536 buckets)
    ksp::Buf biglist;             // buckets above the LDS table's entry capacity (k_bucket_big)
    // phase timers (ksp_engine_set_profiling): events at the phase starts of the last build / join
    bool profiling = false;
    static constexpr int kMaxPhase = 24;
    hipEvent_t ph_ev[kMaxPhase] = {};
    const char* ph_name[kMaxPhase] = {};
    int ph_n = 0;
    float ph_ms[kMaxPhase] = {};
    u32 hb_slots = 0;             // workgroups of k_bucket_group the device holds at once
    bool key_groups_off = false;  // a key has too many holders for the key-by-key list build: sort the entries by block
    bool fused_off = false;       // the bucket-resident middle of stage 1 (fused_kernels.hip.h) cannot take these keys (an oversize bucket,
                                  // sparse sharing that wants match records): the pass-by-pass kernels from now on
    bool fused_flags = false;     // the tile flags and the diagonal work of this build were written with the block lists (k_fms_place, k_fkeys)
    int fused_used = 0;           // (stats) the last build took the bucket-resident path
    bool have_rank_pairs = false; // gp holds (block, rank) of every list word in rank order (key-by-key build)
    bool have_dwork = false;      // ... and dwork the diagonal work / holder sums (k_move_groups)
    ksp::Buf gp, gm, ms_hist;     // group records of the key-by-key build; parked masks; per-chunk block counts of the split (k_ms_*)
    ksp::Buf crank;               // rank of every 4 096th kept entry (what the key-by-key list build reads instead of a rank per entry)
    bool rank1_ok = false;        // R1 holds a rank per kept entry (sort path, postings input; the bucket grouping only writes crank)
    ksp::Buf pmask;               // the membership mask of every list word at its list position (written by k_ms_place)
    bool pmask_on = false;        // ... of the lists the engine holds
    bool scal_fresh = false;      // h_scal[1 .. 11] hold the finished build's values (read back before its last kernels were queued)
    u64 gp_stride = 0;            // gp: gp_stride record values (u64), then as many blocks, ranks and sorted blocks (u32)
    bool reorder = true;          // order the sources by shared-key label before cutting blocks (KSP_REORDER=0: off)
    bool need32 = false;          // some tile pairs two blocks that both hold a source with >= 2^16 k-mers
    int key_bits = 64;
    std::vector<u64> h_off;
    std::vector<u32> h_blk_off;   // distinct-key offsets of the block lists (host copy)
    std::vector<u32> h_blk_max;   // per block: largest per-source k-mer count / weight sum
    std::vector<unsigned char> act32;   // work list: some share of the active tile counts in 32 bits (the second pass of the join is launched for those only)
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
    ksp::Buf tbits, dwork, d_act, d_wg, ticket;   // (ticket: the arrival counter of k_pack_flags_out's workgroups)
    // match-list join (inputs whose list words have few holders): records sorted by tile, first record per active tile
    ksp::Buf mcnt, moff, mt0, mt1, mr0, mr1, mstart;
    bool matches_on = false;      // mt1 / mr1 hold this build's records
    u64 n_matches = 0;
    unsigned char* h_stage = nullptr;   // pinned: diagonal work + overflow flag, then the tile bitmap
    size_t h_stage_bytes = 0;
    unsigned char* h_blk_stage = nullptr;   // pinned: per-block maxima, then the list offsets (stage_block_tables)
    size_t h_blk_stage_bytes = 0;
    bool blk_staged = false;
    bool d_off_sketch = false;          // d_off holds the sketch offsets of h_off (an unchanged set is not uploaded again)
    std::vector<u32> wg_host;           // share -> active tile (kept alive for the asynchronous upload)
    u32* h_sched = nullptr;             // pinned: a small work list (records, then shares) the join reads in place — no upload
    size_t h_sched_words = 0;
    bool sched_in_host = false;
    std::vector<u32> h_mstart;          // match-list mode: first record of every active tile (+ sentinel)
    unsigned long long* h_count = nullptr;   // pinned
    u64* h_scal = nullptr;                   // pinned: [0] max key, [1] Ktot, [2] U
    u64 sort_entries = 0;                    // entries / key bits of the last global radix sort (stats)
    int sort_bits = 0;
    int part_kind = 0;                       // 1 rocPRIM, 2 hand-written partition (stats)
    int part_fail = 0;                       // overflow word of the hand-written partition when it gave up (stats)
    u64 h_scal_words = 0, h_scal_keys = 0;   // list words / distinct shared keys of the lists being finished
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // build, join, first radix sort
    bool time_sort = true;        // events around the partition / first sort (st.ms_sort): an event record is a ~6 us bubble in the stream
    bool lean = false;            // ksp_engine_step_launch without profiling: no timing events at all (st.ms_build / ms_join / ms_sort stay 0),
                                  // the join's count arrives with a sequence number the wait polls (h_count[6]) instead of ev_join_done
    bool join_flag = false;       // the pending join signals through that number
    unsigned long long join_seq = 0;
    hipEvent_t ev_join_done = nullptr;   // behind the count copy of the last ksp_engine_join_launch
    // early work list (ksp_engine_step_launch): the tile flags and block tables leave for the host BEHIND the histogram of
    // the stable split and IN FRONT of its placement pass — the host cuts the join's shares while k_ms_place / k_cidx run,
    // and the join is queued behind them before they end (no idle device between a build and its join)
    hipEvent_t ev_sched = nullptr;
    bool early_ok = false;        // the caller queues the join itself right after the build (step_launch)
    bool sched_early = false;     // this build did it
    bool build_ms_pending = false;   // st.ms_build is read off ev[0] / ev[1] once the build's last kernel has ended
    bool sched_signalled = false;    // the early copy-out signals through a sequence number in pinned memory (h_count[7]), not ev_sched
    unsigned long long sched_seq = 0;
    bool ticket_zeroed = false;
    hipEvent_t ev_rb = nullptr;          // behind a read-back the host waits for while later kernels are already queued
    // ... or no event at all: the words go to pinned memory by a kernel of ours (k_readback) and a sequence number follows
    // them; the host polls the number (h_scal[15]).  An event record is a ~6 us bubble in the stream.
    unsigned long long rb_seq = 0;       // sequence number of the last k_readback
    bool rb_flag = false;                // the pending read-back signals through the number, not through ev_rb
    hipStream_t rb_stream = nullptr;
    double kept_frac = 0.7;              // kept / all entries of the previous build (label sampling before the count is known)
    double piece_ratio = 0;              // ksp_engine_join_to_host: densest found / bound ratio of the engine's previous call (first piece's size)
    ksp::Buf stage[2];                   // ksp_engine_join_to_host: the edges of a piece wait here for their copy
    hipStream_t copy_stream = nullptr;   // ... which runs on this stream, under the join of the next piece
    hipStream_t aux_stream = nullptr;    // the 32-bit-counter pass of a join runs here, beside the 16-bit pass
    hipEvent_t ev_aux[2] = {nullptr, nullptr};
    hipEvent_t ev_copy[2] = {nullptr, nullptr};
    bool join_pending = false;           // a launched join whose count has not been collected (ksp_engine_join_wait)
    u64 join_cap = 0;
    ksp_stats jst{};                     // last_* of the launched join
#ifdef KSP_WGTIME
    ksp::Buf wgt_buf;
    size_t wgt_n = 0;
#endif
    ksp_stats st{};
};

namespace ksp {

static inline unsigned grid_for(u64 n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// wait for the read-back behind which e->ev_rb was recorded (later kernels may already be queued on the stream)
static inline hipError_t wait_readback(ksp_engine* e) {
    if (e->rb_flag) {
        e->rb_flag = false;
        volatile unsigned long long* f = reinterpret_cast<volatile unsigned long long*>(e->h_scal + 15);
        for (unsigned long long spins = 1; *f != e->rb_seq; ++spins) {
            if ((spins & 0xFFFFF) == 0) {   // (now and then: is the stream still alive?)
                const hipError_t q = hipStreamQuery(e->rb_stream);
                if (q == hipSuccess) return *f == e->rb_seq ? hipSuccess : hipErrorUnknown;   // (idle, and the number never came)
                if (q != hipErrorNotReady) return q;
            }
        }
        return hipSuccess;
    }
    static const int mode = [] { const char* m = std::getenv("KSP_DEBUG_RBWAIT"); return m ? std::atoi(m) : 0; }();
    if (mode == 1) return hipEventSynchronize(e->ev_rb);
    hipError_t err;
    while ((err = hipEventQuery(e->ev_rb)) == hipErrorNotReady) {}
    return err;
}

// Blocks of a build.  With the source reordering on, half as many again as the sources need: the spare slots let
// clusters of up to 128 related sources start a block instead of straddling two (k_pack_blocks); what is not needed
// stays empty (blocks without keys cost nothing).  Every engine that builds the same source set gets the same count.
static u32 blocks_for(const u32 n_sources, const bool reorder) {
    const u32 nb0 = (n_sources + TB - 1) / TB;
    const char* al = std::getenv("KSP_ALIGN");   // 0: plain cuts every 128 sources (diagnostic / tests)
    if (!reorder || n_sources <= (u32)TB || n_sources > PACK_MAX || (al && std::atoi(al) == 0)) return nb0;
    u32 nb = nb0 + nb0 / 2 + 1;
    if (n_sources <= 65536u) nb = std::max(nb0, std::min<u32>(nb, 65536u / TB));   // (16-bit entry tags hold block < 512)
    return nb;
}
// stride of the per-source / per-slot maps in smap
static size_t smap_stride(const ksp_engine* e) {
    return (std::max<size_t>(e->n_sources, (size_t)e->nb * TB) + 64) & ~(size_t)63;
}

// start of a phase of stage 1 / the join (only with ksp_engine_set_profiling: an event per phase start)
static inline void phase_mark(ksp_engine* e, hipStream_t st, const char* name) {
    if (!e->profiling || e->ph_n >= ksp_engine::kMaxPhase) return;
    (void)hipEventRecord(e->ph_ev[e->ph_n], st);
    e->ph_name[e->ph_n++] = name;
}
// durations of the phases marked since the last reset; `end`: event behind the last phase (stream is idle)
static inline void phase_close(ksp_engine* e, hipEvent_t end) {
    for (int i = 0; i < e->ph_n; ++i) {
        e->ph_ms[i] = 0;
        (void)hipEventElapsedTime(&e->ph_ms[i], e->ph_ev[i], i + 1 < e->ph_n ? e->ph_ev[i + 1] : end);
    }
}

// The label pass looks at one key in (result + 1): ~64 sampled shared keys per source say as much about a
// source's relatives as all of them.  "Per source" is the small end of the size distribution (the 10th
// percentile of up to 1024 evenly spaced sources), so that small sketches among large ones still get their
// labels (lognormal sizes: C4).
static u32 label_sampling(const ksp_engine* e, u64 kept_entries) {
    const u32 N = std::max<u32>(1, e->n_sources);
    u64 small = kept_entries / N / 4;
    if (e->h_off.size() == (size_t)N + 1 && e->n_entries) {
        std::vector<u64> sz;
        const u32 take = std::min<u32>(N, 1024);
        sz.reserve(take);
        for (u32 i = 0; i < take; ++i) {
            const size_t s = (size_t)((u64)i * N / take);
            sz.push_back(e->h_off[s + 1] - e->h_off[s]);
        }
        std::nth_element(sz.begin(), sz.begin() + take / 10, sz.end());
        small = (u64)((double)sz[take / 10] * (double)kept_entries / (double)e->n_entries);
    }
    u32 every = 1;
    while (every < 64 && small / (2 * every) >= 64) every *= 2;
    if (const char* ev = std::getenv("KSP_DEBUG_LABEL_EVERY")) every = (u32)std::max(1, std::atoi(ev));   // (timing experiments)
    return every - 1;
}

// several small regions zeroed by ONE launch (a build used to issue ~10 runtime fills of a few bytes to a few
// hundred KB each; every one of them is a dispatch of its own)
struct ZeroList {
    u32* p[8];
    u32 words[8];
    int n;
};
__global__ void k_zero_regions(const ZeroList z) {
    for (int r = 0; r < z.n; ++r)
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < z.words[r]; i += gridDim.x * blockDim.x) z.p[r][i] = 0;
}
// several small device arrays to pinned host memory by ONE launch (the kernel stores straight into the mapped host
// buffer; every runtime copy is a dispatch of its own, and a build ended with six of them in a row)
struct CopyList {
    const u32* src[4];
    u32* dst[4];
    u32 words[4];
    int n;
};
__global__ void k_copy_regions(const CopyList c) {
    for (int r = 0; r < c.n; ++r)
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < c.words[r]; i += gridDim.x * blockDim.x) c.dst[r][i] = c.src[r][i];
}
// a few words of the scalar block to pinned host memory, a sequence number behind them (the host polls it: wait_readback)
__global__ void k_readback(const u64* __restrict__ src, u64* __restrict__ dst_host, const u32 words,
                           unsigned long long* __restrict__ flag_host, const unsigned long long seq) {
    if (threadIdx.x < words) __hip_atomic_store(dst_host + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... or riding along with the next kernel (k_label, k_move_groups): the same words, no dispatch of their own
static inline Rider ride_readback(ksp_engine* e, hipStream_t st, const u64* src, u64* dst_host, const u32 words) {
    e->rb_seq += 1;
    e->rb_flag = true;
    e->rb_stream = st;
    return Rider{src, dst_host, reinterpret_cast<unsigned long long*>(e->h_scal + 15), e->rb_seq, words};
}
static inline void launch_readback(ksp_engine* e, hipStream_t st, const u64* src, u64* dst_host, const u32 words) {
    e->rb_seq += 1;
    e->rb_flag = true;
    e->rb_stream = st;
    hipLaunchKernelGGL(k_readback, dim3(1), dim3(64), 0, st, src, dst_host, words, reinterpret_cast<unsigned long long*>(e->h_scal + 15), e->rb_seq);
}
// the tile flags packed straight into pinned host memory, and a few small regions copied out by the same launch (the
// work list's inputs leave the device in one dispatch instead of three: k_pack_flags, two k_copy_regions)
// ticket / flag_host / seq (ticket != nullptr): the workgroup that finishes last writes the sequence number behind everybody's
// words — the host polls it instead of an event (ticket: zero at launch, left at zero).
__global__ void k_pack_flags_out(const unsigned char* __restrict__ flags, const u64 n, u32* __restrict__ bits_out, const CopyList c,
                                 u32* __restrict__ ticket, unsigned long long* __restrict__ flag_host, const unsigned long long seq) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;   // blockDim is a multiple of 64
    const unsigned long long m = __ballot(t < n && flags[t] != 0);
    if ((threadIdx.x & 63) == 0 && t < n) { bits_out[t >> 5] = (u32)m; bits_out[(t >> 5) + 1] = (u32)(m >> 32); }
    for (int r = 0; r < c.n; ++r)
        for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < c.words[r]; i += gridDim.x * blockDim.x) c.dst[r][i] = c.src[r][i];
    if (ticket) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
            __threadfence_system();
            *ticket = 0;
            __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
static inline void copy_add(CopyList& c, const void* src, void* dst, size_t bytes) {
    if (!bytes) return;
    c.src[c.n] = (const u32*)src;
    c.dst[c.n] = (u32*)dst;
    c.words[c.n] = (u32)((bytes + 3) / 4);
    ++c.n;
}
static inline void zero_add(ZeroList& z, void* p, size_t bytes) {
    if (!bytes) return;
    z.p[z.n] = (u32*)p;
    z.words[z.n] = (u32)((bytes + 3) / 4);
    ++z.n;
}

static int launch_sched_kernels(ksp_engine* e, hipStream_t st, bool with_tables = false, bool signal = false);
static int stage_block_tables(ksp_engine* e, hipStream_t st);
// match records for the join (launch_sched_kernels): sparse sharing — few holders per list word
static bool sched_wants_matches(const ksp_engine* e, const u64 K, const bool ranked) {
    const char* jm = std::getenv("KSP_JOIN");
    const bool force = jm && std::string(jm) == "matches", never = jm && std::string(jm) != "matches";
    return ranked && !e->weighted && !never && K < (1ull << 31) && (force || e->n_kept < 4 * K);
}
// st.ms_build of a build whose work list left early: the build's last kernel has ended by the time anybody asks
static hipError_t resolve_build_ms(ksp_engine* e) {
    if (!e->build_ms_pending) return hipSuccess;
    e->build_ms_pending = false;
    hipError_t qe;
    while ((qe = hipEventQuery(e->ev[1])) == hipErrorNotReady) {}
    if (qe != hipSuccess) return qe;
    return hipEventElapsedTime(&e->st.ms_build, e->ev[0], e->ev[1]);
}

// What the parts of a stage-1 build share: the engine's arrays under the names its passes use (build_impl fills it in).
template <class V>
struct Stage1 {
    ksp_engine* e;
    hipStream_t st;
    int phase;
    u64 n;              // entries of this build
    u32 N, nb;          // sources, blocks
    u64* KA;
    V *VA, *VB;
    u64* scal;
    u32 *blk_raw, *blk_pos, *rank1, *crank;
    u32 *label, *iota, *labs, *order, *newidx, *sbound, *sorted_src, *blk_src;   // the per-source maps (smap)
    int bbits;          // bits of a block id
    bool reorder, hand_zeroed;
    u64 m;              // kept entries
};
// the names of a Stage1 as locals (the bodies below were written inside build_impl)
#define KSP_STAGE1_LOCALS(c)                                                                                             \
    constexpr bool W = std::is_same<V, u64>::value;                                                                      \
    ksp_engine* const e = (c).e;                                                                                         \
    const hipStream_t st = (c).st;                                                                                       \
    const int phase = (c).phase;                                                                                         \
    const u64 n = (c).n, m = (c).m;                                                                                      \
    const u32 N = (c).N, nb = (c).nb;                                                                                    \
    const unsigned bs = 256;                                                                                             \
    u64* const KA = (c).KA;                                                                                              \
    V *const VA = (c).VA, *const VB = (c).VB;                                                                            \
    u64* const scal = (c).scal;                                                                                          \
    u32 *const blk_raw = (c).blk_raw, *const blk_pos = (c).blk_pos, *const rank1 = (c).rank1, *const crank = (c).crank;   \
    u32 *const label = (c).label, *const iota = (c).iota, *const labs = (c).labs, *const order = (c).order;              \
    u32 *const newidx = (c).newidx, *const sbound = (c).sbound, *const sorted_src = (c).sorted_src, *const blk_src = (c).blk_src; \
    const int bbits = (c).bbits;                                                                                         \
    const bool reorder = (c).reorder, hand_zeroed = (c).hand_zeroed;                                                     \
    int rc = KSP_OK;                                                                                                     \
    size_t tb = 0;                                                                                                       \
    (void)W; (void)phase; (void)n; (void)m; (void)N; (void)nb; (void)bs; (void)KA; (void)VA; (void)VB; (void)scal;       \
    (void)blk_raw; (void)blk_pos; (void)rank1; (void)crank; (void)label; (void)iota; (void)labs; (void)order;            \
    (void)newidx; (void)sbound; (void)sorted_src; (void)blk_src; (void)bbits; (void)reorder; (void)hand_zeroed;          \
    (void)rc; (void)tb

// the sources in (label, id) order: engine index of every source, block boundaries at cluster boundaries
template <class V>
static int stage1_source_order(const Stage1<V>& c) {
    KSP_STAGE1_LOCALS(c);
    if (reorder) {
        // order the sources by (label, id) and move the kept entries to the new indices
        int lbits = 1;
        while (lbits < 32 && (N >> lbits)) ++lbits;
        tb = 0;
        u32* sort_out = e->padded ? sorted_src : order;
        KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, label, labs, iota, sort_out, (size_t)N, 0, lbits, st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, label, labs, iota, sort_out, (size_t)N, 0, lbits, st));
        if (e->padded) {   // block boundaries at cluster boundaries where the spare slots allow
            hipLaunchKernelGGL(k_pack_blocks, dim3(1), dim3(1024), 0, st, labs, N, nb, blk_src);
            hipLaunchKernelGGL(k_place_sources, dim3(grid_for((u64)nb * TB, bs)), dim3(bs), 0, st, sorted_src, blk_src, newidx, order,
                               sbound, e->blk_max.as<u32>(), nb);
        } else {
            hipLaunchKernelGGL(k_perm_bound, dim3(grid_for(N, bs)), dim3(bs), 0, st, order, newidx, sbound, e->blk_max.as<u32>(), N);
        }
    }
    return KSP_OK;
}

// The block lists, key by key (k_key_groups ... k_ms_place / the library sort of the group records ... k_cidx).  done = false:
// a key with thousands of holders in too many blocks — the caller takes the sort-by-block path (and the engine remembers).
template <class V>
static int stage1_lists_by_key(const Stage1<V>& c, bool& done) {
    KSP_STAGE1_LOCALS(c);
    done = false;
    // ---- the block lists, key by key (see k_key_groups) ------------------------------------------------
    if (!e->key_groups_off && m < (1ull << 32) - KG_CHUNK) {
        const u32 U = (u32)e->h_scal[2];
        const bool post_in = phase == 3 || e->post_slice;   // postings input: the key offsets are the caller's
        const u32* firstp = post_in ? e->post_off : (const u32*)e->FK.p;   // where each key's entries start (sentinel at U)
        u64* gsum = (u64*)KA;                 // per key: groups | masks << 32   (KA: the sorted keys are dead)
        u64* goff = gsum + (U + 2);
        u32* tmp_blk = (u32*)e->KB.p;         // records parked at entry positions (KB: free since the grouping by key)
        u32* tmp_info = tmp_blk + (m + 2);
        // gm: per key the mask, block and posting word of its first group, then the parked masks of further groups
        if ((rc = e->gm.ensure(((size_t)U + 4) * 24 + ((size_t)m / 4 + 4) * 16 + KG_HUGE_CAP * 4))) return rc;
        uint4* mask0 = e->gm.as<uint4>();
        uint4* tmp_mask = mask0 + (U + 4);
        u32 *blk0 = (u32*)(tmp_mask + (m / 4 + 4)), *info0 = blk0 + (U + 4);
        u32* huge_list = nb <= KG_HUGE_NB ? info0 + (U + 4) : nullptr;   // keys with more than KG_MAXC holders
        u32* wkey = W ? (u32*)VB : nullptr;   // (VB: the partitioned tags are dead)
        u32* d_kovf = (u32*)(scal + 11);
        const u32 chunks = (u32)((m + KG_CHUNK - 1) / KG_CHUNK);
        phase_mark(e, st, "key groups");
        if (!(hand_zeroed && phase == 0)) KSP_HIP(hipMemsetAsync(d_kovf, 0, 8, st));   // (the hand-written partition's build zeroes the whole scalar block at its start)
        if (post_in) KSP_HIP(hipMemsetAsync(gsum, 0, ((size_t)U + 2) * 8, st));   // (keys without entries — postings input only — are visited by no chunk)
        hipLaunchKernelGGL((k_key_groups<V, W>), dim3(chunks), dim3(KG_THREADS), 0, st, VA, crank, firstp, newidx, (u32)m, U,
                           gsum, blk0, info0, mask0, tmp_blk, tmp_info, tmp_mask, wkey, d_kovf,
                           std::getenv("KSP_DEBUG_COOP") ? std::max<u32>(KG_COOP_MIN, (u32)std::atoi(std::getenv("KSP_DEBUG_COOP"))) : KG_COOP,   // (timing experiments: the wave-per-key threshold)
                           huge_list);
        if (huge_list)
            hipLaunchKernelGGL((k_key_groups_huge<V, W>), dim3(256), dim3(256), 0, st, VA, firstp, newidx, nb, gsum, blk0,
                               info0, mask0, tmp_blk, tmp_info, tmp_mask, wkey, d_kovf, huge_list);
        {   // goff = exclusive prefix sums of gsum, the totals to scal[1] / scal[7] (stage1_kernels: k_pair_*)
            const u32 ntiles = grid_for(U, PS_TILE);
            if ((rc = e->tmp.ensure(((size_t)ntiles + 2) * 8))) return rc;
            hipLaunchKernelGGL(k_pair_tile_sums, dim3(ntiles), dim3(PS_THREADS), 0, st, gsum, U, (u64*)e->tmp.p);
            hipLaunchKernelGGL(k_pair_scan_tiles, dim3(ntiles), dim3(PS_THREADS), 0, st, gsum, goff, U, (const u64*)e->tmp.p, scal);
        }
        // the number of groups sizes the sort of the groups and what follows; the records themselves are packed (k_move_groups)
        // while the host waits for it: their arrays take the bound K <= m
        const Rider rb = ride_readback(e, st, scal + 1, e->h_scal + 1, 11);   // [1] groups ... [11] overflow: with k_move_groups
        const u64 Kcap = m;
        if ((rc = e->gp.ensure((Kcap + 4) * 20))) return rc;
        e->gp_stride = Kcap + 4;
        u64* rec_val = e->gp.as<u64>();
        u32 *rec_blk = (u32*)(rec_val + (Kcap + 4)), *rec_rank = rec_blk + (Kcap + 4), *sblk = rec_rank + (Kcap + 4);
        unsigned long long* work = nullptr;
        {
            // (the diagonal work and holder sums of the join's schedule come with the move when the blocks fit its LDS table)
            phase_mark(e, st, "block lists");
            if (nb <= KG_WORK && phase != 2) {
                if ((rc = e->dwork.ensure(((size_t)nb + 2) * 8))) return rc;
                if (!e->pre_zeroed_work) KSP_HIP(hipMemsetAsync(e->dwork.p, 0, ((size_t)nb + 2) * 8, st));
                e->pre_zeroed_work = false;
                work = e->dwork.as<unsigned long long>();
            }
            u32 mg_cap = 512u;
            if (const char* mc = std::getenv("KSP_DEBUG_MOVE_GRID")) mg_cap = (u32)std::max(1, std::atoi(mc));   // (timing experiments)
            hipLaunchKernelGGL(k_move_groups, dim3(std::min<u32>(grid_for(U, bs), mg_cap)), dim3(bs), 0, st, gsum, goff, firstp,
                               blk0, info0, mask0, tmp_blk, tmp_info, tmp_mask, rec_blk, rec_val, rec_rank, e->mm.as<uint4>(),
                               U, work, nb, d_kovf, rb);
        }
        KSP_HIP(wait_readback(e));
        if ((u32)e->h_scal[11]) {
            e->key_groups_off = true;   // a key with thousands of holders: this engine sorts by block from now on
        } else {
            const u64 K = std::max<u64>(1, e->h_scal[1]);
            u64* sval = (u64*)KA;             // (the per-key counts and offsets are dead once the records are packed)
            e->have_dwork = work != nullptr;
            const char* msv = std::getenv("KSP_MS");   // 0: the library sort (diagnostic / tests)
            // (up to 256 blocks; the 1 024-block tables are slower than the library's two radix passes — C3: block lists 1.04 -> 1.38 ms
            //  for 0.23 ms of join, C4 4.1 -> 5.3 — and only run when KSP_MS=1024 asks for them: tests)
            const u32 ms_max = (msv && std::atoi(msv) == 1024) ? MS_MAXB : 256u;
            if (nb <= ms_max && Kcap / MS_CHUNK < (1u << 20) && !(msv && std::atoi(msv) == 0)) {
                // stable split of the records on the block id, written straight into the padded lists (stage1_kernels: k_ms_*)
                const u32 mb = nb <= 256 ? 256u : 1024u;   // table size (two instantiations)
                const u32 chunks_cap = grid_for(Kcap, MS_CHUNK), chunks = grid_for(K, MS_CHUNK);
                if ((rc = e->ms_hist.ensure(((size_t)chunks_cap + 1) * mb * 4 + 4096))) return rc;
                u32* hist = e->ms_hist.as<u32>();
                u32* tot = hist + (size_t)chunks_cap * mb;
                if (mb == 256) hipLaunchKernelGGL(k_ms_hist<256>, dim3(chunks), dim3(MS_THREADS), 0, st, rec_blk, scal, hist);
                else hipLaunchKernelGGL(k_ms_hist<1024>, dim3(chunks), dim3(MS_THREADS), 0, st, rec_blk, scal, hist);
                hipLaunchKernelGGL(k_ms_scan, dim3(nb), dim3(256), 0, st, hist, mb, scal, tot, blk_raw, blk_pos, nb);
                // early work list (step_launch): everything the host needs to cut the join's shares exists now — the records in
                // rank order (tile flags), the diagonal work, the list offsets — and leaves for pinned memory in front of the
                // placement pass; the host works while k_ms_place and k_cidx run.  (On a stream of its own, beside the placement
                // pass, the hand-over between the streams cost more than the four small kernels: 1.334 against 1.320 ms per step.)
                e->sched_early = false;
                if (e->early_ok && phase == 0 && !e->profiling && e->n_kept && !sched_wants_matches(e, K, true) && !std::getenv("KSP_DEBUG_LATE_SCHED")) {
                    e->have_rank_pairs = true;
                    e->h_scal_words = e->h_scal[1];
                    e->h_scal_keys = e->h_scal[2];
                    if ((rc = launch_sched_kernels(e, st, true, true))) return rc;
                    if (!e->blk_staged && (rc = stage_block_tables(e, st))) return rc;
                    if (!(e->sched_signalled && e->blk_staged)) { e->sched_signalled = false; KSP_HIP(hipEventRecord(e->ev_sched, st)); }   // (else: the last workgroup of the copy-out writes a sequence number the host polls)
                    e->sched_early = true;
                }
                uint4* pm = nullptr;
                if (!W && m >= 4 * K) {   // unweighted lists with multi-source postings (no match records: the rule of
                                          // launch_sched_kernels): the join's bit-sliced paths read the masks at the list positions
                    if ((rc = e->pmask.ensure((Kcap + (u64)nb * (WIN + 4) + 4 * WIN) * 16))) return rc;
                    pm = e->pmask.as<uint4>();
                }
                if (mb == 256)
                    hipLaunchKernelGGL((k_ms_place<W, 256>), dim3(chunks), dim3(MS_THREADS), 0, st, rec_blk, rec_val, scal, hist, blk_pos, nb, wkey,
                                       e->bkeys.as<u32>(), e->info.as<u32>(), W ? e->bw.as<u32>() : nullptr, e->mm.as<uint4>(), pm, blk_raw, PAD);
                else
                    hipLaunchKernelGGL((k_ms_place<W, 1024>), dim3(chunks), dim3(MS_THREADS), 0, st, rec_blk, rec_val, scal, hist, blk_pos, nb, wkey,
                                       e->bkeys.as<u32>(), e->info.as<u32>(), W ? e->bw.as<u32>() : nullptr, e->mm.as<uint4>(), pm, blk_raw, PAD);
                e->pmask_on = pm != nullptr;
            } else {
            tb = 0;   // (temporary storage for the bound, not for this build's K: K moves a little from build to build — the labels'
                          //  atomics race — and a buffer that has to grow in the middle of a build costs a device-wide stall)
                KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, rec_blk, sblk, rec_val, sval, (size_t)Kcap, 0, bbits, st));
                if ((rc = e->tmp.ensure(tb))) return rc;
                tb = 0;
                KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, rec_blk, sblk, rec_val, sval, (size_t)K, 0, bbits, st));
                KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, rec_blk, sblk, rec_val, sval, (size_t)K, 0, bbits, st));
                if (nb <= 1024) {   // (one workgroup while the serial layout of the starts is short: 7 813 blocks took 0.32 ms in it)
                    hipLaunchKernelGGL(k_blk_raw_pos, dim3(1), dim3(1024), 0, st, sblk, (u32)e->h_scal[1], blk_raw, blk_pos, scal, nb);
                } else {
                    hipLaunchKernelGGL(k_blk_raw_groups, dim3(grid_for((u64)nb + 1, bs)), dim3(bs), 0, st, sblk, (u32)e->h_scal[1], blk_raw, nb);
                    hipLaunchKernelGGL(k_blk_pos, dim3(1), dim3(1024), 0, st, blk_raw, blk_pos, scal, nb);
                }
                hipLaunchKernelGGL(k_pad, dim3(nb + 1), dim3(256), 0, st, blk_raw, blk_pos, e->bkeys.as<u32>(), nb, PAD);
                hipLaunchKernelGGL((k_place_groups<W>), dim3(grid_for(K, bs)), dim3(bs), 0, st, sblk, sval, blk_raw, blk_pos, wkey,
                                   e->bkeys.as<u32>(), e->info.as<u32>(), W ? e->bw.as<u32>() : nullptr, (u32)e->h_scal[1]);
            }
            {   // the fine cell index was sized from the raw entries of a block; the lists are an order of magnitude shorter
                // (pruned, one word per key and block): ~32 words of an average list x 4 per cell is as fine as the join
                // ever looks (it merges cells up to ~216 keys anyway) — 16 x fewer bisections on C2 (32 -> 4 us)
                const u64 avg = K / nb + 1;
                u32 nc = NP;
                while ((u64)nc * 32 < 4 * avg && nc < e->ncell) nc <<= 1;
                e->ncell = std::min(e->ncell, nc);
            }
            hipLaunchKernelGGL(k_cidx, dim3(grid_for((u64)nb * (e->ncell + 1), bs)), dim3(bs), 0, st, e->bkeys.as<u32>(),
                               blk_raw, blk_pos, scal, e->part.as<u32>(), nb, e->ncell);
            KSP_HIP(hipGetLastError());
            e->scal_fresh = phase == 0;   // (h_scal[1] .. [11] are this build's: build_common need not fetch them again)
            e->have_rank_pairs = true;   // rec_rank / rec_blk: (rank, block) of every list word in rank order
            done = true;
            return KSP_OK;
        }
    }
    return KSP_OK;
}

// The block lists by sorting the kept entries by block (the fallback of stage1_lists_by_key, and inputs beyond its limits).
template <class V>
static int stage1_lists_by_sort(const Stage1<V>& c) {
    KSP_STAGE1_LOCALS(c);
    // ---- the block lists by sorting the entries by block -------------------------------------------------
    if (!e->rank1_ok) {   // (the grouping wrote crank[] only: a rank per entry from first[])
        const u32* fp = (phase == 3 || e->post_slice) ? e->post_off : (const u32*)e->FK.p;
        hipLaunchKernelGGL(k_rank_fill, dim3(1024), dim3(256), 0, st, fp, (u32)e->h_scal[2], rank1);
        e->rank1_ok = true;
    }
    if (reorder) hipLaunchKernelGGL((k_retag<V>), dim3(grid_for(m / (16 / sizeof(V)) + 1, bs)), dim3(bs), 0, st, VA, newidx, m);
    // sort 2: stable by block id (bits [8, 8+bbits) of the tag), payload = rank:  VA,rank1 -> VB,rk2
    u32* rk2 = (u32*)KA;                   // KA (sorted keys) is dead from here on
    tb = 0;
    const int bbeg = sizeof(V) == 2 ? 7 : 8;   // the block id inside a compact / canonical tag
    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, VA, VB, rank1, rk2, m, bbeg, bbeg + bbits, st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, VA, VB, rank1, rk2, m, bbeg, bbeg + bbits, st));
    // now: rk2 = ranks sorted by (block, rank); VB = tags in the same order.  KB, VA, R1 are free.
    V* T = VB;
    u32* flag = (u32*)VA;
    u32* grank = (u32*)e->KB.p;            // rank of every (block, key) group (up to m)
    u32* estart = (u32*)e->KB.p + (n + 2); // first entry of every group (up to m+1)
    const HeadFn<V> head{rk2, T};
    auto hf = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u64>(0), head);
    {
        HeadScatterIt<V> out{{head, estart, grank, scal, m}, 0};
        tb = 0;
        KSP_HIP(rocprim::inclusive_scan(nullptr, tb, hf, out, m, rocprim::plus<u32>(), st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::inclusive_scan(e->tmp.p, tb, hf, out, m, rocprim::plus<u32>(), st));
    }
    // the number of distinct (block, key) groups sizes the posting passes (after the source reordering it is
    // an order of magnitude below the entry count: one 8-byte read-back pays for itself)
    KSP_HIP(hipMemcpyAsync(e->h_scal + 1, scal + 1, 8, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL((k_blk_raw<V>), dim3(grid_for((u64)nb + 1, bs)), dim3(bs), 0, st, T, estart, scal, blk_raw, nb, m);
    hipLaunchKernelGGL(k_blk_pos, dim3(1), dim3(1024), 0, st, blk_raw, blk_pos, scal, nb);
    hipLaunchKernelGGL(k_pad, dim3(nb + 1), dim3(256), 0, st, blk_raw, blk_pos, e->bkeys.as<u32>(), nb, PAD);
    KSP_HIP(hipStreamSynchronize(st));
    const u64 K = std::max<u64>(1, e->h_scal[1]);
    u32* mmsz = flag;                      // (VA is free: the head flags are computed on the fly)
    u32* mmoff = (u32*)KA;                 // rk2 is dead after the head scan
    hipLaunchKernelGGL((k_emit_keys<V>), dim3(grid_for(K, bs)), dim3(bs), 0, st, grank, T, estart, blk_raw, blk_pos,
                       e->bkeys.as<u32>(), e->h_scal[1]);
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

template <class V>
static int build_impl(ksp_engine* e, const u64* d_keys, const u32* d_w, hipStream_t st, const int phase) {
    constexpr bool W = std::is_same<V, u64>::value;   // weighted: 64-bit tags carry the key's weight
    // phase 0: the whole of stage 1;  1: up to the source labels (key-range slice, before the labels of all
    // slices are combined);  2: the rest (source order from the final labels, block lists);  3: postings
    // input (ksp_engine_build_postings: sorting and pruning are already done by the caller's inverted index)
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
    if ((rc = e->blk_max.ensure((2 * (size_t)nb + 4) * 4))) return rc;
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
    if ((rc = e->crank.ensure((n / CR_CHUNK + 4) * 4))) return rc;
    u32* crank = e->crank.as<u32>();
    const size_t NN = smap_stride(e);
    if ((rc = e->smap.ensure(7 * NN * 4))) return rc;
    // per-source maps: [0] label, [1] iota, [2] sorted labels, [3] order (= engine index -> source id; ~0: a hole),
    // [4] newidx (source id -> engine index), [5] bound of a source's pair counters, [6] sources in (label, id) order
    u32* sm = e->smap.as<u32>();
    u32 *label = sm, *iota = sm + NN, *labs = sm + 2 * NN, *order = sm + 3 * NN, *newidx = sm + 4 * NN, *sbound = sm + 5 * NN;
    u32* sorted_src = sm + 6 * NN;
    u32* blk_src = e->blk_max.as<u32>() + ((size_t)nb + 1);   // (behind the per-block maxima: one staging copy for both)
    const bool reorder = e->reorder;
    u32 label_max = std::max<u32>(256, N / 16);   // holders above which a key is ignored by the label pass
    if (const char* lm = std::getenv("KSP_DEBUG_LABEL_MAX")) label_max = (u32)std::max(1, std::atoi(lm));
    // the label pass over the kept keys (first[] = where each key's entries start); KB is free whenever it runs
    auto run_label = [&](const u32* firstp, const u32 n_keys, const u64 kept, const u64* scal_dev = nullptr, const Rider rider = Rider{nullptr, nullptr, nullptr, 0, 0}) {
        const u32 skip = label_sampling(e, kept);
        // labels on memory lines of their own while they are lowered — for source sets whose labels would otherwise share a
        // few hundred lines (10 000 sources: 98 -> 39 us).  A large set spreads its atomics by itself, and 128 bytes per
        // source turn every look at a label into a line of its own from HBM: 1 M sources, k_label 2.46 ms and 17 GB fetched
        // (a quarter of that build) against a 4 MB table that stays in L2.
        u32 spread_max = 262144;   // (100 000 sources: still 1 % faster spread; 1 M: 0.9 ms slower)
        if (const char* sv = std::getenv("KSP_DEBUG_LABEL_SPREAD")) spread_max = (u32)std::atoi(sv);   // (timing experiments: sources up to which the labels are spread)
        const int ls = N <= spread_max && e->KB.bytes >= (size_t)N * 128 ? 5 : 0;
        u32* lab = ls ? (u32*)e->KB.p : label;
        if (ls) hipLaunchKernelGGL(k_label_spread, dim3(grid_for(N, bs)), dim3(bs), 0, st, lab, ls, N);
        hipLaunchKernelGGL((k_label<V>), dim3(grid_for(n_keys / (skip + 1) + 1, bs)), dim3(bs), 0, st, VA, firstp, lab, ls,
                           skip, label_max, n_keys, scal_dev, rider);
        if (ls) hipLaunchKernelGGL(k_label_gather, dim3(grid_for(N, bs)), dim3(bs), 0, st, lab, ls, label, N);
    };
    bool hand_zeroed = false;   // this build began with the hand-written partition's zeroing launch (the whole scalar block)
    int bbits = 1;
    while ((1u << bbits) < nb) ++bbits;
    size_t tb = 0;
    u64 m = e->n_kept;   // (phase 2: set by phase 1)
    bool order_queued = false;   // the source order (label sort, block cuts, placement) was queued together with the label pass
    if (phase == 3 || phase == 4) {   // (4: a slice of a postings input — stops at the labels, ksp_engine_slice_finish runs phase 2)
        m = n;
        e->n_kept = m;
        hipLaunchKernelGGL(k_iota4, dim3(grid_for(N, bs)), dim3(bs), 0, st, iota, order, newidx, label, N);
        KSP_HIP(hipMemsetAsync(e->blk_max.p, 0, ((size_t)nb + 1) * 4, st));
        KSP_HIP(hipMemsetAsync(sbound, 0, (size_t)N * 4, st));
        const u32 nk = e->post_nkeys;
        KSP_HIP(hipMemsetAsync(scal + 4, 0, 8, st));
        hipLaunchKernelGGL((k_post_expand<V, W>), dim3(grid_for(nk, bs)), dim3(bs), 0, st, e->post_off, e->post_src,
                           e->post_w, VA, rank1, sbound, nk, N, (u32*)(scal + 4));
        hipLaunchKernelGGL(k_crank_from_rank, dim3(grid_for(m / CR_CHUNK + 1, bs)), dim3(bs), 0, st, rank1, crank, (u32)m);   // (postings: a rank per entry exists)
        e->rank1_ok = true;
        {   // U = number of keys (what the prune scan reports on the sketch path)
            e->h_scal[2] = nk;
            KSP_HIP(hipMemcpyAsync(scal + 2, e->h_scal + 2, 8, hipMemcpyHostToDevice, st));
        }
        if (reorder) {
            run_label(e->post_off, nk, m);
        } else {
            hipLaunchKernelGGL(k_blk_bound, dim3(grid_for(N, bs)), dim3(bs), 0, st, sbound, newidx, e->blk_max.as<u32>(), N);
        }
        if (phase == 4) return KSP_OK;
    } else if (phase != 2) {

    // the hand-written partition (partition_kernels.hip.h) finds the key range on the device and makes the
    // source tags itself: no read-back, no tagging pass.  Unweighted whole builds whose bucket count fits
    // its two levels; everything else (weighted sketches, key-range slices, > 2^16 buckets) takes the
    // rocPRIM partition below.
    u32 nb_hand = 0;   // buckets of the hand-written partition (any number up to 2^21; above 2^16: three levels; 0: not used)
    if (!W && phase == 0 && e->nparts == 1 && !e->hash_off && !e->full_sort && !e->part_off && n >= e->part_min) {
        u32 mean = HB_HAND_MEAN;
        if (const char* bm = std::getenv("KSP_DEBUG_BUCKET_MEAN")) mean = (u32)std::max(8, std::atoi(bm));   // (timing experiments; tests: tiny buckets force the middle level)
        u64 want = (n + mean - 1) / mean;
        if (want > 65536 && (n + 65535) / 65536 <= HB_HAND_MEAN_MAX && !std::getenv("KSP_DEBUG_BUCKET_MEAN"))
            want = 65536;   // (somewhat larger buckets rather than a third level)
        if (want <= (256u * 32u * 256u) && !(want > 65536 && std::getenv("KSP_DEBUG_NO_MID"))) nb_hand = (u32)std::max<u64>(1, want);
    }
    const bool hand = nb_hand > 0;
    // the segment partition (k_seg_bounds / k_seg_scatter): two levels, runs long enough to leave a dozen entries per
    // (source, range) and no run so long that the one wave that streams it would be the launch
    bool seg = false, seg3_used = false;
    u32 seg_cap = 0, seg_nb1 = 0;
    // (one source's segment must fit a tile: a run so long that its share of one range would not is a build for the
    //  paged levels — known before anything is launched, instead of finding out from the overflow word)
    auto seg_runs_fit = [&](const u32 r1) {
        u64 longest = 0;
        for (u32 s_ = 0; s_ < N; ++s_) longest = std::max(longest, e->h_off[s_ + 1] - e->h_off[s_]);
        return longest / std::max<u32>(1, r1) <= SEG_FILL;
    };
    int seg_pb2 = 0;
    if (hand && nb_hand <= 65536 && !e->seg_off && e->h_off.size() == (size_t)N + 1 && N) {
        const char* sv = std::getenv("KSP_SEG");   // 0: never, 1: whatever the segment length (diagnostic / tests)
        while (((nb_hand + (1u << seg_pb2) - 1) >> seg_pb2) > 256) ++seg_pb2;
        // (fewer, longer segments — 256 final buckets per range, C2: 51 entries per segment instead of 26 — were measured with
        //  the vectorised boundary pass: partition 0.42 -> 0.44 ms, the shorter write runs cost more than the longer reads save)
        const int pb2_min = seg_pb2;
        if (const char* pv = std::getenv("KSP_DEBUG_SEG_PB2")) seg_pb2 = std::min(8, std::max(pb2_min, std::atoi(pv)));   // (timing experiments)
        seg_nb1 = (nb_hand + (1u << seg_pb2) - 1) >> seg_pb2;
        const u64 mean_seg = n / N / seg_nb1;
        seg = (sv ? std::atoi(sv) != 0 : mean_seg >= SEG_MIN_LEN && seg_runs_fit(seg_nb1));
        if (seg) {
            const u64 mean = n / nb_hand + 1;
            seg_cap = (u32)((mean + mean / 2 + 128 + 63) & ~63ull);   // (a multiple of 64 places: every bucket starts on a memory line)
            if ((u64)nb_hand * seg_cap + P2_TILE >= (1ull << 31)) seg = false;
        }
    }
    const u64 nslots = seg ? (u64)nb_hand * seg_cap + P2_TILE : n;   // places of the partitioned entries (fixed ranges per bucket: with gaps)
    if (seg) {
        if ((rc = e->KA.ensure((nslots + 4) * 8))) return rc;
        if ((rc = e->VB.ensure((nslots + 4) * sizeof(V)))) return rc;
        if ((rc = e->KB.ensure((nslots / 2 + 1 + 2 * (u64)nb_hand + nb_hand / 2 + 8) * 8))) return rc;
        KA = e->KA.as<u64>();   // (the buffers may have moved)
        VB = e->VB.as<V>();
    }
    e->pre_zeroed_work = e->pre_zeroed_bits = false;
    // layout of the hand-written partition's arena (see the partition step below): level 1 (`A`), and for more than
    // 65 536 buckets a middle level (`M`) between it and the final scatter
    int hp_pb2 = 0, hp_pbm = 0;
    u32 hp_nb1 = 0, hp_nchunks = 0, hp_ngroups = 0;
    size_t hp_zero_words = 0, hp_pages_a = 0, hp_pages_m = 0;
    PartLists hp_a{}, hp_m{};
    u32 *hp_gcnt = nullptr, *hp_gbase = nullptr, *hp_src = nullptr;
    if (hand) {
        if (nb_hand <= 65536) {
            while (((nb_hand + (1u << hp_pb2) - 1) >> hp_pb2) > 256) ++hp_pb2;
        } else {
            // three levels: the middle one orders whole pages in LDS and writes long runs, so it takes as many bits as
            // it can (up to 5) and level 1 — whose runs are a source's keys inside one bucket — as few as possible
            hp_pb2 = 8;
            const u32 pre = (nb_hand + 255u) >> 8;
            while (hp_pbm < 5 && ((pre + (1u << hp_pbm) - 1) >> hp_pbm) > 64) ++hp_pbm;
            while (((pre + (1u << hp_pbm) - 1) >> hp_pbm) > 256) ++hp_pbm;
        }
        const u32 prefixes = (nb_hand + (1u << hp_pb2) - 1) >> hp_pb2;           // groups of 2^pb2 final buckets
        hp_nb1 = (prefixes + (1u << hp_pbm) - 1) >> hp_pbm;                         // level-1 buckets (<= 256)
        hp_ngroups = hp_nb1 << hp_pbm;
        auto plan = [&](PartLists& pl, const u32 buckets, const u32 subs, size_t& pages) {
            const u32 lists = buckets * subs;
            pl.subs = subs;
            pl.ptw = (u32)std::min<u64>(P1_PTW_MAX, 8 * (((n / lists) >> P1_PLOG) + 1) + 16);
            const u64 per = n / subs;
            pl.pool_pages = (u32)((per >> P1_PLOG) + (per >> (P1_PLOG + 3)) + buckets + 8);   // pages per sub-list class
            pages = (size_t)pl.pool_pages * subs;
            return (size_t)subs * P1_LINE + (size_t)lists * P1_LINE + (size_t)lists * pl.ptw + pages;   // words, all zeroed per build
        };
        const size_t words_a = plan(hp_a, hp_nb1, P1_R, hp_pages_a);
        const size_t words_m = hp_pbm ? plan(hp_m, hp_ngroups, 2, hp_pages_m) : 0;
        hp_nchunks = grid_for(n, P1_CH);
        // arena: [level 1 | middle level | bucket counters] zeroed per build, then group bases and chunk sources
        hp_zero_words = words_a + words_m + ((size_t)nb_hand + 1);
        if ((rc = e->parena.ensure((hp_zero_words + hp_ngroups + 2 + hp_nchunks + 2) * 4))) return rc;
        auto carve = [&](PartLists& pl, u32* base, const u32 buckets) {
            const size_t lists = (size_t)buckets * pl.subs;
            pl.pools = base;
            pl.cursors = pl.pools + (size_t)pl.subs * P1_LINE;
            pl.pt = pl.cursors + lists * P1_LINE;
            pl.owner = pl.pt + lists * pl.ptw;
        };
        carve(hp_a, e->parena.as<u32>(), hp_nb1);
        if (hp_pbm) carve(hp_m, e->parena.as<u32>() + words_a, hp_ngroups);
        hp_gcnt = e->parena.as<u32>() + words_a + words_m;
        hp_gbase = hp_gcnt + ((size_t)nb_hand + 1);
        hp_src = hp_gbase + (hp_ngroups + 2);
        // everything this build needs zeroed, in one launch: the scalar block, the per-block maxima, the partition's
        // arena, the bucket totals, and (small inputs) the diagonal work and the tile bitmap of the work list
        ZeroList z{};
        zero_add(z, scal, 128);
        zero_add(z, e->blk_max.p, ((size_t)nb + 1) * 4);
        zero_add(z, e->parena.p, hp_zero_words * 4);
        zero_add(z, (u64*)e->KB.p + (nslots / 2 + 1), (size_t)nb_hand * 8);   // bsum (see the partition step)
        if (nb <= KG_WORK) {
            if ((rc = e->dwork.ensure(((size_t)nb + 2) * 8))) return rc;
            zero_add(z, e->dwork.p, ((size_t)nb + 2) * 8);
            e->pre_zeroed_work = true;
        }
        const u64 T = (u64)nb * (nb + 1) / 2;
        if (T <= (1ull << 22)) {
            const size_t bit_words = (size_t)(((T + 63) / 64) * 2 + 2);
            if ((rc = e->tbits.ensure(bit_words * 4 + T + 64))) return rc;
            zero_add(z, e->tbits.p, bit_words * 4 + T + 64);
            e->pre_zeroed_bits = true;
        }
        phase_mark(e, st, "key range + source sizes");
        hipLaunchKernelGGL(k_zero_regions, dim3(256), dim3(256), 0, st, z);
        hand_zeroed = true;
        hipLaunchKernelGGL(k_prep_sources, dim3(grid_for(N, bs)), dim3(bs), 0, st, d_keys, d_off, (unsigned long long*)scal, sbound,
                           iota, order, newidx, label, N);
    } else {
        KSP_HIP(hipMemsetAsync(scal + 4, 0, 8 * 11, st));   // [4] .. [14]: overflow words, counters of the partition
    }
    // key range (one 8-byte D2H, unless the caller passed key_bits)
    if (!hand && e->key_bits <= 0) {
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
    if (!hand) phase_mark(e, st, "tags + source sizes");
    if ((W || e->nparts == 1) && !hand)   // (weighted slices still need the per-source weight sums of all entries)
        hipLaunchKernelGGL((k_tag<V, W>), dim3(N), dim3(256), 0, st, d_off, d_w, VA, sbound);
    if (!hand) {
        if (!W) hipLaunchKernelGGL(k_src_size, dim3(grid_for(N, bs)), dim3(bs), 0, st, d_off, sbound, N);
        hipLaunchKernelGGL(k_iota4, dim3(grid_for(N, bs)), dim3(bs), 0, st, iota, order, newidx, label, N);   // (order, newidx: identity until the labels are known)
    }
    if (!hand) KSP_HIP(hipMemsetAsync(e->blk_max.p, 0, ((size_t)nb + 1) * 4, st));
    if (!reorder) hipLaunchKernelGGL(k_blk_bound, dim3(grid_for(N, bs)), dim3(bs), 0, st, sbound, newidx, e->blk_max.as<u32>(), N);
    // slice mode (multi-GPU build): keep only the entries of this part's key range — one contiguous
    // sub-run per (sorted) source, so the cost is proportional to the slice, not to the sketch set
    const u64* keys_in = d_keys;
    const V* tags_in = VA;
    u64 nw = n;
    int topbit = kbits;   // the bucket partition takes the bits just below this one
    if (e->nparts > 1) {
        // equal shares of [0, largest key] (or of [0, 2^key_bits) when the caller fixed key_bits)
        const unsigned __int128 span = e->have_max_key ? (unsigned __int128)e->max_key + 1
                                       : kbits >= 64   ? ((unsigned __int128)1 << 64)
                                                       : ((unsigned __int128)1 << kbits);
        const u64 lo = (u64)((span * e->part_id) / e->nparts);
        const u64 hi = (u64)((span * (e->part_id + 1)) / e->nparts - 1);
        // a slice spans hi - lo: the bits below that width spread its keys evenly over the buckets (the bits
        // above take at most two values inside the slice; the buckets compare whole keys anyway)
        topbit = 1;
        while (topbit < kbits && ((hi - lo) >> topbit)) ++topbit;
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
        if (nw >= (1ull << 30)) { set_error("build_slice: more than 2^30 key entries in this slice (use more parts)"); return KSP_E_LIMIT; }
        if ((rc = e->FK.ensure((nw + 4) * 8))) return rc;
        if ((rc = e->FT.ensure((nw + 4) * sizeof(V)))) return rc;
        hipLaunchKernelGGL((k_range_copy<V, W>), dim3(N), dim3(128), 0, st, d_keys, d_w, d_off, first, cnt, fpos,
                           e->FK.as<u64>(), e->FT.as<V>());
        keys_in = e->FK.as<u64>();
        tags_in = e->FT.as<V>();
    }
    e->n_kept = 0;
    m = 0;
    bool labels_queued = false;   // the label pass was queued before the grouping's counts were read back
    order_queued = false;
    if (nw == 0) return KSP_OK;   // (slice mode only) no key of this range: labels stay the identity
    // sort 1: all entries by the top 32 significant key bits (payload = tag [+weight]):
    // d_keys,VA -> KA,VB; then order the rare mixed runs by the full key (k_fix_runs)
    // (rocPRIM 4.2 / ROCm 7.2 mis-sorts 64-bit keys on any bit range [b > 0, 64) below ~1M items —
    //  found with a stand-alone sweep on MI355X; ranges ending below bit 64 are fine — so keys
    //  that use all 64 bits take the full-width sort.)
    const int shift = (e->full_sort || kbits >= 64) ? 0 : std::max(0, kbits - 32);
    u32* d_ovf = (u32*)(scal + 4);   // set by k_fix_runs when a run is too long; checked at the end of the build
    if (!hand) KSP_HIP(hipMemsetAsync(d_ovf, 0, 8, st));   // (the hand-written partition's build zeroes the scalar block at its start)
    tb = 0;
    // grouping by hash bucket (see k_bucket_group): partition on the top pb key bits only — buckets of
    // 400-800 entries for uniform hashes (up to twice that when the keys span just over half of [0, 2^kbits))
    // (never a bit range that ends at bit 64 — see the rocPRIM note below: bit 63 is left to the buckets,
    //  which compare whole keys; folding the two halves of the key range keeps the buckets even)
    if (topbit > 63) topbit = 63;
    int pb = 0;
    if (hand) { pb = 1; while ((1u << pb) < nb_hand) ++pb; }   // (statistics; the hand-written partition takes any bucket count)
    else if ((phase == 0 || phase == 1) && !e->hash_off && !e->full_sort && nw >= 4096u) {
        pb = 1;
        while ((nw >> pb) > HB_MEAN) ++pb;
        if (pb > topbit) pb = 0;   // (few distinct keys, many holders each: the sort path)
    }
    if ((rc = e->FK.ensure((nw / 2 + 16) * 4))) return rc;
    u32* first = (u32*)e->FK.p;            // first kept entry of every rank (FK: the slice's input keys are dead after sort 1)
    if (pb) {
        const int shiftb = topbit - pb;
        const u32 nbuckets = hand ? nb_hand : 1u << pb;
        // KB is free until the grouping scans: per-entry records, then the bucket tables
        unsigned short* rec = (unsigned short*)e->KB.p;   // (16 bits per partitioned entry; the tables behind keep their place)
        u64* bsum = (u64*)e->KB.p + ((seg ? nslots : nw) / 2 + 1);
        u64* bbase = bsum + nbuckets;
        u32* bstart = (u32*)(bbase + nbuckets);   // nbuckets + 1
        u32* d_hovf = (u32*)(scal + 9);
        // tables of the segment partition for `r1` ranges: boundary table, source groups and 2 048-entry chunks (host side,
        // rebuilt when the offsets or the range count change)
        auto seg_tables = [&](const u32 r1) -> int {
            const size_t tbl_words = (size_t)N * (r1 + 1);
            if ((rc = e->seg_tbl.ensure(tbl_words * 4))) return rc;
            if (!e->seg_groups_ok || e->seg_groups_nb1 != r1) {
                std::vector<u32>& gs = e->seg_groups;
                gs.clear();
                gs.push_back(0);
                const u64 per = (u64)SEG_FILL * r1;
                u64 acc = 0;
                u32 cnt_s = 0;
                for (u32 s_ = 0; s_ < N; ++s_) {
                    const u64 len = e->h_off[s_ + 1] - e->h_off[s_];
                    if (cnt_s && (acc + len > per || cnt_s == SEG_SMAX)) { gs.push_back(s_); acc = 0; cnt_s = 0; }
                    acc += len;
                    ++cnt_s;
                }
                gs.push_back(N);
                std::vector<uint4>& ck = e->seg_chunks;   // chunks of k_seg_bounds: 2 048 consecutive entries of one run each
                ck.clear();
                for (u32 s_ = 0; s_ < N; ++s_) {
                    const u64 b_ = e->h_off[s_], len = e->h_off[s_ + 1] - b_;
                    u64 a = 0;
                    do {
                        const u64 c = std::min<u64>(SEG_BPART, len - a);
                        ck.push_back(make_uint4((u32)(b_ + a), (u32)a, s_, (u32)c | (a + c == len ? 0x80000000u : 0u)));
                        a += c;
                    } while (a < len);
                }
                if ((rc = e->seg_chk.ensure(ck.size() * 16))) return rc;
                KSP_HIP(hipMemcpyAsync(e->seg_chk.p, ck.data(), ck.size() * 16, hipMemcpyHostToDevice, st));
                e->seg_groups_nb1 = r1;
                e->seg_groups_ok = true;
                if ((rc = e->seg_grp.ensure(gs.size() * 4 + 16 + 258 * 8))) return rc;   // (+ the ranges' first keys, k_seg_prep)
                KSP_HIP(hipMemcpyAsync(e->seg_grp.p, gs.data(), gs.size() * 4, hipMemcpyHostToDevice, st));
            }
            return KSP_OK;
        };
        BucketBounds bb{bstart, nullptr, 0u};
        if (seg) {
            // level 1 is already in the sketches (sorted runs): boundaries of every source's segments, then the scatter
            // gathers its tiles from the segments; fixed places per bucket, the cursors count what arrived
            if ((rc = seg_tables(seg_nb1))) return rc;
            const u32 ngroups_s = (u32)e->seg_groups.size() - 1;
            phase_mark(e, st, "partition");
            if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[4], st));
            u64* kmin = (u64*)((char*)e->seg_grp.p + (((e->seg_groups.size() * 4) + 15) & ~(size_t)15));   // (behind the groups)
            hipLaunchKernelGGL(k_seg_prep, dim3(1), dim3(256), 0, st, scal, nbuckets, seg_pb2, seg_nb1, kmin);   // (the multiplier of this build, the ranges' first keys)
            hipLaunchKernelGGL(k_seg_bounds, dim3((u32)e->seg_chunks.size()), dim3(64 * SEG_BW), 0, st, d_keys, e->seg_chk.as<uint4>(), scal, kmin,
                               seg_pb2, nbuckets - 1, seg_nb1, e->seg_tbl.as<u32>());
            hipLaunchKernelGGL((k_seg_scatter<V>), dim3(ngroups_s * seg_nb1), dim3(P2_THREADS), 0, st, d_keys, d_off, e->seg_tbl.as<u32>(),
                               e->seg_grp.as<u32>(), scal, seg_pb2, nbuckets - 1, seg_nb1, seg_cap, hp_gcnt, KA, VB);
            if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[5], st));
            phase_mark(e, st, "bucket grouping");
            bb = BucketBounds{nullptr, hp_gcnt, seg_cap};
        } else if (hand) {
            // partition by bucket = floor(key * nbuckets / (max key + 1)): d_keys -> level-1 pages [-> middle pages] -> KA, VB, bstart
            // three levels: level 1 can be read off the sorted runs too (k_seg_mid), for runs that leave a dozen entries per
            // (source, level-1 bucket)
            bool seg3 = false;
            if (hp_pbm && !e->seg_off && e->h_off.size() == (size_t)N + 1 && N) {
                const char* sv = std::getenv("KSP_SEG");
                seg3 = sv ? std::atoi(sv) != 0 : n / N / hp_nb1 >= SEG_MIN_LEN && seg_runs_fit(hp_nb1);
            }
            seg3_used = seg3;
            if (!seg3) {
                if ((rc = e->PK.ensure(hp_pages_a * P1_PAGE * 8))) return rc;
                if ((rc = e->PT.ensure(hp_pages_a * P1_PAGE * sizeof(V)))) return rc;
                if ((rc = e->PD.ensure(hp_pages_a * P1_PAGE))) return rc;
            } else if ((rc = seg_tables(hp_nb1))) return rc;
            if (hp_pbm) {
                if ((rc = e->PK2.ensure(hp_pages_m * P1_PAGE * 8))) return rc;
                if ((rc = e->PT2.ensure(hp_pages_m * P1_PAGE * sizeof(V)))) return rc;
                if ((rc = e->PD2.ensure(hp_pages_m * P1_PAGE))) return rc;
            }
            phase_mark(e, st, "partition");
            if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[4], st));
            if (seg3) {
                u64* kmin = (u64*)((char*)e->seg_grp.p + (((e->seg_groups.size() * 4) + 15) & ~(size_t)15));
                hipLaunchKernelGGL(k_seg_prep, dim3(1), dim3(256), 0, st, scal, nbuckets, hp_pb2 + hp_pbm, hp_nb1, kmin);
                hipLaunchKernelGGL(k_seg_bounds, dim3((u32)e->seg_chunks.size()), dim3(64 * SEG_BW), 0, st, d_keys, e->seg_chk.as<uint4>(), scal, kmin,
                                   hp_pb2 + hp_pbm, nbuckets - 1, hp_nb1, e->seg_tbl.as<u32>());
                hipLaunchKernelGGL((k_seg_mid<V>), dim3(((u32)e->seg_groups.size() - 1) * hp_nb1), dim3(P2_THREADS), 0, st, d_keys, d_off,
                                   e->seg_tbl.as<u32>(), e->seg_grp.as<u32>(), scal, hp_pb2, hp_pbm, nbuckets - 1, hp_nb1, hp_m, e->PK2.as<u64>(),
                                   e->PT2.as<V>(), e->PD2.as<u8>());
            } else {
            hipLaunchKernelGGL(k_part_src, dim3(grid_for((u64)hp_nchunks + 1, bs)), dim3(bs), 0, st, d_off, N, hp_nchunks, hp_src, scal,
                               nbuckets);
            // (short sources — fewer than two consecutive entries per level-1 bucket on average — are ordered in LDS before
            //  they are written: see k_part1; KSP_DEBUG_PART_SORTED=0 / 1 forces the choice: timing experiments, tests)
            bool p1_sorted = N && n / N < 2ull * hp_nb1;
            if (const char* sv = std::getenv("KSP_DEBUG_PART_SORTED")) p1_sorted = std::atoi(sv) != 0;
            if (p1_sorted)
                hipLaunchKernelGGL((k_part1<V, true>), dim3(hp_nchunks), dim3(P1_THREADS), 0, st, d_keys, d_off, N, (u32)nw, scal,
                                   hp_pb2 + hp_pbm, hp_pb2, nbuckets - 1, hp_a, hp_src, e->PK.as<u64>(), e->PT.as<V>(), e->PD.as<u8>());
            else
            hipLaunchKernelGGL((k_part1<V>), dim3(hp_nchunks), dim3(P1_THREADS), 0, st, d_keys, d_off, N, (u32)nw, scal,
                               hp_pb2 + hp_pbm, hp_pb2, nbuckets - 1, hp_a, hp_src, e->PK.as<u64>(), e->PT.as<V>(), e->PD.as<u8>());
            }
            if (hp_pbm && !seg3)
                hipLaunchKernelGGL((k_part_mid<V>), dim3((u32)hp_pages_a), dim3(P2_THREADS), 0, st, scal, hp_a, hp_m, hp_pb2, hp_pbm,
                                   nbuckets - 1, e->PK.as<u64>(), e->PT.as<V>(), e->PD.as<u8>(), e->PK2.as<u64>(), e->PT2.as<V>(),
                                   e->PD2.as<u8>());
            const PartLists& last = hp_pbm ? hp_m : hp_a;
            const size_t pages_last = hp_pbm ? hp_pages_m : hp_pages_a;
            const u64* Kl = hp_pbm ? e->PK2.as<u64>() : e->PK.as<u64>();
            const V* Tl = hp_pbm ? e->PT2.as<V>() : e->PT.as<V>();
            const u8* Dl = hp_pbm ? e->PD2.as<u8>() : e->PD.as<u8>();
            hipLaunchKernelGGL(k_hist2, dim3((u32)pages_last), dim3(256), 0, st, scal, last, hp_pb2, Dl, hp_gcnt);
            hipLaunchKernelGGL(k_group_base, dim3(1), dim3(1024), 0, st, scal, last, hp_ngroups, (u32)nw, hp_gbase);
            hipLaunchKernelGGL(k_scan2, dim3(hp_ngroups), dim3(256), 0, st, hp_gbase, hp_pb2, nbuckets, hp_ngroups, hp_gcnt, bstart);
            hipLaunchKernelGGL((k_scatter2<V>), dim3((u32)pages_last), dim3(P2_THREADS), 0, st, scal, last, hp_pb2, nbuckets - 1, Kl, Tl,
                               hp_gcnt, KA, VB);
            if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[5], st));
            phase_mark(e, st, "bucket grouping");   // (bsum: zeroed with the rest at the start of the build)
        } else {
        phase_mark(e, st, "partition");
        KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, KA, tags_in, VB, nw, shiftb, topbit, st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[4], st));
        KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, keys_in, KA, tags_in, VB, nw, shiftb, topbit, st));
        if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[5], st));
        phase_mark(e, st, "bucket grouping");
        KSP_HIP(hipMemsetAsync(d_hovf, 0, 8, st));
        KSP_HIP(hipMemsetAsync(bsum, 0, (size_t)nbuckets * 8, st));
        hipLaunchKernelGGL(k_bucket_bounds, dim3(grid_for((u64)nbuckets + 1, bs)), dim3(bs), 0, st, KA, nw, shiftb, nbuckets,
                           bstart);
        }
        e->sort_entries = nw;
        e->sort_bits = pb;
        e->part_kind = (seg || seg3_used) ? 3 : hand ? 2 : 1;
        // ---- the bucket-resident middle of stage 1 (fused_kernels.hip.h): grouping + emit + labels in one kernel, the
        // group records straight to rank order, one read-back.  Unweighted whole builds on the hand-written partition
        // whose blocks fit the split's tables; anything it cannot take (an oversize bucket, sparse sharing) sets
        // fused_off and the build is repeated pass by pass.  Measured on C2 (round 3): correct, 0.6 GB less HBM traffic per
        // step, but 0.14 ms SLOWER than the pass-by-pass kernels (bucket-at-a-time kernels are bound by LDS latency and
        // barriers, not by HBM: DESIGN.md section 5) — so it only runs when KSP_FUSED=1 asks for it (tests run both).
        e->fused_flags = false;
        e->fused_used = 0;
        if constexpr (!W) {
            const char* fv = std::getenv("KSP_FUSED");
            const char* msv = std::getenv("KSP_MS");
            const u32 ms_max = (msv && std::atoi(msv) == 1024) ? MS_MAXB : 256u;
            const u64 Tt = (u64)nb * (nb + 1) / 2;
            if (hand && phase == 0 && reorder && nb <= ms_max && nb <= FK_NB_MAX && Tt <= (1ull << 22) && !e->fused_off && !e->key_groups_off &&
                (fv && std::atoi(fv) == 1) && !(msv && std::atoi(msv) == 0) && e->pre_zeroed_bits && e->pre_zeroed_work) {
                if ((rc = e->biglist.ensure(((size_t)nbuckets + 1) * 4))) return rc;
                if ((rc = e->VA.ensure((nslots + 4) * sizeof(V)))) return rc;
                if ((rc = e->FK.ensure((nslots / 2 + (u64)nbuckets + 16) * 4))) return rc;
                if ((rc = e->mm.ensure((nw / (INLINE_MAX + 1) + 16) * 16))) return rc;   // (a mask per 5 kept entries at most)
                VA = e->VA.as<V>();
                u32* kst = (u32*)e->FK.p;
                if (!e->hb_slots) {
                    int per_cu = 0, cus = 0;
                    KSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bucket_group, HB_THREADS, 0));
                    KSP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
                    e->hb_slots = (u32)std::max(1, per_cu * cus);
                }
                // labels on lines of their own while they are lowered (the head of KB: no per-entry records on this path)
                const u32 skip = label_sampling(e, (u64)((double)nw * e->kept_frac));
                const int ls = e->KB.bytes >= (size_t)N * 128 && (size_t)N * 128 <= (size_t)(nslots / 2) * 8 ? 5 : 0;
                u32* lab = ls ? (u32*)e->KB.p : label;
                if (ls) hipLaunchKernelGGL(k_label_spread, dim3(grid_for(N, bs)), dim3(bs), 0, st, lab, ls, N);
                hipLaunchKernelGGL((k_fgroup<V>), dim3(std::min(nbuckets, e->hb_slots)), dim3(HB_THREADS), 0, st, KA, VB, bb, nbuckets, VA, kst,
                                   bsum, d_hovf, e->biglist.as<u32>(), lab, ls, skip, label_max);
                size_t tb2 = 0;
                KSP_HIP(rocprim::exclusive_scan(nullptr, tb2, bsum, bbase, (u64)0, (size_t)nbuckets, rocprim::plus<u64>(), st));
                if ((rc = e->tmp.ensure(tb2))) return rc;
                KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb2, bsum, bbase, (u64)0, (size_t)nbuckets, rocprim::plus<u64>(), st));
                hipLaunchKernelGGL(k_ftotals, dim3(1), dim3(64), 0, st, bsum, bbase, nbuckets, scal);
                phase_mark(e, st, "source labels + order");
                if (ls) hipLaunchKernelGGL(k_label_gather, dim3(grid_for(N, bs)), dim3(bs), 0, st, lab, ls, label, N);
                {
                    int lbits = 1;
                    while (lbits < 32 && (N >> lbits)) ++lbits;
                    tb = 0;
                    u32* sort_out = e->padded ? sorted_src : order;
                    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, label, labs, iota, sort_out, (size_t)N, 0, lbits, st));
                    if ((rc = e->tmp.ensure(tb))) return rc;
                    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, label, labs, iota, sort_out, (size_t)N, 0, lbits, st));
                    if (e->padded) {
                        hipLaunchKernelGGL(k_pack_blocks, dim3(1), dim3(1024), 0, st, labs, N, nb, blk_src);
                        hipLaunchKernelGGL(k_place_sources, dim3(grid_for((u64)nb * TB, bs)), dim3(bs), 0, st, sorted_src, blk_src, newidx, order,
                                           sbound, e->blk_max.as<u32>(), nb);
                    } else {
                        hipLaunchKernelGGL(k_perm_bound, dim3(grid_for(N, bs)), dim3(bs), 0, st, order, newidx, sbound, e->blk_max.as<u32>(), N);
                    }
                }
                // the (block, key) groups, chunk by chunk, straight into rank order
                phase_mark(e, st, "key groups");
                u32 gb = 8;
                if (const char* gv = std::getenv("KSP_DEBUG_FK_GB")) gb = (u32)std::min<int>(FK_GBMAX, std::max(1, std::atoi(gv)));   // (timing experiments)
                const u32 chunks = grid_for(nbuckets, gb);
                const u32 mb = nb <= 256 ? 256u : 1024u;
                const u64 Kcap = nw;   // records <= kept entries <= entries
                if ((rc = e->gp.ensure((Kcap + 4) * 12))) return rc;
                e->gp_stride = Kcap + 4;
                u64* rec_val = e->gp.as<u64>();
                u32* rec_blk = (u32*)(rec_val + (Kcap + 4));
                if ((rc = e->ms_hist.ensure(((size_t)chunks + 1) * mb * 4 + (size_t)chunks * 4 + 8192))) return rc;
                u32* hist = e->ms_hist.as<u32>();
                u32* tot = hist + (size_t)chunks * mb;
                u32* nrec = tot + mb + 64;
                FkOut fo{rec_blk, rec_val, nrec, hist, e->mm.as<uint4>(), e->dwork.as<unsigned long long>()};
                if (mb == 256)
                    hipLaunchKernelGGL((k_fkeys<V, 256>), dim3(chunks), dim3(FK_THREADS), 0, st, VA, kst, bb, bsum, bbase, nbuckets, gb, newidx, N, nb, fo, scal);
                else
                    hipLaunchKernelGGL((k_fkeys<V, 1024>), dim3(chunks), dim3(FK_THREADS), 0, st, VA, kst, bb, bsum, bbase, nbuckets, gb, newidx, N, nb, fo, scal);
                e->pre_zeroed_work = false;
                phase_mark(e, st, "block lists");
                hipLaunchKernelGGL(k_fms_scan, dim3(nb), dim3(256), 0, st, hist, mb, chunks, scal, tot, blk_raw, blk_pos, nb);
                KSP_HIP(hipMemcpyAsync(e->h_scal, scal, 120, hipMemcpyDeviceToHost, st));   // [0] max key, [1] list words, [2] keys, [6] entries, [9] / [14] overflow (one copy)
                KSP_HIP(hipEventRecord(e->ev_rb, st));
                hipLaunchKernelGGL(k_pad, dim3(nb + 1), dim3(256), 0, st, blk_raw, blk_pos, e->bkeys.as<u32>(), nb, PAD);
                if ((rc = e->pmask.ensure((Kcap + (u64)nb * (WIN + 4) + 4 * WIN) * 16))) return rc;
                const size_t bit_words = (size_t)(((Tt + 63) / 64) * 2 + 2);
                unsigned char* flags = (unsigned char*)e->tbits.p + bit_words * 4;
                if (mb == 256)
                    hipLaunchKernelGGL((k_fms_place<256>), dim3(chunks), dim3(MS_THREADS), 0, st, rec_blk, rec_val, nrec, bbase, gb, hist, blk_pos, nb,
                                       e->bkeys.as<u32>(), e->info.as<u32>(), e->mm.as<uint4>(), e->pmask.as<uint4>(), flags, scal);
                else
                    hipLaunchKernelGGL((k_fms_place<1024>), dim3(chunks), dim3(MS_THREADS), 0, st, rec_blk, rec_val, nrec, bbase, gb, hist, blk_pos, nb,
                                       e->bkeys.as<u32>(), e->info.as<u32>(), e->mm.as<uint4>(), e->pmask.as<uint4>(), flags, scal);
                KSP_HIP(wait_readback(e));
                if ((seg || seg3_used) && ((u32)e->h_scal[PC_OVF] == 4 || (u32)e->h_scal[PC_OVF] == 5)) {
                    e->seg_off = true;
                    e->part_fail = (int)(u32)e->h_scal[PC_OVF];
                    return build_impl<V>(e, d_keys, d_w, st, phase);
                }
                if ((u32)e->h_scal[PC_OVF]) {
                    e->part_off = true;
                    e->part_fail = (int)(u32)e->h_scal[PC_OVF];
                    return build_impl<V>(e, d_keys, d_w, st, phase);
                }
                e->max_key = e->h_scal[0];
                e->have_max_key = true;
                {
                    int bits = 1;
                    while (bits < 64 && (e->max_key >> bits)) ++bits;
                    e->key_bits = bits;
                }
                const char* jm = std::getenv("KSP_JOIN");
                const bool want_matches = (jm && std::string(jm) == "matches") || (!jm && e->h_scal[6] < 4 * e->h_scal[1]);
                if ((u32)(e->h_scal[9] >> 32) || (u32)e->h_scal[9] || want_matches) {   // an oversize bucket / sparse sharing: pass by pass from now on
                    e->fused_off = true;
                    return build_impl<V>(e, d_keys, d_w, st, phase);
                }
                m = e->h_scal[6];
                e->n_kept = m;
                if (nw) e->kept_frac = std::max(0.05, (double)m / (double)nw);
                e->rank1_ok = false;
                e->fused_used = 1;
                if (m == 0) return KSP_OK;
                const u64 K = std::max<u64>(1, e->h_scal[1]);
                {
                    const u64 avg = K / nb + 1;
                    u32 nc = NP;
                    while ((u64)nc * 32 < 4 * avg && nc < e->ncell) nc <<= 1;
                    e->ncell = std::min(e->ncell, nc);
                }
                hipLaunchKernelGGL(k_cidx, dim3(grid_for((u64)nb * (e->ncell + 1), bs)), dim3(bs), 0, st, e->bkeys.as<u32>(),
                                   blk_raw, blk_pos, scal, e->part.as<u32>(), nb, e->ncell);
                KSP_HIP(hipGetLastError());
                e->pmask_on = true;
                e->have_dwork = true;
                e->have_rank_pairs = false;
                e->fused_flags = true;
                e->pre_zeroed_bits = false;
                e->scal_fresh = true;
                return KSP_OK;
            }
        }
        if (!e->hb_slots) {   // persistent workgroups: as many as fit on the device at once
            int per_cu = 0, cus = 0;
            KSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bucket_group, HB_THREADS, 0));
            KSP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
            e->hb_slots = (u32)std::max(1, per_cu * cus);
        }
        // buckets above HB_CAP entries (d_hovf[1] counts them): a slot per bucket, so data whose keys have hundreds of
        // holders each (C3: a third of the buckets at a mean of 2 000) stays on this path instead of falling back to the sort
        if ((rc = e->biglist.ensure(((size_t)nbuckets + 1) * 4))) return rc;
        u32* big_list = e->biglist.as<u32>();
        hipLaunchKernelGGL(k_bucket_group, dim3(std::min(nbuckets, e->hb_slots)), dim3(HB_THREADS), 0, st, KA, bb,
                           nbuckets, (u32)nw, rec, bsum, d_hovf, big_list);
        // (a rank per kept entry only when the sort-by-block fallback is known to follow: the key-by-key build reads crank[])
        u32* rank_out = e->key_groups_off ? rank1 : nullptr;
        e->rank1_ok = rank_out != nullptr;
        hipLaunchKernelGGL((k_bucket_big<V, 0>), dim3(1024), dim3(HB_THREADS), 0, st, KA, VB, bb, big_list, d_hovf,
                           bsum, (const u64*)nullptr, VA, rank_out, first, crank);
        size_t tb2 = 0;
        KSP_HIP(rocprim::exclusive_scan(nullptr, tb2, bsum, bbase, (u64)0, (size_t)nbuckets, rocprim::plus<u64>(), st));
        if ((rc = e->tmp.ensure(tb2))) return rc;
        KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb2, bsum, bbase, (u64)0, (size_t)nbuckets, rocprim::plus<u64>(), st));
        hipLaunchKernelGGL((k_bucket_emit<V>), dim3((nbuckets + HB_EMIT - 1) / HB_EMIT), dim3(HB_THREADS), 0, st, rec, VB, bb,
                           bbase, bsum, nbuckets, VA, rank_out, first, scal, crank);
        hipLaunchKernelGGL((k_bucket_big<V, 1>), dim3(1024), dim3(HB_THREADS), 0, st, KA, VB, bb, big_list, d_hovf,
                           bsum, bbase, VA, rank_out, first, crank);
        // [0] max key, [2] keys, [6] entries, [9] / [14] overflow.  The kept-entry count sizes every later pass — but the label
        // pass can be queued without it (its key count comes from the device, its sampling rate from the kept fraction of
        // the engine's previous build), so the device works on while the host waits for the words
        if (!reorder) KSP_HIP(hipMemcpyAsync(e->h_scal, scal, 120, hipMemcpyDeviceToHost, st));
        if (reorder) {
            phase_mark(e, st, "source labels + order");
            run_label(first, (u32)std::min<u64>(nw / 2 + 1, 0x7FFFFFFFu), (u64)((double)nw * e->kept_frac), scal,
                      ride_readback(e, st, scal, e->h_scal, 15));   // (the words ride along with k_label)
            labels_queued = true;
            if (phase == 0) {   // (... and neither does the order of the sources need it: sort, block cuts and placement are queued too)
                const Stage1<V> so{e, st, phase, n, N, nb, KA, VA, VB, scal, blk_raw, blk_pos, rank1, crank, label, iota, labs, order, newidx,
                                   sbound, sorted_src, blk_src, bbits, reorder, hand_zeroed, 0};
                if ((rc = stage1_source_order(so))) return rc;
                order_queued = true;
            }
            KSP_HIP(wait_readback(e));
        } else {
            KSP_HIP(hipStreamSynchronize(st));
        }
        if (hand) {
            if ((seg || seg3_used) && ((u32)e->h_scal[PC_OVF] == 4 || (u32)e->h_scal[PC_OVF] == 5)) {   // a tile or a bucket of the segment partition overflowed: the paged levels from now on
                e->seg_off = true;
                e->part_fail = (int)(u32)e->h_scal[PC_OVF];
                return build_impl<V>(e, d_keys, d_w, st, phase);
            }
            if ((u32)e->h_scal[PC_OVF]) {   // the page tables could not hold these keys: the library partition from now on
                e->part_off = true;
                e->part_fail = (int)(u32)e->h_scal[PC_OVF];
                return build_impl<V>(e, d_keys, d_w, st, phase);
            }
            e->max_key = e->h_scal[0];
            e->have_max_key = true;
            int bits = 1;
            while (bits < 64 && (e->max_key >> bits)) ++bits;
            e->key_bits = bits;
        }
        if ((u32)e->h_scal[9]) {   // a bucket did not fit (skewed keys): this engine sorts from now on
            e->hash_off = true;
            return build_impl<V>(e, d_keys, d_w, st, phase);
        }
    } else {
    KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, KA, tags_in, VB, nw, shift, kbits, st));
    if ((rc = e->tmp.ensure(tb))) return rc;
    if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[4], st));
    KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, keys_in, KA, tags_in, VB, nw, shift, kbits, st));
    if (e->time_sort) KSP_HIP(hipEventRecord(e->ev[5], st));
    e->sort_entries = nw;
    e->sort_bits = kbits - shift;
    e->part_kind = 1;
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
    {
        auto pf = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u64>(0), PruneFn{KA, nw});
        PruneScatterIt<V> out{{KA, VB, VA, rank1, first, scal, nw}, 0};
        tb = 0;
        KSP_HIP(rocprim::inclusive_scan(nullptr, tb, pf, out, nw, rocprim::plus<u64>(), st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::inclusive_scan(e->tmp.p, tb, pf, out, nw, rocprim::plus<u64>(), st));
    }
    KSP_HIP(hipMemcpyAsync(e->h_scal + 2, scal + 2, 40, hipMemcpyDeviceToHost, st));   // [2] keys ... [6] entries (one copy)
    KSP_HIP(hipStreamSynchronize(st));   // the kept-entry count sizes every later pass
    e->rank1_ok = true;   // (the prune scan writes a rank per kept entry)
    if (e->h_scal[6])
        hipLaunchKernelGGL(k_crank_from_rank, dim3(grid_for(e->h_scal[6] / CR_CHUNK + 1, bs)), dim3(bs), 0, st, rank1, crank, (u32)e->h_scal[6]);
    }
    m = e->h_scal[6];
    e->n_kept = m;
    if (nw) e->kept_frac = std::max(0.05, (double)m / (double)nw);
    if (m == 0 && phase == 0) return KSP_OK;   // no key is shared by two sources: no pair at all
    if (reorder && m && !labels_queued) {
        // label = smallest source id among the holders of a source's shared keys
        phase_mark(e, st, "source labels + order");
        run_label(first, (u32)e->h_scal[2], m);
    }
    if (phase == 1) return KSP_OK;
    }   // phase != 2
    Stage1<V> s1{e, st, phase, n, N, nb, KA, VA, VB, scal, blk_raw, blk_pos, rank1, crank, label, iota, labs, order, newidx, sbound,
                 sorted_src, blk_src, bbits, reorder, hand_zeroed, m};
    if (reorder && !order_queued && (rc = stage1_source_order(s1))) return rc;
    if (m == 0) return KSP_OK;
    e->have_rank_pairs = false;
    e->have_dwork = false;
    if (!e->key_groups_off && m < (1ull << 32) - KG_CHUNK) {
        bool done = false;
        if ((rc = stage1_lists_by_key(s1, done)) || done) return rc;
    }
    return stage1_lists_by_sort(s1);
}

// tag type of a build: 64-bit (weighted), compact 16-bit (unweighted, <= 65536 sources) or canonical 32-bit
static int build_dispatch(ksp_engine* e, const u64* d_keys, const u32* d_w, hipStream_t st, const int phase) {
    if (e->weighted) return build_impl<u64>(e, d_keys, d_w, st, phase);
    if (e->n_sources <= 65536u && !std::getenv("KSP_TAG32")) return build_impl<u16>(e, d_keys, d_w, st, phase);
    return build_impl<u32>(e, d_keys, d_w, st, phase);
}

// host-side bookkeeping once the full block lists sit in the engine's arrays
// per-block maxima and list offsets to pinned host memory, in stream order (before the build's last synchronisation:
// finish_build then reads them without a copy of its own)
static int block_table_regions(ksp_engine* e, CopyList& c) {
    const size_t bytes = ((size_t)e->nb + 1) * 4;
    if (e->h_blk_stage_bytes < 3 * bytes) {
        if (e->h_blk_stage) (void)hipHostFree(e->h_blk_stage);
        e->h_blk_stage = nullptr; e->h_blk_stage_bytes = 0;
        KSP_HIP(hipHostMalloc((void**)&e->h_blk_stage, 3 * bytes + 4096));
        e->h_blk_stage_bytes = 3 * bytes + 4096;
    }
    // [maxima | sources before every block (padded layouts)], then the list offsets
    copy_add(c, e->blk_max.p, e->h_blk_stage, e->padded ? 2 * bytes : bytes);
    copy_add(c, e->blk_raw.p, e->h_blk_stage + 2 * bytes, bytes);
    e->blk_staged = true;
    return KSP_OK;
}
static int stage_block_tables(ksp_engine* e, hipStream_t st) {
    CopyList c{};
    int rc = block_table_regions(e, c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_copy_regions, dim3(16), dim3(256), 0, st, c);
    return KSP_OK;
}

// Last step of stage 1 (single build and assemble alike): the bitmap of block pairs that share a
// key and the pair-update count of every diagonal tile; finish_build turns them into the work list.
static int launch_sched_kernels(ksp_engine* e, hipStream_t st, const bool with_tables, const bool signal) {
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
    const bool fused = e->fused_flags;        // the bucket-resident build wrote the tile flags and the diagonal work with its lists
    const bool ranked = e->have_rank_pairs;   // the key-by-key build left the pairs in rank order: nothing to sort
    if (!ranked && !fused) {
        if ((rc = e->KA.ensure((K + 4) * 8))) return rc;
        if ((rc = e->KB.ensure((K + 4) * 8))) return rc;
    }
    u32 *pr = ranked ? nullptr : (u32*)e->KA.p, *pb = ranked ? nullptr : pr + (K + 4);
    // (gp: K + 4 record values, then the blocks, then the ranks — see build_impl)
    u32 *pr2 = ranked ? e->gp.as<u32>() + 3 * e->gp_stride : (u32*)e->KB.p, *pb2 = ranked ? e->gp.as<u32>() + 2 * e->gp_stride : pr2 + (K + 4);
    unsigned char* flags = (unsigned char*)e->tbits.p + bit_words * 4;
    phase_mark(e, st, "work list");
    if (!e->pre_zeroed_bits && !fused) KSP_HIP(hipMemsetAsync(e->tbits.p, 0, bit_words * 4 + T + 64, st));
    e->pre_zeroed_bits = false;
    if (!fused && !(ranked && e->have_dwork)) {
        KSP_HIP(hipMemsetAsync(e->dwork.p, 0, ((size_t)nb + 2) * 8, st));
        const u32 shares = (u32)std::min<u64>(64, std::max<u64>(1, 2048 / nb));
        hipLaunchKernelGGL(k_list_pairs, dim3(nb, shares), dim3(256), 0, st, e->bkeys.as<u32>(), e->info.as<u32>(),
                           e->mm.as<uint4>(), e->blk_raw.as<u32>(), e->blk_pos.as<u32>(), pr, pb,
                           e->dwork.as<unsigned long long>());
    }
    if (!ranked && !fused) {
        int rbits = 1;
        while (rbits < 32 && (U >> rbits)) ++rbits;
        size_t tb = 0;
        KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, pr, pr2, pb, pb2, (size_t)K, 0, rbits, st));
        if ((rc = e->tmp.ensure(tb))) return rc;
        KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, pr, pr2, pb, pb2, (size_t)K, 0, rbits, st));
    }
    if (!fused) hipLaunchKernelGGL(k_tile_flags, dim3(grid_for(K, 256)), dim3(256), 0, st, pr2, pb2, K, nb, flags);
    const bool direct = T <= (1ull << 20);   // (a larger bitmap goes through device memory: one copy instead of 8-byte stores over PCIe)
    if (!direct) hipLaunchKernelGGL(k_pack_flags, dim3(grid_for(T, 256)), dim3(256), 0, st, flags, T, e->tbits.as<u32>());
    // match records for the join (sparse sharing: few holders per list word — the lists are long and a block pair
    // matches next to nothing of them; KSP_JOIN=matches / search forces the choice)
    e->matches_on = false;
    e->n_matches = 0;
    {
        if (sched_wants_matches(e, K, ranked)) {
            phase_mark(e, st, "match records");
            if ((rc = e->mcnt.ensure((K + 4) * 4))) return rc;
            if ((rc = e->moff.ensure((K + 4) * 8))) return rc;
            hipLaunchKernelGGL(k_match_count, dim3(grid_for(K, 256)), dim3(256), 0, st, pr2, (u32)K, e->mcnt.as<u32>());
            size_t tb = 0;
            KSP_HIP(rocprim::exclusive_scan(nullptr, tb, e->mcnt.as<u32>(), e->moff.as<u64>(), (u64)0, (size_t)K, rocprim::plus<u64>(), st));
            if ((rc = e->tmp.ensure(tb))) return rc;
            KSP_HIP(rocprim::exclusive_scan(e->tmp.p, tb, e->mcnt.as<u32>(), e->moff.as<u64>(), (u64)0, (size_t)K, rocprim::plus<u64>(), st));
            KSP_HIP(hipMemcpyAsync(e->h_scal + 12, e->moff.as<u64>() + (K - 1), 8, hipMemcpyDeviceToHost, st));
            KSP_HIP(hipMemcpyAsync(e->h_scal + 13, e->mcnt.as<u32>() + (K - 1), 4, hipMemcpyDeviceToHost, st));
            KSP_HIP(hipStreamSynchronize(st));   // (the record count sizes the buffers and the sort)
            const u64 M = e->h_scal[12] + (u32)e->h_scal[13];
            if (M > 0 && M < (1ull << 31) && M <= 32 * K) {
                if ((rc = e->mt0.ensure(M * 4)) || (rc = e->mt1.ensure(M * 4)) || (rc = e->mr0.ensure(M * 8)) || (rc = e->mr1.ensure(M * 8))) return rc;
                const u64* rec_val = e->gp.as<u64>();
                hipLaunchKernelGGL(k_match_emit, dim3(grid_for(K, 256)), dim3(256), 0, st, pr2, pb2, rec_val, (u32)K, e->moff.as<u64>(), nb,
                                   e->mt0.as<u32>(), e->mr0.as<u64>());
                int tbits_n = 1;
                while (tbits_n < 32 && (T >> tbits_n)) ++tbits_n;
                tb = 0;
                KSP_HIP(rocprim::radix_sort_pairs(nullptr, tb, e->mt0.as<u32>(), e->mt1.as<u32>(), e->mr0.as<u64>(), e->mr1.as<u64>(), (size_t)M, 0, tbits_n, st));
                if ((rc = e->tmp.ensure(tb))) return rc;
                KSP_HIP(rocprim::radix_sort_pairs(e->tmp.p, tb, e->mt0.as<u32>(), e->mt1.as<u32>(), e->mr0.as<u64>(), e->mr1.as<u64>(), (size_t)M, 0, tbits_n, st));
                e->matches_on = true;
                e->n_matches = M;
            }
        }
    }
    // results to pinned host memory in stream order: the caller's end-of-build synchronisation covers them
    const size_t stage_bytes = bit_words * 4 + ((size_t)nb + 2) * 8;
    if (e->h_stage_bytes < stage_bytes) {
        if (e->h_stage) (void)hipHostFree(e->h_stage);
        e->h_stage = nullptr; e->h_stage_bytes = 0;
        KSP_HIP(hipHostMalloc((void**)&e->h_stage, stage_bytes + stage_bytes / 4 + 4096));
        e->h_stage_bytes = stage_bytes + stage_bytes / 4 + 4096;
    }
    {
        CopyList c{};
        copy_add(c, e->dwork.p, e->h_stage, ((size_t)nb + 2) * 8);
        if (!direct) copy_add(c, e->tbits.p, e->h_stage + ((size_t)nb + 2) * 8, bit_words * 4);
        if (with_tables && (rc = block_table_regions(e, c))) return rc;   // (the per-block maxima and list offsets ride along)
        e->sched_signalled = false;
        if (direct && signal) {   // (the last workgroup signals: no event record behind the launch)
            if ((rc = e->ticket.ensure(64))) return rc;
            if (!e->ticket_zeroed) { KSP_HIP(hipMemsetAsync(e->ticket.p, 0, 64, st)); e->ticket_zeroed = true; }
            e->sched_seq += 1;
            hipLaunchKernelGGL(k_pack_flags_out, dim3(grid_for(T, 256)), dim3(256), 0, st, flags, T, reinterpret_cast<u32*>(e->h_stage + ((size_t)nb + 2) * 8), c,
                               e->ticket.as<u32>(), reinterpret_cast<unsigned long long*>(e->h_count + 7), e->sched_seq);
            e->sched_signalled = true;
        } else if (direct) hipLaunchKernelGGL(k_pack_flags_out, dim3(grid_for(T, 256)), dim3(256), 0, st, flags, T, reinterpret_cast<u32*>(e->h_stage + ((size_t)nb + 2) * 8), c,
                                              (u32*)nullptr, (unsigned long long*)nullptr, 0ull);
        else hipLaunchKernelGGL(k_copy_regions, dim3(64), dim3(256), 0, st, c);
    }
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
    hipError_t err = hipHostMalloc((void**)&e->h_count, 64);
    if (err == hipSuccess) std::memset(e->h_count, 0, 64);   // ([7]: the sequence number of the early work list's copy-out)
    if (err == hipSuccess) err = hipHostMalloc((void**)&e->h_scal, 128);
    if (err == hipSuccess) std::memset(e->h_scal, 0, 128);   // ([15]: the sequence number of k_readback)
    for (int i = 0; i < 6 && err == hipSuccess; ++i) err = hipEventCreate(&e->ev[i]);
    if (err == hipSuccess) err = hipEventCreate(&e->ev_join_done);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_sched, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_rb, hipEventDisableTiming);
    for (int i = 0; i < ksp_engine::kMaxPhase && err == hipSuccess; ++i) err = hipEventCreate(&e->ph_ev[i]);
    if (err != hipSuccess) {
        set_error(std::string("ksp_engine_create: ") + hipGetErrorString(err));
        ksp_engine_destroy(e);   // (frees whatever was created)
        return KSP_E_HIP;
    }
    *out = e;
    return KSP_OK;
}

void ksp_engine_destroy(ksp_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    ksp::Buf* bufs[] = {&e->d_off, &e->KA, &e->KB, &e->VA, &e->VB, &e->R1, &e->FK, &e->FT, &e->asm_small, &e->tmp, &e->bkeys, &e->info,
                        &e->bw, &e->mm, &e->blk_raw, &e->blk_pos, &e->blk_max, &e->part, &e->scalars, &e->count, &e->tailbuf, &e->smap, &e->tbits, &e->dwork, &e->d_act, &e->ticket,
                        &e->d_wg, &e->gp, &e->gm, &e->ms_hist, &e->pmask, &e->crank, &e->PK, &e->PT, &e->PD, &e->parena, &e->PK2, &e->PT2, &e->PD2, &e->seg_tbl, &e->seg_grp, &e->seg_chk, &e->biglist, &e->mcnt, &e->moff, &e->mt0, &e->mt1,
                        &e->mr0, &e->mr1, &e->mstart, &e->stage[0], &e->stage[1]};
    for (auto* b : bufs) b->release();
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
    for (int i = 0; i < 2; ++i) if (e->ev_aux[i]) (void)hipEventDestroy(e->ev_aux[i]);
    for (int i = 0; i < 2; ++i) if (e->ev_copy[i]) (void)hipEventDestroy(e->ev_copy[i]);
    if (e->h_count) (void)hipHostFree(e->h_count);
    if (e->h_scal) (void)hipHostFree(e->h_scal);
    if (e->h_sched) (void)hipHostFree(e->h_sched);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    if (e->h_blk_stage) (void)hipHostFree(e->h_blk_stage);
    for (int i = 0; i < 6; ++i) if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    if (e->ev_join_done) (void)hipEventDestroy(e->ev_join_done);
    if (e->ev_sched) (void)hipEventDestroy(e->ev_sched);
    if (e->ev_rb) (void)hipEventDestroy(e->ev_rb);
    for (int i = 0; i < ksp_engine::kMaxPhase; ++i) if (e->ph_ev[i]) (void)hipEventDestroy(e->ph_ev[i]);
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
    e->st.n_join_workgroups = e->st.n_tiles;   // (dense mode: one workgroup per tile)
    e->st.n_kept_entries = e->n_kept;
    e->st.n_kept_keys = e->h_scal_keys;
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
    u64 diag_cost = 40;
    if (const char* dc = std::getenv("KSP_DEBUG_DIAGCOST")) diag_cost = (u64)std::max(1, std::atoi(dc));   // (timing experiments)
    u64 off_cost = 24;   // (8 — the search alone — left the off-diagonal shares of C2 at twice the time of the diagonal ones: join 0.283 -> 0.246 ms)
    if (const char* oc = std::getenv("KSP_DEBUG_OFFCOST")) off_cost = (u64)std::max(1, std::atoi(oc));   // (timing experiments)
    auto cost_of = [&](u32 I, u32 J) -> u64 {
        if (I == J && !e->weighted) return diag_cost * words_of(I) + 20000;   // bit-sliced: 16 popcounts x 528 patches per 64 keys
        return I == J ? dw[I] / 2 + 10 * words_of(I) + 20000 : off_cost * (words_of(I) + words_of(J)) + 20000;
    };
    // pass 1: active tiles
    u64 total = 0;
    e->act_tid.reserve((size_t)active);
    const char* only = std::getenv("KSP_DEBUG_ONLY");   // timing experiments: "diag" / "off" (results are incomplete)
    const bool skip_diag = only && std::string(only) == "off", skip_off = only && std::string(only) == "diag";
    for (u32 I = 0; I < nb; ++I) {
        const u64 row = tile_row_start(I, nb);
        if (dw[I] && !skip_diag) e->act_tid.push_back(row);
        for (u64 t = row + 1; !skip_off && t < row + (nb - I);) {
            const u32 wrd = bits[t >> 5] >> (t & 31);
            if (!wrd) { t = (t | 31) + 1; continue; }
            const u64 tt = t + (u64)__builtin_ctz(wrd);
            if (tt >= row + (nb - I)) break;
            e->act_tid.push_back(tt);
            t = tt + 1;
        }
    }
    const size_t A = e->act_tid.size();
    // match-list mode: the records of every tile (their number is the tile's work: a few heavy tiles hold most of them)
    std::vector<u32>& ms = e->h_mstart;
    ms.clear();
    if (e->matches_on) {
        std::vector<u32> tids(A + 1, 0);
        for (size_t i = 0; i < A; ++i) tids[i] = (u32)e->act_tid[i];
        if ((rc = e->mstart.ensure((A + 2) * 4)) || (rc = e->d_act.ensure((4 * (A + 1)) * 4))) return rc;
        ms.resize(A + 1);
        KSP_HIP(hipMemcpyAsync(e->d_act.p, tids.data(), (A + 1) * 4, hipMemcpyHostToDevice, e->sched_stream));   // (d_act: scratch until the records go up)
        hipLaunchKernelGGL(k_match_bounds, dim3(grid_for(A + 1, 256)), dim3(256), 0, e->sched_stream, e->d_act.as<u32>(), (u32)A,
                           e->mt1.as<u32>(), (u32)e->n_matches, e->mstart.as<u32>());
        KSP_HIP(hipMemcpyAsync(ms.data(), e->mstart.p, (A + 1) * 4, hipMemcpyDeviceToHost, e->sched_stream));
        KSP_HIP(hipStreamSynchronize(e->sched_stream));
    }
    std::vector<u64> cost(A);
    for (size_t i = 0; i < A; ++i) {
        u32 I, J;
        tile_decode(e->act_tid[i], nb, I, J);
        cost[i] = (e->matches_on && I != J) ? 6 * (u64)(ms[i + 1] - ms[i]) + 20000 : cost_of(I, J);
        total += cost[i];
    }
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
    e->act32.assign(A, 0);
    u32 nsplit = 0;
    for (size_t i = 0; i < A; ++i) {
        u32 I, J;
        tile_decode(e->act_tid[i], nb, I, J);
        u64 sp = (cost[i] + target - 1) / target;
        sp = std::min<u64>(std::max<u64>(sp, 1), 32);
        // a share that counts fewer than 2^16 keys keeps 16-bit counters whatever its blocks hold (k_join: the shares of a
        // tile meet in the tile's 32-bit buffer): the diagonal tile of a block with a source of >= 2^16 k-mers is cut finer
        if (I == J && !e->weighted && e->h_blk_max[I] >= 65536u && words_of(I) >= 65536u && !std::getenv("KSP_DEBUG_NO16CUT"))
            sp = std::max<u64>(sp, std::min<u64>(words_of(I) / 65535u + 1, 256));
        if (e->matches_on && I != J) {
            // a share of records, not of estimated cycles: a round of 512 records is a chain of memory round trips,
            // and a few tiles hold most of the records (the blocks of the largest sketches) — as one workgroup each
            // they were the whole join (C4: 16 of 19 ms)
            u64 per_share = std::min<u64>(std::max<u64>(e->n_matches / ((u64)e->slots * 4) + 1, 8192), 1u << 20);
            if (e->h_blk_max[I] >= 65536u && e->h_blk_max[J] >= 65536u && !std::getenv("KSP_DEBUG_NO16CUT")) per_share = std::min<u64>(per_share, 65535);   // (16-bit counters: above)
            sp = std::min<u64>(std::max<u64>(((u64)(ms[i + 1] - ms[i]) + per_share - 1) / per_share, 1), 1024);
        }
        {   // does any share of this tile count in 32 bits?  (the rule of k_join, with the largest share of the tile)
            u64 lim = std::min(e->h_blk_max[I], e->h_blk_max[J]);
            if (!e->weighted) {
                const u64 mine = (e->matches_on && I != J) ? ((u64)(ms[i + 1] - ms[i]) + sp - 1) / sp
                               : I == J                    ? (words_of(I) + sp - 1) / sp
                                                           : std::min(words_of(I), words_of(J));
                lim = std::min(lim, mine);
            }
            e->act32[i] = lim >= 65536u;
        }
        u32* r = &e->act_rec[4 * i];
        r[0] = I; r[1] = J; r[2] = (u32)wg.size(); r[3] = nsplit;
        if (sp > 1) ++nsplit;
        for (u64 q = 0; q < sp; ++q) wg.push_back((u32)i);
        if (wg.size() > 0x7FFFFFF0ull) return KSP_OK;   // (absurdly many shares: stay dense)
    }
    u32* r = &e->act_rec[4 * A];
    r[0] = 0; r[1] = 0; r[2] = (u32)wg.size(); r[3] = nsplit;
    // a work list of a few KB stays in pinned host memory and the join's workgroups read their record from there (two
    // uploads = two more dispatches between the build and the join otherwise); larger ones go to the device
    e->sched_in_host = false;
    const size_t sched_words = e->act_rec.size() + wg.size();
    if (!e->matches_on && sched_words <= 16384) {
        if (e->h_sched_words < sched_words) {
            if (e->h_sched) (void)hipHostFree(e->h_sched);
            e->h_sched = nullptr; e->h_sched_words = 0;
            KSP_HIP(hipHostMalloc((void**)&e->h_sched, 16384 * 4));
            e->h_sched_words = 16384;
        }
        std::memcpy(e->h_sched, e->act_rec.data(), e->act_rec.size() * 4);
        if (!wg.empty()) std::memcpy(e->h_sched + e->act_rec.size(), wg.data(), wg.size() * 4);
        e->sched_in_host = true;
    } else {
        if ((rc = e->d_act.ensure(e->act_rec.size() * 4))) return rc;
        if ((rc = e->d_wg.ensure(std::max<size_t>(1, wg.size()) * 4))) return rc;
        KSP_HIP(hipMemcpyAsync(e->d_act.p, e->act_rec.data(), e->act_rec.size() * 4, hipMemcpyHostToDevice, e->sched_stream));
        if (!wg.empty()) KSP_HIP(hipMemcpyAsync(e->d_wg.p, wg.data(), wg.size() * 4, hipMemcpyHostToDevice, e->sched_stream));
    }
    e->sched_on = true;
    e->st.n_active_tiles = A;
    e->st.n_match_records = e->matches_on ? e->n_matches : 0;
    e->st.n_kept_entries = e->n_kept;
    e->st.n_kept_keys = e->h_scal_keys;
    e->st.n_join_workgroups = wg.size();
    return KSP_OK;
}

static int finish_build(ksp_engine* e) {
    e->st.key_bits = e->key_bits;
    if (e->blk_staged) {
        const size_t bytes = ((size_t)e->nb + 1) * 4;
        e->h_blk_off.resize((size_t)e->nb + 1);
        std::memcpy(e->h_blk_max.data(), e->h_blk_stage, bytes);
        std::memcpy(e->h_blk_off.data(), e->h_blk_stage + 2 * bytes, bytes);
        if (e->padded) std::memcpy(e->h_blk_src.data(), e->h_blk_stage + bytes, bytes);
        e->blk_staged = false;
        u32 big_blocks = 0;
        for (u32 b = 0; b < e->nb; ++b) big_blocks += e->h_blk_max[b] >= 65536u;
        e->need32 = big_blocks >= 1;
        e->st.n_block_keys = e->h_blk_off[e->nb];
        int rc = build_schedule(e);
        if (rc) return rc;
        e->built = true;
        return KSP_OK;
    }
    KSP_HIP(hipMemcpy(e->h_blk_max.data(), e->blk_max.p, ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToHost));
    if (e->padded)
        KSP_HIP(hipMemcpy(e->h_blk_src.data(), e->blk_max.as<u32>() + ((size_t)e->nb + 1), ((size_t)e->nb + 1) * 4, hipMemcpyDeviceToHost));
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
    e->post_slice = false;
    e->sched_on = false; e->collect = false; e->have_bits = false; e->matches_on = false; e->pmask_on = false;   // (nothing of the previous build's work list survives)
    e->act_tid.clear(); e->act_rec.clear();
    for (u32 s = 0; s < n_sources; ++s)
        if (h_offsets[s + 1] < h_offsets[s]) { set_error("build: offsets not monotone"); return KSP_E_ARG; }
    const u64 n = n_sources ? h_offsets[n_sources] - h_offsets[0] : 0;
    if (n_sources && h_offsets[0] != 0) { set_error("build: offsets[0] must be 0"); return KSP_E_ARG; }
    // 32-bit entry positions: a whole build takes fewer than 2^30 entries; a key-range slice of a larger set is
    // fine as long as the slice itself stays below (checked once its size is known) — ksp_pairwise_host cuts such
    // sets into enough slices by itself
    if (n >= (1ull << 30) && !(slice && nparts > 1)) { set_error("build: more than 2^30 key entries per call"); return KSP_E_LIMIT; }
    if (n && !d_keys) { set_error("build: d_keys is NULL"); return KSP_E_ARG; }
    if (const char* ro = std::getenv("KSP_REORDER")) e->reorder = std::atoi(ro) != 0;   // diagnostic / tests
    e->n_sources = n_sources;
    e->n_entries = n;
    e->nb = blocks_for(n_sources, e->reorder);
    e->padded = e->nb > (n_sources + TB - 1) / TB;
    e->weighted = d_weights != nullptr;
    e->key_bits = key_bits;
    e->have_max_key = false;
    e->nparts = nparts;
    e->part_id = part;
    const bool same_offsets = e->d_off_sketch && e->h_off.size() == (size_t)n_sources + 1 &&
                              std::memcmp(e->h_off.data(), h_offsets, ((size_t)n_sources + 1) * 8) == 0;
    if (!same_offsets) { e->h_off.assign(h_offsets, h_offsets + n_sources + 1); e->d_off_sketch = false; e->seg_groups_ok = false; }
    e->blk_staged = false;
    if (std::getenv("KSP_FULL_SORT")) e->full_sort = true;   // diagnostic: sort on all key bits
    if (const char* hg = std::getenv("KSP_HASH_GROUP")) e->hash_off = std::atoi(hg) == 0;   // diagnostic / tests
    if (const char* kg = std::getenv("KSP_KEY_GROUPS")) e->key_groups_off = std::atoi(kg) == 0;
    if (const char* pp = std::getenv("KSP_PARTITION")) { e->part_off = std::string(pp) == "rocprim"; e->part_fail = 0; }   // diagnostic / tests
    if (const char* pm = std::getenv("KSP_PART_MIN")) e->part_min = (u32)std::max(1, std::atoi(pm));
    e->ph_n = 0;
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
    e->h_blk_src.resize((size_t)e->nb + 1);
    for (u32 b = 0; b <= e->nb; ++b) e->h_blk_src[b] = (u32)std::min<u64>(n_sources, (u64)b * TB);   // (until a padded build reports its own)
    std::memset(e->slice_hdr, 0, sizeof e->slice_hdr);
    if (n == 0 || e->nb == 0) return KSP_OK;   // nothing can intersect
    int rc;
    if ((rc = e->d_off.ensure(((size_t)n_sources + 1) * 8))) return rc;
    KSP_HIP(resolve_build_ms(e));   // (the previous build's time, before its events are recorded again)
    if (!e->lean) KSP_HIP(hipEventRecord(e->ev[0], st));
    if (!same_offsets) {   // (the engine's own copy is the source: the caller's array may go away before the copy has run)
        KSP_HIP(hipMemcpyAsync(e->d_off.p, e->h_off.data(), ((size_t)n_sources + 1) * 8, hipMemcpyHostToDevice, st));
        e->d_off_sketch = true;
    }
    {   // fine cells: ~32 entries of the largest block per cell, power of two, index kept below 1 GiB
        u64 dmax = 0;
        for (u32 b = 0; (u64)b * TB < n_sources; ++b) {
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
    if ((rc = e->blk_max.ensure((2 * (size_t)e->nb + 4) * 4))) return rc;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int phase = slice ? 1 : 0;   // a slice stops at the source labels (ksp_engine_slice_finish does the rest)
        e->scal_fresh = false;
        e->sched_early = false;
        rc = build_dispatch(e, d_keys, d_weights, st, phase);
        if (rc) return rc;
        if (!e->scal_fresh) {   // (the key-by-key list build has read [1] .. [11] back already: nothing changes them after)
            KSP_HIP(hipMemcpyAsync(e->h_scal + 1, e->scalars.as<u64>() + 1, 64, hipMemcpyDeviceToHost, st));
            KSP_HIP(hipStreamSynchronize(st));
        }
        if ((u32)e->h_scal[4] == 0) break;
        // pathological key distribution (thousands of distinct keys share their top 32 bits):
        // redo with a full-width sort and remember it for later builds on this engine
        e->full_sort = true;
    }
    if (!e->sched_early) e->have_bits = false;
    if (!slice && e->n_kept && !e->sched_early) {   // the work list of the join (slices: after the assemble)
        e->h_scal_words = e->h_scal[1];
        e->h_scal_keys = e->h_scal[2];
        if ((rc = launch_sched_kernels(e, st, true))) return rc;
        if (!e->blk_staged && (rc = stage_block_tables(e, st))) return rc;
    }
    const bool flagged = e->sched_early && e->sched_signalled;
    if (!(e->lean && flagged)) KSP_HIP(hipEventRecord(e->ev[1], st));   // (lean + signalled: nobody looks at the event)
    // (polling the event instead of a blocking wait: the join cannot be cut into shares before the build's tables have
    //  landed, and waking a blocked host thread is tens of microseconds of device idle time per step)
    {
        if (flagged) {   // (the copy-out's last workgroup wrote the number: no event in the stream)
            volatile unsigned long long* f = reinterpret_cast<volatile unsigned long long*>(e->h_count + 7);
            for (unsigned long long spins = 1; *f != e->sched_seq; ++spins) {
                if ((spins & 0xFFFFF) == 0) {   // (now and then: is the stream still alive?)
                    const hipError_t q = hipStreamQuery(st);
                    if (q == hipSuccess && *f != e->sched_seq) KSP_HIP(hipErrorUnknown);
                    if (q != hipErrorNotReady && q != hipSuccess) KSP_HIP(q);
                }
            }
        } else {
        const hipEvent_t landed = e->sched_early ? e->ev_sched : e->ev[1];   // (early work list: the placement pass is still running)
        hipError_t qe;
        while ((qe = hipEventQuery(landed)) == hipErrorNotReady) {}
        KSP_HIP(qe);
        }
    }
    if (e->lean) {
        e->st.ms_build = 0;           // (no events were recorded)
    } else if (e->sched_early) {
        e->build_ms_pending = true;   // (read off the events by whoever asks first: ksp_engine_get_stats, the join's wait, the next build)
        e->st.ms_build = 0;
    } else {
        KSP_HIP(hipEventElapsedTime(&e->st.ms_build, e->ev[0], e->ev[1]));
        phase_close(e, e->ev[1]);
    }
    e->st.key_bits = e->key_bits;
    e->st.ms_sort = 0;
    e->st.sort_entries = e->sort_entries;
    e->st.sort_bits = e->sort_bits;
    e->st.partition_kind = e->sort_entries ? e->part_kind : 0;
    e->st.partition_fallback = e->part_fail;
    e->st.stage1_kind = e->fused_used;
    if (e->sort_entries && e->time_sort) KSP_HIP(hipEventElapsedTime(&e->st.ms_sort, e->ev[4], e->ev[5]));
    return KSP_OK;
}

static int build_postings_common(ksp_engine* e, const uint64_t* h_key_off, const uint32_t* d_sources, const uint32_t* d_key_weights,
                                 uint32_t n_keys, uint32_t n_sources, void* stream, const bool slice) {
    if (!e || (n_keys && (!h_key_off || !d_sources))) { set_error("build_postings: NULL argument"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    e->built = false;
    e->slice_ready = false;
    e->sched_on = false; e->collect = false; e->have_bits = false; e->matches_on = false; e->pmask_on = false;
    e->act_tid.clear(); e->act_rec.clear();
    e->ph_n = 0;
    e->slice_phase = 0;
    e->post_slice = false;
    const u64 n = n_keys ? h_key_off[n_keys] : 0;
    if (n_keys && h_key_off[0] != 0) { set_error("build_postings: key_off[0] must be 0"); return KSP_E_ARG; }
    for (u32 k = 0; k < n_keys; ++k)
        if (h_key_off[k + 1] < h_key_off[k] + 2) { set_error("build_postings: every key needs at least two holders"); return KSP_E_ARG; }
    if (n >= (1ull << 30)) { set_error("build_postings: more than 2^30 entries per call"); return KSP_E_LIMIT; }
    if (const char* ro = std::getenv("KSP_REORDER")) e->reorder = std::atoi(ro) != 0;
    e->n_sources = n_sources;
    e->n_entries = n;
    e->nb = blocks_for(n_sources, e->reorder);
    e->padded = e->nb > (n_sources + TB - 1) / TB;
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
    e->h_blk_src.resize((size_t)e->nb + 1);
    for (u32 b = 0; b <= e->nb; ++b) e->h_blk_src[b] = (u32)std::min<u64>(n_sources, (u64)b * TB);
    e->have_bits = false;
    std::memset(e->slice_hdr, 0, sizeof e->slice_hdr);
    if (n == 0 || e->nb == 0) {   // nothing can intersect
        e->st.n_block_keys = 0;
        e->need32 = false;
        e->built = !slice;
        if (slice) e->slice_phase = 1;
        return KSP_OK;
    }
    int rc;
    KSP_HIP(resolve_build_ms(e));   // (the previous build's time, before its events are recorded again)
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
    if ((rc = e->blk_max.ensure((2 * (size_t)e->nb + 4) * 4))) return rc;
    // key offsets as 32-bit device array (= first entry of every rank)
    if ((rc = e->d_off.ensure(((size_t)n_keys + 2) * 4))) return rc;
    {
        std::vector<u32> off32((size_t)n_keys + 1);
        for (u32 k = 0; k <= n_keys; ++k) off32[k] = (u32)h_key_off[k];
        KSP_HIP(hipMemcpyAsync(e->d_off.p, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, st));
        KSP_HIP(hipStreamSynchronize(st));   // (off32 is a local)
    }
    e->d_off_sketch = false;   // (d_off now holds the key offsets)
    e->post_off = e->d_off.as<u32>();
    e->post_src = d_sources;
    e->post_w = d_key_weights;
    e->post_nkeys = n_keys;
    rc = build_dispatch(e, nullptr, nullptr, st, slice ? 4 : 3);
    e->post_src = e->post_w = nullptr;
    if (!slice) e->post_off = nullptr;
    if (rc) return rc;
    if (slice) {   // (a slice: up to the source labels; ksp_engine_slice_finish builds its lists once all slices' labels are combined)
        KSP_HIP(hipMemcpyAsync(e->h_scal + 4, e->scalars.as<u64>() + 4, 8, hipMemcpyDeviceToHost, st));
        KSP_HIP(hipEventRecord(e->ev[1], st));
        KSP_HIP(hipStreamSynchronize(st));
        if ((u32)e->h_scal[4]) { set_error("build_postings: a source index is >= n_sources"); return KSP_E_ARG; }
        e->h_scal[4] = 0;
        KSP_HIP(hipEventElapsedTime(&e->st.ms_build, e->ev[0], e->ev[1]));
        e->post_slice = true;
        e->slice_phase = 1;
        return KSP_OK;
    }
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
    phase_close(e, e->ev[1]);
    return finish_build(e);
}

int ksp_engine_build_postings(ksp_engine* e, const uint64_t* h_key_off, const uint32_t* d_sources,
                              const uint32_t* d_key_weights, uint32_t n_keys, uint32_t n_sources, void* stream) {
    return build_postings_common(e, h_key_off, d_sources, d_key_weights, n_keys, n_sources, stream, false);
}
// One slice of an inverted index — any subset of its keys, every key with ALL its holders (so the pruning of stage 1 has
// nothing to do and every slice's ranks are its own key order) — up to the source labels; then exactly the calls of a
// key-range slice of sketches: ksp_engine_slice_labels, (MIN over the slices), ksp_engine_slice_finish, _sizes, _export,
// ksp_engine_assemble.  This is how an index of 2^30 memberships or more goes through (slices built in turn or on several
// GPUs), and how the devices of $KSPIDER_DEVICES share stage 1 of the reference's own entry point.
int ksp_engine_build_postings_slice(ksp_engine* e, const uint64_t* h_key_off, const uint32_t* d_sources,
                                    const uint32_t* d_key_weights, uint32_t n_keys, uint32_t n_sources, void* stream) {
    return build_postings_common(e, h_key_off, d_sources, d_key_weights, n_keys, n_sources, stream, true);
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
    KSP_HIP(resolve_build_ms(e));   // (the previous build's time, before its events are recorded again)
    KSP_HIP(hipEventRecord(e->ev[0], st));
    if (d_labels) KSP_HIP(hipMemcpyAsync(label_array(e), d_labels, (size_t)e->n_sources * 4, hipMemcpyDeviceToDevice, st));
    rc = build_dispatch(e, nullptr, nullptr, st, 2);
    if (rc) return rc;
    KSP_HIP(hipMemcpyAsync(e->h_scal + 1, e->scalars.as<u64>() + 1, 64, hipMemcpyDeviceToHost, st));
    if (e->n_kept == 0) {
        // empty slice: valid (all-pad) lists so that export / assemble need no special case
        if ((rc = e->blk_raw.ensure(((size_t)e->nb + 2) * 4))) return rc;
        KSP_HIP(hipMemsetAsync(e->blk_raw.p, 0, ((size_t)e->nb + 2) * 4, st));
        hipLaunchKernelGGL(k_blk_pos, dim3(1), dim3(1024), 0, st, e->blk_raw.as<u32>(), e->blk_pos.as<u32>(),
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
    e->pmask_on = false;   // (the assembled lists have no positional masks)
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
    KSP_HIP(resolve_build_ms(e));   // (the previous build's time, before its events are recorded again)
    KSP_HIP(hipEventRecord(e->ev[0], st));
    KSP_HIP(hipMemcpyAsync(e->asm_small.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
    u64* scal = e->scalars.as<u64>();
    hipLaunchKernelGGL(k_asm_counts, dim3(1), dim3(64), 0, st, d_blk_raw_all, nb + 1, nparts, nb, e->blk_raw.as<u32>());
    hipLaunchKernelGGL(k_blk_pos, dim3(1), dim3(1024), 0, st, e->blk_raw.as<u32>(), e->blk_pos.as<u32>(), scal, nb);
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
    e->have_rank_pairs = false;   // (the assembled lists: pairs from the lists themselves)
    e->have_dwork = false;
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
    const std::vector<u32>& P = e->h_blk_src;   // sources in the blocks before block b
    u64 t = t0;
    while (t < t1) {
        u32 I, J;
        tile_decode(t, e->nb, I, J);
        // rest of row I inside [t, t1)
        u64 row_end = tile_row_start((u64)I + 1, e->nb);
        u64 stop = std::min(row_end, t1);
        const u64 nI = P[I + 1] - P[I];
        // tiles (I, J .. J + cnt - 1) in closed form: the diagonal one, then the sources of the other blocks
        u64 cnt = stop - t, J0 = J;
        if (J0 == I) { pairs += nI * (nI ? nI - 1 : 0) / 2; ++J0; --cnt; }
        if (cnt) pairs += nI * (u64)(P[J0 + cnt] - P[J0]);
        t = stop;
    }
    return pairs;
}

uint64_t ksp_engine_edge_bound(const ksp_engine* e, uint64_t t0, uint64_t t1) {
    if (!e || !e->nb || !e->built) return 0;
    if (!e->sched_on) return ksp_engine_tile_pairs(e, t0, t1);
    const size_t a0 = std::lower_bound(e->act_tid.begin(), e->act_tid.end(), t0) - e->act_tid.begin();
    const size_t a1 = std::lower_bound(e->act_tid.begin(), e->act_tid.end(), t1) - e->act_tid.begin();
    const std::vector<u32>& P = e->h_blk_src;
    u64 pairs = 0;
    for (size_t i = a0; i < a1; ++i) {
        const u32 I = e->act_rec[4 * i], J = e->act_rec[4 * i + 1];
        const u64 nI = P[I + 1] - P[I], nJ = P[J + 1] - P[J];
        pairs += (I == J) ? nI * (nI ? nI - 1 : 0) / 2 : nI * nJ;
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

int ksp_engine_join_launch(ksp_engine* e, uint64_t tile_begin, uint64_t tile_end, ksp_edge* d_edges, uint64_t capacity, void* stream) {
    if (!e) { set_error("join: NULL argument"); return KSP_E_ARG; }
    if (!e->built) { set_error("join: build_blocks has not been run"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    const u64 T = ksp_engine_num_tiles(e);
    if (tile_end > T) tile_end = T;
    e->join_pending = false;
    e->jst = ksp_stats{};
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
    a.pmask = e->pmask_on ? e->pmask.as<uint4>() : nullptr;
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
    ZeroList zj{};
    zero_add(zj, a.out_count, 8);
    if (!e->lean) KSP_HIP(hipEventRecord(e->ev[2], st));
    dim3 block(JW * 64);
    a.blk_max = e->blk_max.as<u32>();
    a.inv = e->smap.as<u32>() + 3 * smap_stride(e);
    a.sched = nullptr; a.act = nullptr; a.wg0 = 0; a.split0 = 0; a.tail_done = nullptr;
    a.collect = e->collect ? 1u : 0u;
    a.mrec = nullptr; a.mstart = nullptr;
#ifdef KSP_WGTIME
    ksp::Buf& wgt_buf = e->wgt_buf;   // (timing builds only, tools/wg_times.py)
    a.wgt = nullptr;
    const char* wgt_file = std::getenv("KSP_WGTIME_FILE");
    size_t& wgt_n = e->wgt_n;
    wgt_n = 0;
#endif
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
        a.sched = e->sched_in_host ? e->h_sched + e->act_rec.size() : e->d_wg.as<u32>();
        a.act = e->sched_in_host ? e->h_sched : e->d_act.as<u32>();
        if (e->matches_on) { a.mrec = e->mr1.as<u64>(); a.mstart = e->mstart.as<u32>(); }
        a.split0 = spA;
        a.n_normal = 0; a.tail_sp = 1;
        a.tailbuf = nullptr;
        const u32 nsplit = spB - spA;
        if (nsplit) {
            if ((rc = e->tailbuf.ensure(((size_t)nsplit * TB * TB + nsplit + 16) * 4))) return rc;
            a.tailbuf = e->tailbuf.as<u32>();
            a.tail_done = a.tailbuf + (size_t)nsplit * TB * TB;
            if (((size_t)nsplit * TB * TB + nsplit + 16) < (1ull << 30)) zero_add(zj, a.tailbuf, ((size_t)nsplit * TB * TB + nsplit + 16) * 4);
            else KSP_HIP(hipMemsetAsync(a.tailbuf, 0, ((size_t)nsplit * TB * TB + nsplit + 16) * 4, st));
        }
        hipLaunchKernelGGL(k_zero_regions, dim3(256), dim3(256), 0, st, zj);   // (the edge counter and the split tiles' buffers: one launch)
        zj.n = 0;
#ifdef KSP_WGTIME
        if (wgt_file) {
            wgt_n = wgB;
            if ((rc = wgt_buf.ensure(wgt_n * 128))) return rc;
            KSP_HIP(hipMemsetAsync(wgt_buf.p, 0, wgt_n * 128, st));
            a.wgt = wgt_buf.as<unsigned long long>();
        }
#endif
        // Two passes over the work list when some tile pairs two blocks that both hold a source of >= 2^16 k-mers: the
        // tiles with packed 16-bit counters, then those with 32-bit counters (64 KB of LDS: two workgroups per CU).  The
        // second pass runs on a stream of its own BESIDE the first — its few fat workgroups leave most of every CU's
        // wave slots free (metagenome bins: 9.2 + 4.3 ms one after the other).
        // (... and not at all when every share of the range stays below 2^16: build_schedule knows every share's bound)
        bool pass32 = false;
        for (size_t i = act0; e->need32 && !pass32 && i < act1; ++i) pass32 = e->act32[i] != 0;
        bool two_streams = pass32 && !std::getenv("KSP_DEBUG_ONE_STREAM");
        if (two_streams) {
            if (!e->aux_stream && hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking) != hipSuccess) two_streams = false;
            for (int i = 0; i < 2 && two_streams; ++i)
                if (!e->ev_aux[i] && hipEventCreateWithFlags(&e->ev_aux[i], hipEventDisableTiming) != hipSuccess) two_streams = false;
        }
        if (two_streams) {
            KSP_HIP(hipEventRecord(e->ev_aux[0], st));                    // (the zeroing launch above)
            KSP_HIP(hipStreamWaitEvent(e->aux_stream, e->ev_aux[0], 0));
        }
        for (int pass = 0; pass < (pass32 ? 2 : 1); ++pass) {
            hipStream_t ps = (pass == 1 && two_streams) ? e->aux_stream : st;
            for (u64 w = wgA; w < wgB; w += kMaxTilesPerLaunch) {
                a.wg0 = (u32)w;
                const dim3 grid((u32)std::min<u64>(kMaxTilesPerLaunch, wgB - w));
                const bool c16 = pass == 0;
                if (e->use_cells) {
                    if (e->weighted) { if (c16) hipLaunchKernelGGL((k_join<true, true, true>), grid, block, 0, ps, a); else hipLaunchKernelGGL((k_join<true, false, true>), grid, block, 0, ps, a); }
                    else { if (c16) hipLaunchKernelGGL((k_join<false, true, true>), grid, block, 0, ps, a); else hipLaunchKernelGGL((k_join<false, false, true>), grid, block, 0, ps, a); }
                } else {
                    if (e->weighted) { if (c16) hipLaunchKernelGGL((k_join<true, true, false>), grid, block, 0, ps, a); else hipLaunchKernelGGL((k_join<true, false, false>), grid, block, 0, ps, a); }
                    else { if (c16) hipLaunchKernelGGL((k_join<false, true, false>), grid, block, 0, ps, a); else hipLaunchKernelGGL((k_join<false, false, false>), grid, block, 0, ps, a); }
                }
            }
        }
        if (two_streams) {
            KSP_HIP(hipEventRecord(e->ev_aux[1], e->aux_stream));
            KSP_HIP(hipStreamWaitEvent(st, e->ev_aux[1], 0));             // (the count is read behind both passes)
        }
        KSP_HIP(hipGetLastError());
    }
    if (zj.n) { hipLaunchKernelGGL(k_zero_regions, dim3(1), dim3(64), 0, st, zj); zj.n = 0; }   // (dense mode: the edge counter)
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
    e->join_flag = false;
    if (e->lean) {   // the count and a sequence number behind it, by a kernel of ours: no event in the stream
        e->join_seq += 1;
        e->join_flag = true;
        e->rb_stream = st;
        hipLaunchKernelGGL(k_readback, dim3(1), dim3(64), 0, st, reinterpret_cast<const u64*>(a.out_count), reinterpret_cast<u64*>(e->h_count), 1u,
                           e->h_count + 6, e->join_seq);
    } else {
        KSP_HIP(hipEventRecord(e->ev[3], st));
        KSP_HIP(hipMemcpyAsync(e->h_count, a.out_count, 8, hipMemcpyDeviceToHost, st));
        KSP_HIP(hipEventRecord(e->ev_join_done, st));
    }
    // what ksp_engine_join_wait reports besides the count: known now (a build may start on this engine before the wait)
    e->jst.last_tiles = tile_end - tile_begin;
    e->jst.last_active_tiles = e->sched_on ? (u64)(act1 - act0) : tile_end - tile_begin;
    e->jst.last_pairs = ksp_engine_tile_pairs(e, tile_begin, tile_end);
    {   // bytes the kernel streams from both block lists; self tiles read info [+ weight] only
        const u64 per = 4;   // only the 32-bit ranks are streamed; posting words are gathered on matches
        u64 bytes = 0;
        for (size_t i = act0; e->sched_on && i < act1; ++i) {   // work-list mode: the active tiles only
            const u32 I = e->act_rec[4 * i], J = e->act_rec[4 * i + 1];
            const u64 kI = e->h_blk_off[I + 1] - e->h_blk_off[I], kJ = e->h_blk_off[J + 1] - e->h_blk_off[J];
            bytes += (J == I) ? kI * (e->weighted ? 8 : 4) : (kI + kJ) * per;
        }
        for (u64 t = tile_begin; !e->sched_on && t < tile_end;) {   // dense mode, row by row in closed form (h_blk_off is a prefix sum)
            u32 I, J;
            tile_decode(t, e->nb, I, J);
            u64 stop = std::min(tile_row_start((u64)I + 1, e->nb), tile_end);
            u64 kI = e->h_blk_off[I + 1] - e->h_blk_off[I];
            u64 cnt = stop - t, J0 = J;
            if (J0 == I) { bytes += kI * (e->weighted ? 8 : 4); ++J0; --cnt; }
            if (cnt) bytes += (cnt * kI + (u64)(e->h_blk_off[J0 + cnt] - e->h_blk_off[J0])) * per;
            t = stop;
        }
        e->jst.last_stream_bytes = bytes;
    }
    e->join_cap = capacity;
    e->join_pending = true;
    return KSP_OK;
}

int ksp_engine_join_wait(ksp_engine* e, uint64_t* h_count) {
    if (!e || !h_count) { set_error("join: NULL argument"); return KSP_E_ARG; }
    *h_count = 0;
    if (!e->join_pending) return KSP_OK;   // (nothing was launched: no tile in range, or no shared key at all)
    e->join_pending = false;
    KSP_HIP(hipSetDevice(e->device));
    if (e->join_flag) {
        e->join_flag = false;
        volatile unsigned long long* f = reinterpret_cast<volatile unsigned long long*>(e->h_count + 6);
        for (unsigned long long spins = 1; *f != e->join_seq; ++spins) {
            if ((spins & 0xFFFFF) == 0) {   // (now and then: is the stream still alive?)
                const hipError_t q = hipStreamQuery(e->rb_stream);
                if (q == hipSuccess && *f != e->join_seq) KSP_HIP(hipErrorUnknown);
                if (q != hipErrorNotReady && q != hipSuccess) KSP_HIP(q);
            }
        }
        e->st.ms_join = 0;
    } else {
        KSP_HIP(hipEventSynchronize(e->ev_join_done));
        KSP_HIP(hipEventElapsedTime(&e->st.ms_join, e->ev[2], e->ev[3]));
    }
    *h_count = *reinterpret_cast<volatile unsigned long long*>(e->h_count);
    e->st.last_tiles = e->jst.last_tiles;
    e->st.last_active_tiles = e->jst.last_active_tiles;
    e->st.last_pairs = e->jst.last_pairs;
    e->st.last_stream_bytes = e->jst.last_stream_bytes;
    e->st.last_edges = *h_count;
#ifdef KSP_WGTIME
    if (const char* wgt_file = std::getenv("KSP_WGTIME_FILE")) {
        const size_t wgt_n = e->wgt_n;
        if (wgt_n) {
            std::vector<unsigned long long> h(wgt_n * 16);
            KSP_HIP(hipMemcpy(h.data(), e->wgt_buf.p, wgt_n * 128, hipMemcpyDeviceToHost));
            if (FILE* f = std::fopen(wgt_file, "wb")) { std::fwrite(h.data(), 128, wgt_n, f); std::fclose(f); }
        }
    }
#endif
    if (*h_count > e->join_cap) {
        set_error("join: edge buffer too small (" + std::to_string(*h_count) + " > " + std::to_string(e->join_cap) + ")");
        return KSP_E_OVERFLOW;
    }
    return KSP_OK;
}

int ksp_engine_join(ksp_engine* e, uint64_t tile_begin, uint64_t tile_end, ksp_edge* d_edges, uint64_t capacity,
                    uint64_t* h_count, void* stream) {
    if (!h_count) { set_error("join: NULL argument"); return KSP_E_ARG; }
    *h_count = 0;
    int rc = ksp_engine_join_launch(e, tile_begin, tile_end, d_edges, capacity, stream);
    if (rc) return rc;
    return ksp_engine_join_wait(e, h_count);
}

// The edges of tiles [tile_begin, tile_end) straight into the caller's (pinned) host buffer: the range is cut into
// pieces by the edge bound, piece k + 1 is joined while piece k travels over PCIe on a stream of its own.  The
// reference has no such step (its pair map lives in host memory, src/pairwise.cpp:191); here the result of the
// large configurations is hundreds of MB and its copy was a third to a half of their step when it followed the join.
int ksp_engine_join_to_host(ksp_engine* e, uint64_t tile_begin, uint64_t tile_end, ksp_edge* h_edges, uint64_t capacity,
                            uint64_t* h_count, void* stream) {
    if (!e || !h_count) { set_error("join_to_host: NULL argument"); return KSP_E_ARG; }
    *h_count = 0;
    if (!e->built) { set_error("join: build_blocks has not been run"); return KSP_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    KSP_HIP(hipSetDevice(e->device));
    const u64 T = ksp_engine_num_tiles(e);
    if (tile_end > T) tile_end = T;
    if (tile_begin >= tile_end) return KSP_OK;
    if (!e->copy_stream) KSP_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i)
        if (!e->ev_copy[i]) KSP_HIP(hipEventCreateWithFlags(&e->ev_copy[i], hipEventDisableTiming));
    // Pieces of whole tiles.  The staging buffers hold `cap` edges; the first piece is cut by the edge BOUND (it cannot
    // overflow), the later ones by the bound x the densest ratio of found / bound seen so far (+ 50 %): with clustered
    // sources a few per cent of a tile's pairs share anything (1M read groups: 4.5 %), and pieces cut by the bound alone
    // were dozens of small launches.  A piece that does overflow its staging buffer is joined again, halved.
    u64 cap = 1ull << 24;   // 256 MB of edges per staging buffer
    if (const char* pv = std::getenv("KSP_DEBUG_PIECE")) cap = std::max<u64>((u64)TB * TB, std::strtoull(pv, nullptr, 10));
    int rc = KSP_OK;
    for (int i = 0; i < 2; ++i)
        if ((rc = e->stage[i].ensure((cap + 16) * sizeof(ksp_edge)))) return rc;
    const std::vector<u32>& P = e->h_blk_src;
    size_t ai = e->sched_on ? std::lower_bound(e->act_tid.begin(), e->act_tid.end(), tile_begin) - e->act_tid.begin() : 0;
    const size_t a_end = e->sched_on ? std::lower_bound(e->act_tid.begin(), e->act_tid.end(), tile_end) - e->act_tid.begin() : 0;
    u64 total = 0;
    float ms = 0;
    u64 tiles = 0, act = 0, pairs = 0, bytes = 0;
    bool overflow = false;
    double ratio = e->piece_ratio;   // densest found / bound so far (0: nothing joined yet on this engine; else what the previous
                                     // call saw — a caller that joins the same kind of data again starts with pieces of the right size)
    double seen = 0;
    u64 shrink = 1;        // (a piece that overflowed: the limit divided by this until one fits)
    u64 t = tile_begin;
    for (u32 k = 0; t < tile_end;) {
        u64 limit = ratio > 0 ? (u64)((double)cap / (1.5 * ratio)) : cap;
        limit = std::max<u64>(limit / shrink, 1);
        // the piece [t, t_next): whole tiles while their bound fits the limit (one tile at least)
        u64 t_next = tile_end, bound = 0;
        size_t aj = ai;
        if (e->sched_on) {
            for (; aj < a_end; ++aj) {
                const u32 I = e->act_rec[4 * aj], J = e->act_rec[4 * aj + 1];
                const u64 nI = P[I + 1] - P[I], nJ = P[J + 1] - P[J];
                const u64 b = (I == J) ? nI * (nI ? nI - 1 : 0) / 2 : nI * nJ;
                if (bound && bound + b > limit) { t_next = e->act_tid[aj]; break; }
                bound += b;
            }
        } else {
            const u64 per = std::max<u64>(1, limit / ((u64)TB * TB));
            t_next = std::min(tile_end, t + per);
            bound = ksp_engine_tile_pairs(e, t, t_next);
        }
        Buf& sb = e->stage[k & 1];
        if (k >= 2) KSP_HIP(hipStreamWaitEvent(st, e->ev_copy[k & 1], 0));   // (the copy that last read this staging buffer)
        u64 cnt = 0;
        rc = ksp_engine_join(e, t, t_next, sb.as<ksp_edge>(), cap + 16, &cnt, st);   // (returns when the piece is joined; the piece before is on its way meanwhile)
        ms += e->st.ms_join;
        if (rc == KSP_E_OVERFLOW && (t_next > t + 1) && shrink < (1ull << 40)) {   // denser than anything before: the same tiles in smaller pieces
            if (bound) ratio = std::max(ratio, std::min(1.0, (double)cnt / (double)bound));
            shrink *= 2;
            continue;
        }
        if (rc) return rc;
        shrink = 1;
        if (bound) { const double r1 = std::min(1.0, (double)cnt / (double)bound); ratio = std::max(ratio, r1); seen = std::max(seen, r1); }
        tiles += e->st.last_tiles; act += e->st.last_active_tiles; pairs += e->st.last_pairs; bytes += e->st.last_stream_bytes;
        if (total + cnt > capacity) overflow = true;
        else if (cnt) {
            if (!h_edges) { set_error("join_to_host: h_edges is NULL"); return KSP_E_ARG; }
            KSP_HIP(hipMemcpyAsync(h_edges + total, sb.p, cnt * sizeof(ksp_edge), hipMemcpyDeviceToHost, e->copy_stream));
        }
        KSP_HIP(hipEventRecord(e->ev_copy[k & 1], e->copy_stream));
        total += cnt;
        t = t_next;
        ai = aj;
        ++k;
    }
    KSP_HIP(hipStreamSynchronize(e->copy_stream));
    e->piece_ratio = seen;
    *h_count = total;
    e->st.ms_join = ms; e->st.last_tiles = tiles; e->st.last_active_tiles = act; e->st.last_pairs = pairs; e->st.last_stream_bytes = bytes;
    e->st.last_edges = total;
    if (overflow) {
        set_error("join_to_host: host buffer too small (" + std::to_string(total) + " > " + std::to_string(capacity) + ")");
        return KSP_E_OVERFLOW;
    }
    return KSP_OK;
}

// One step of a pipelined caller in ONE call: stage 1, this rank's tile range of equal estimated work, and — when its edge
// bound fits the caller's buffer — the join launched right behind it.  What bench.py did from Python between a build and
// the launch of its join (cuts, bound, buffer check: ~50 us during which the device had nothing to do) happens here.
// KSP_E_OVERFLOW: the bound (in *bound) exceeds `capacity`; nothing was launched — grow the buffer and call
// ksp_engine_join_launch(e, range[0], range[1], ...) yourself.
int ksp_engine_step_launch(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights, const uint64_t* h_offsets, uint32_t n_sources,
                           int key_bits, uint32_t part, uint32_t nparts, ksp_edge* d_edges, uint64_t capacity, uint64_t range[2],
                           uint64_t* bound, uint64_t* prev_count, int* prev_status, float* prev_ms_join, void* stream) {
    if (!range || !bound || !prev_count || !prev_status || nparts == 0 || part >= nparts) { set_error("step_launch: bad argument"); return KSP_E_ARG; }
    *prev_count = 0;
    *prev_status = KSP_OK;
    if (prev_ms_join) *prev_ms_join = 0;
    const bool had_join = e && e->join_pending;
    const u64 had_cap = e ? e->join_cap : 0;
    if (e) e->early_ok = true;   // (the join follows at once, on the same stream: the work list may leave the build early)
    const bool lean = e && !e->profiling && !std::getenv("KSP_DEBUG_STEP_EVENTS");   // (no timing events in the stream: each is a ~6 us bubble)
    if (e) { e->time_sort = !lean; e->lean = lean; }
    int rc = ksp_engine_build_blocks(e, d_keys, d_weights, h_offsets, n_sources, key_bits, stream);
    if (e) { e->early_ok = false; e->time_sort = true; }
    if (rc) { if (e) e->lean = false; return rc; }
    if (had_join) {   // the join launched before this build ran in front of it on the stream: its count is there
        e->join_pending = true;
        e->join_cap = had_cap;
        *prev_status = ksp_engine_join_wait(e, prev_count);
        if (prev_ms_join) *prev_ms_join = e->st.ms_join;
    }
    std::vector<u64> cuts((size_t)nparts + 1);
    if ((rc = ksp_engine_balanced_cuts(e, nparts, cuts.data()))) { e->lean = false; return rc; }
    range[0] = cuts[part]; range[1] = cuts[(size_t)part + 1];
    *bound = ksp_engine_edge_bound(e, range[0], range[1]);
    if (*bound + 1 > capacity) { e->lean = false; set_error("step_launch: the edge bound exceeds the buffer (nothing launched)"); return KSP_E_OVERFLOW; }
    rc = ksp_engine_join_launch(e, range[0], range[1], d_edges, capacity, stream);
    e->lean = false;
    return rc;
}

int ksp_engine_set_profiling(ksp_engine* e, int on) {
    if (!e) return KSP_E_ARG;
    e->profiling = on != 0;
    e->ph_n = 0;
    return KSP_OK;
}

int ksp_engine_phase_times(const ksp_engine* e, const char** names, float* ms, int cap) {
    if (!e || cap < 0 || (cap && (!names || !ms))) return 0;
    const int n = std::min(cap, e->ph_n);
    for (int i = 0; i < n; ++i) { names[i] = e->ph_name[i]; ms[i] = e->ph_ms[i]; }
    return n;
}

int ksp_engine_get_stats(const ksp_engine* e, ksp_stats* out) {
    if (!e || !out) return KSP_E_ARG;
    KSP_HIP(resolve_build_ms(const_cast<ksp_engine*>(e)));
    *out = e->st;
    return KSP_OK;
}

// (diagnostics) engine index of every source after the reordering: block = index / 128
int ksp_engine_source_order(const ksp_engine* e, uint32_t* h_newidx /* n_sources */) {
    if (!e || !h_newidx) return KSP_E_ARG;
    if (!e->n_sources || !e->smap.p) return KSP_OK;
    const size_t NN = smap_stride(e);
    KSP_HIP(hipMemcpy(h_newidx, e->smap.as<u32>() + 4 * NN, (size_t)e->n_sources * 4, hipMemcpyDeviceToHost));
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

// results handed to the caller live in pinned host memory (the device-to-host copy runs at the full link
// rate straight into the buffer the caller gets); ksp_free tells them from malloc'ed blocks by this registry
static std::mutex g_pinned_mu;
static std::set<void*> g_pinned;
static void* alloc_result(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(g_pinned_mu);
    g_pinned.insert(p);
    return p;
}
void ksp_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pinned_mu);
        auto it = g_pinned.find(p);
        if (it != g_pinned.end()) { g_pinned.erase(it); (void)hipHostFree(p); return; }
    }
    std::free(p);
}

// ---- collecting the result: join -> [gather to the first device over xGMI] -> device sort -> pinned host ----
// edges of tiles [t0, t1) of a built engine into a device buffer that grows to the size the join reports
static int join_range_grow(ksp_engine* e, u64 t0, u64 t1, Buf& buf, u64& count, float& ms_join) {
    count = 0;
    if (t0 >= t1) return KSP_OK;
    int rc;
    if (buf.bytes < sizeof(ksp_edge)) {
        const u64 first = std::min<u64>(ksp_engine_edge_bound(e, t0, t1) + 1, 1ull << 26);   // at most 1 GiB to begin with
        if ((rc = buf.ensure(first * sizeof(ksp_edge)))) return rc;
    }
    for (int attempt = 0; attempt < 3; ++attempt) {
        u64 cnt = 0;
        rc = ksp_engine_join(e, t0, t1, buf.as<ksp_edge>(), buf.bytes / sizeof(ksp_edge), &cnt, nullptr);
        ms_join += e->st.ms_join;
        if (rc == KSP_OK) { count = cnt; return KSP_OK; }
        if (rc != KSP_E_OVERFLOW) return rc;
        if ((rc = buf.ensure((cnt + cnt / 16 + 1024) * sizeof(ksp_edge)))) {   // (the count the join reported, plus slack)
            set_error("pairwise_host: the edges of this tile range do not fit in device memory");
            return KSP_E_LIMIT;
        }
    }
    set_error("pairwise_host: edge count kept growing between joins");
    return KSP_E_HIP;
}

__global__ void k_edge_split(const ksp_edge* __restrict__ ed, u64 n, u64* __restrict__ key, u64* __restrict__ val) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const ksp_edge x = ed[i];
        key[i] = ((u64)x.source_1 << 32) | x.source_2;
        val[i] = x.shared;
    }
}
__global__ void k_edge_merge(const u64* __restrict__ key, const u64* __restrict__ val, u64 n, ksp_edge* __restrict__ ed) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        ksp_edge x;
        x.source_1 = (u32)(key[i] >> 32);
        x.source_2 = (u32)key[i];
        x.shared = val[i];
        ed[i] = x;
    }
}
// rows in (source_1, source_2) order — the order the TSV writer emits (the reference's own order is hash-map
// iteration order, src/pairwise.cpp:253) — by one device radix sort over the significant key bits
static int sort_edges_device(ksp_edge* d_edges, u64 n, u32 n_sources) {
    if (n < 2) return KSP_OK;
    Buf k0, k1, v0, v1, tmp;
    int rc = KSP_OK;
    do {
        if ((rc = k0.ensure(n * 8)) || (rc = k1.ensure(n * 8)) || (rc = v0.ensure(n * 8)) || (rc = v1.ensure(n * 8))) break;
        const unsigned grid = (unsigned)std::min<u64>((n + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(k_edge_split, dim3(grid), dim3(256), 0, nullptr, d_edges, n, k0.as<u64>(), v0.as<u64>());
        int sbits = 1;
        while (sbits < 32 && (n_sources >> sbits)) ++sbits;
        size_t tb = 0;
        hipError_t err = rocprim::radix_sort_pairs(nullptr, tb, k0.as<u64>(), k1.as<u64>(), v0.as<u64>(), v1.as<u64>(), (size_t)n, 0, 32 + sbits, (hipStream_t) nullptr);
        if (err == hipSuccess && !(rc = tmp.ensure(tb)))
            err = rocprim::radix_sort_pairs(tmp.p, tb, k0.as<u64>(), k1.as<u64>(), v0.as<u64>(), v1.as<u64>(), (size_t)n, 0, 32 + sbits, (hipStream_t) nullptr);
        if (rc) break;
        if (err != hipSuccess) { set_error(std::string("sort_edges: ") + hipGetErrorString(err)); rc = KSP_E_HIP; break; }
        hipLaunchKernelGGL(k_edge_merge, dim3(grid), dim3(256), 0, nullptr, k1.as<u64>(), v1.as<u64>(), n, d_edges);
        err = hipDeviceSynchronize();
        if (err != hipSuccess) { set_error(std::string("sort_edges: ") + hipGetErrorString(err)); rc = KSP_E_HIP; }
    } while (0);
    k0.release(); k1.release(); v0.release(); v1.release(); tmp.release();
    return rc;
}

namespace {
struct MultiJob {
    // input (host): sketches (keys / weights / offsets) or an inverted index (key_off / sources / key_weights)
    const u64* keys = nullptr; const u32* weights = nullptr; const u64* offsets = nullptr;
    const u64* key_off = nullptr; const u32* sources = nullptr; const u32* key_weights = nullptr;
    u32 n_keys = 0, n_sources = 0;
    bool postings = false;
    ksp::CcRequest* cc = nullptr;   // also wanted: the components of the result, from the edges while they are on the device
};
}  // namespace

// The whole job on `nd` devices, one host thread + one engine per device (the reference entry points call this
// with the devices of $KSPIDER_DEVICES; nd = 1 is the single-GPU path):
//   stage 1   device i builds its slice — sketch input: its 1/nd share of the hash range; postings input (the reference's
//             entry point): 1/nd of the colours (whole keys, cut by memberships) — the labels are MIN-combined and the
//             slices exchanged device to device (peer copies over xGMI), then every device assembles the full lists:
//             the same exchange kspider_amd/dist.py does with RCCL collectives between processes.  Inputs of 2^30
//             entries or more are cut into more slices than there are devices (several workers per device).
//   stage 2   device i joins the tile range [cuts[i], cuts[i+1]) of equal estimated work (same cuts everywhere:
//             checked), the edges are gathered to the first device (peer copies), sorted there and copied into
//             pinned host memory.
static int run_multi(const MultiJob& job, const int* devices, int nd, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats) {
    *out_edges = nullptr;
    *n_edges = 0;
    if (nd < 1 || nd > 64) { set_error("pairwise: between 1 and 64 devices"); return KSP_E_ARG; }
    const u32 N = job.n_sources;
    const u64 n = job.postings ? (job.n_keys ? job.key_off[job.n_keys] : 0) : (N ? job.offsets[N] : 0);
    // A sketch set of 2^30 entries or more does not fit one build (32-bit entry positions): it is cut into hash-range
    // slices of at most ~0.9 * 2^30 entries, built one engine each and assembled — the multi-GPU machinery with several
    // workers per device ($KSP_SLICES forces a slice count: tests).
    std::vector<int> expanded;
    std::vector<u32> pk_cut;   // postings input in slices: slice s holds the keys [pk_cut[s], pk_cut[s + 1])
    if (!job.postings) {
        u64 want = n < (1ull << 30) ? 1 : n / 900000000ull + 1;   // (equal shares of the hash range: ~equal sizes for hashes)
        if (const char* sl = std::getenv("KSP_SLICES")) want = std::max<u64>(want, (u64)std::max(1, std::atoi(sl)));
        if (want > (u64)nd) {
            if (want > 64) { set_error("pairwise_host: more than 2^35 key entries"); return KSP_E_LIMIT; }
            for (u64 i = 0; i < want; ++i) expanded.push_back(devices[i % (u64)nd]);
            devices = expanded.data();
            nd = (int)want;
        }
    } else {
        // An inverted index (the reference's colour -> sources map, src/pairwise.cpp:95-111, which has no size limit): slices
        // of whole keys with fewer than 2^30 memberships each — one per device of the job, more (built side by side on the
        // devices there are) when the index is larger than that, $KSP_SLICES forces a count (tests).  Every device then
        // builds 1 / nd of the colours instead of all of them.
        u64 want = n < (1ull << 30) ? 1 : n / 900000000ull + 1;
        if (const char* sl = std::getenv("KSP_SLICES")) want = std::max<u64>(want, (u64)std::max(1, std::atoi(sl)));
        want = std::max<u64>(want, (u64)nd);
        if (want > (u64)job.n_keys && n < (1ull << 30)) want = 1;   // (fewer colours than slices: every device builds the few there are, the tiles are still shared)
        if (want > 1) {
            if (want > 64) { set_error("pairwise: more than 2^35 colour memberships"); return KSP_E_LIMIT; }
            if (want > (u64)nd) {
                for (u64 i = 0; i < want; ++i) expanded.push_back(devices[i % (u64)nd]);
                devices = expanded.data();
                nd = (int)want;
            }
            pk_cut.assign((size_t)nd + 1, 0);
            u32 k = 0;
            for (int sidx = 1; sidx < nd; ++sidx) {   // cut where the memberships reach s / nd of all (whole keys, no empty slice)
                const u64 goal = n / (u64)nd * (u64)sidx;
                while (k < job.n_keys && job.key_off[k] < goal) ++k;
                k = std::max<u32>(k, pk_cut[(size_t)sidx - 1] + 1);
                k = std::min<u32>(k, job.n_keys - (u32)(nd - sidx));
                pk_cut[(size_t)sidx] = k;
            }
            pk_cut[(size_t)nd] = job.n_keys;
            for (int sidx = 0; sidx < nd; ++sidx)
                if (job.key_off[pk_cut[(size_t)sidx + 1]] - job.key_off[pk_cut[(size_t)sidx]] >= (1ull << 30)) {
                    set_error("pairwise: one colour range holds 2^30 memberships or more (a single colour that large?)");
                    return KSP_E_LIMIT;
                }
        }
    }
    const bool sliced = job.postings ? !pk_cut.empty() : nd > 1;
    struct Dev {
        ksp_engine* e = nullptr;
        void *d_a = nullptr, *d_b = nullptr;      // keys + weights, or sources + key weights
        Buf edges, labels, gather[6], exp[6];
        u64 count = 0, sizes[4] = {0, 0, 0, 0};
        std::vector<u64> cuts;
        float ms_join = 0;
        int rc = KSP_OK;
        bool borrowed = false;                    // d_a / d_b belong to another worker on the same device
        std::string err;
    };
    std::vector<Dev> dev((size_t)nd);
    FailBarrier bar(nd);   // (the failure decision is latched once per barrier generation: host_sync.h)
    std::vector<u32> lab_min;                 // MIN-combined labels (host)
    std::vector<std::vector<u32>> lab_dev((size_t)nd);
    std::vector<u64> all_sizes((size_t)nd * 4, 0);
    u64 total = 0;
    std::vector<u64> edge_off((size_t)nd + 1, 0);
    Buf merged;                               // all edges on the first device (nd > 1)

    auto body = [&](int i) {
        Dev& D = dev[(size_t)i];
        const int device = devices[i];
        auto fail = [&](int rc) { D.rc = rc; D.err = g_error; bar.fail(); };
        auto sync_point = [&]() { return bar.sync(); };
        int rc = ksp_engine_create(device, &D.e);
        if (rc) fail(rc);
        for (int j = 0; j < nd && !rc; ++j)   // direct xGMI copies between the devices of this job (already enabled: fine)
            if (devices[j] != device) { (void)hipDeviceEnablePeerAccess(devices[j], 0); (void)hipGetLastError(); }
        // ---- stage 1 --------------------------------------------------------------------------------------------
        int owner = i;   // the first worker on my device uploads the input; the others use its copy
        for (int j = 0; j < i; ++j)
            if (devices[j] == device) { owner = j; break; }
        if (job.postings && sliced) owner = i;   // (every worker uploads its own slice of the index)
        std::vector<u64> my_off;                 // key offsets of my slice, from 0
        u32 my_keys = 0;
        if (!bar.failed_hint() && job.postings && sliced) {
            const u32 k0 = pk_cut[(size_t)i], k1 = pk_cut[(size_t)i + 1];
            my_keys = k1 - k0;
            my_off.resize((size_t)my_keys + 1);
            const u64 e0 = job.key_off[k0];
            for (u32 k = 0; k <= my_keys; ++k) my_off[k] = job.key_off[k0 + k] - e0;
            const u64 ne = my_off[my_keys];
            if ((rc = ksp_device_malloc(device, ne * 4, &D.d_a)) || (rc = ksp_memcpy_h2d(D.d_a, job.sources + e0, ne * 4))) fail(rc);
            if (!rc && job.key_weights &&   // (allocated even for an empty slice: a non-NULL weight array is what makes a build weighted)
                ((rc = ksp_device_malloc(device, (u64)my_keys * 4, &D.d_b)) || (rc = ksp_memcpy_h2d(D.d_b, job.key_weights + k0, (u64)my_keys * 4))))
                fail(rc);
        } else if (!bar.failed_hint() && n && owner == i) {
            if (job.postings) {
                if ((rc = ksp_device_malloc(device, n * 4, &D.d_a)) || (rc = ksp_memcpy_h2d(D.d_a, job.sources, n * 4))) fail(rc);
                if (!rc && job.key_weights &&
                    ((rc = ksp_device_malloc(device, (u64)job.n_keys * 4, &D.d_b)) || (rc = ksp_memcpy_h2d(D.d_b, job.key_weights, (u64)job.n_keys * 4))))
                    fail(rc);
            } else {
                if ((rc = ksp_device_malloc(device, n * 8, &D.d_a)) || (rc = ksp_memcpy_h2d(D.d_a, job.keys, n * 8))) fail(rc);
                if (!rc && job.weights && ((rc = ksp_device_malloc(device, n * 4, &D.d_b)) || (rc = ksp_memcpy_h2d(D.d_b, job.weights, n * 4))))
                    fail(rc);
            }
        }
        if (nd > 1 && sync_point()) return;   // (the uploads are complete: ksp_memcpy_h2d is synchronous)
        if (owner != i) { D.d_a = dev[(size_t)owner].d_a; D.d_b = dev[(size_t)owner].d_b; D.borrowed = true; }
        if (!bar.failed_hint()) {
            if (job.postings && sliced)
                rc = ksp_engine_build_postings_slice(D.e, my_off.data(), (const u32*)D.d_a, (const u32*)D.d_b, my_keys, N, nullptr);
            else if (job.postings)
                rc = ksp_engine_build_postings(D.e, job.key_off, (const u32*)D.d_a, (const u32*)D.d_b, job.n_keys, N, nullptr);
            else if (nd == 1)
                rc = ksp_engine_build_blocks(D.e, (const u64*)D.d_a, (const u32*)D.d_b, job.offsets, N, 0, nullptr);
            else
                rc = ksp_engine_build_slice(D.e, (const u64*)D.d_a, (const u32*)D.d_b, job.offsets, N, 0, (u32)i, (u32)nd, nullptr);
            if (rc) fail(rc);
        }
        if (sliced) {   // the slices become the full lists on every device
            // common source order: element-wise MIN of the devices' labels
            if (!bar.failed_hint() && N) {
                lab_dev[(size_t)i].resize(N);
                if ((rc = D.labels.ensure((size_t)N * 4)) || (rc = ksp_engine_slice_labels(D.e, D.labels.as<u32>(), nullptr)) ||
                    (rc = ksp_memcpy_d2h(lab_dev[(size_t)i].data(), D.labels.p, (u64)N * 4)))
                    fail(rc);
            }
            if (sync_point()) return;
            if (i == 0) {
                lab_min = lab_dev[0];
                for (int j = 1; j < nd; ++j)
                    for (u32 s = 0; s < N; ++s) lab_min[s] = std::min(lab_min[s], lab_dev[(size_t)j][s]);
            }
            if (sync_point()) return;
            if (N && ((rc = ksp_memcpy_h2d(D.labels.p, lab_min.data(), (u64)N * 4)) || (rc = ksp_engine_slice_finish(D.e, D.labels.as<u32>(), nullptr))))
                fail(rc);
            else if (!N && (rc = ksp_engine_slice_finish(D.e, nullptr, nullptr)))
                fail(rc);
            if (!bar.failed_hint() && (rc = ksp_engine_slice_sizes(D.e, D.sizes))) fail(rc);
            for (int q = 0; q < 4; ++q) all_sizes[(size_t)i * 4 + q] = D.sizes[q];
            if (sync_point()) return;
            // every device receives every slice: [part][stride] buffers, filled by peer copies from the owners
            u64 lstride = 4, bigstride = 1;
            for (int j = 0; j < nd; ++j) { lstride = std::max(lstride, all_sizes[(size_t)j * 4]); bigstride = std::max(bigstride, all_sizes[(size_t)j * 4 + 2]); }
            const u32 nb = D.e->nb;   // (the same on every device: a function of the source count)
            const bool weighted = job.postings ? job.key_weights != nullptr : job.weights != nullptr;
            const size_t bytes[6] = {lstride * 4, lstride * 4, weighted ? lstride * 4 : 0, ((size_t)nb + 1) * 4, ((size_t)nb + 1) * 4, bigstride * 16};
            for (int q = 0; q < 6 && !rc; ++q) {
                if (!bytes[q]) continue;
                if ((rc = D.exp[q].ensure(bytes[q])) || (rc = D.gather[q].ensure(bytes[q] * (size_t)nd))) fail(rc);
                else if (hipMemset(D.exp[q].p, 0, bytes[q]) != hipSuccess || hipMemset(D.gather[q].p, 0, bytes[q] * (size_t)nd) != hipSuccess) { set_error("pairwise: hipMemset"); fail(rc = KSP_E_HIP); }
            }
            if (!bar.failed_hint() && n &&
                (rc = ksp_engine_slice_export(D.e, D.exp[0].as<u32>(), D.exp[1].as<u32>(), weighted ? D.exp[2].as<u32>() : nullptr,
                                              D.exp[3].as<u32>(), D.exp[4].as<u32>(), D.exp[5].p, nullptr)))
                fail(rc);
            if (sync_point()) return;
            for (int j = 0; j < nd && !bar.failed_hint(); ++j)      // my slice into device j's gather buffers
                for (int q = 0; q < 6; ++q) {
                    if (!bytes[q]) continue;
                    if (hipMemcpyPeer((char*)dev[(size_t)j].gather[q].p + bytes[q] * (size_t)i, devices[j], D.exp[q].p, device, bytes[q]) != hipSuccess) {
                        set_error("pairwise: peer copy of a block-list slice failed");
                        fail(KSP_E_HIP);
                        break;
                    }
                }
            if (sync_point()) return;
            if ((rc = ksp_engine_assemble(D.e, (u32)nd, all_sizes.data(), D.gather[0].as<u32>(), D.gather[1].as<u32>(),
                                          weighted ? D.gather[2].as<u32>() : nullptr, lstride, D.gather[3].as<u32>(), D.gather[4].as<u32>(),
                                          D.gather[5].p, bigstride, nullptr)))
                fail(rc);
        }
        if (sync_point()) return;
        // ---- stage 2: my share of the tiles ------------------------------------------------------------------
        D.cuts.assign((size_t)nd + 1, 0);
        if ((rc = ksp_engine_balanced_cuts(D.e, (u32)nd, D.cuts.data()))) fail(rc);
        if (sync_point()) return;
        if (i == 0)
            for (int j = 1; j < nd; ++j)
                if (dev[(size_t)j].cuts != dev[0].cuts || dev[(size_t)j].e->st.n_block_keys != dev[0].e->st.n_block_keys) {
                    set_error("pairwise: the devices built different block lists (cannot shard the tiles)");
                    fail(KSP_E_HIP);
                }
        if (sync_point()) return;
        if ((rc = join_range_grow(D.e, D.cuts[(size_t)i], D.cuts[(size_t)i + 1], D.edges, D.count, D.ms_join))) fail(rc);
        if (sync_point()) return;
        if (i == 0) {
            for (int j = 0; j < nd; ++j) edge_off[(size_t)j + 1] = edge_off[(size_t)j] + dev[(size_t)j].count;
            total = edge_off[(size_t)nd];
            if (nd > 1 && total && (rc = merged.ensure(total * sizeof(ksp_edge)))) fail(rc);
        }
        if (sync_point()) return;
        if (nd > 1 && D.count &&      // the gather to the first device: one peer copy per device (xGMI)
            hipMemcpyPeer((char*)merged.p + edge_off[(size_t)i] * sizeof(ksp_edge), devices[0], D.edges.p, device, D.count * sizeof(ksp_edge)) != hipSuccess) {
            set_error("pairwise: peer copy of the edges failed");
            fail(KSP_E_HIP);
        }
        if (sync_point()) return;
        if (i == 0) {
            ksp_edge* d_all = nd > 1 ? merged.as<ksp_edge>() : D.edges.as<ksp_edge>();
            if (total && (rc = sort_edges_device(d_all, total, N))) { fail(rc); return; }
            if (job.cc && job.cc->labels) {   // clustering from HBM: the edges never travel to the host and back as text for this
                job.cc->labels->assign((size_t)N, 0);
                Buf d_cnt;
                if (N && ((rc = d_cnt.ensure((size_t)N * 4)) || (rc = ksp_memcpy_h2d(d_cnt.p, job.cc->kmer_counts, (u64)N * 4)) ||
                          (rc = cc_edges_on_device(N, d_all, total, d_cnt.as<u32>(), job.cc->col, job.cc->cutoff, job.cc->labels->data(), &job.cc->n_kept)))) {
                    d_cnt.release();
                    fail(rc);
                    return;
                }
                d_cnt.release();
            }
            ksp_edge* out = (ksp_edge*)alloc_result(total * sizeof(ksp_edge));
            if (!out) { set_error("pairwise_host: out of pinned host memory"); fail(KSP_E_LIMIT); return; }
            if (total && (rc = ksp_memcpy_d2h(out, d_all, total * sizeof(ksp_edge)))) { ksp_free(out); fail(rc); return; }
            *out_edges = out;
            *n_edges = total;
            if (stats) {
                ksp_engine_get_stats(D.e, stats);
                stats->ms_join = 0; stats->last_tiles = 0; stats->last_pairs = 0;
                for (int j = 0; j < nd; ++j) {
                    stats->ms_join = std::max(stats->ms_join, dev[(size_t)j].ms_join);
                    stats->last_tiles += dev[(size_t)j].e->st.last_tiles;
                    stats->last_pairs += dev[(size_t)j].e->st.last_pairs;
                }
                stats->last_edges = total;
            }
        }
    };
    if (nd == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nd; ++i) th.emplace_back(body, i);
        for (auto& t : th) t.join();
    }
    int rc = KSP_OK;
    for (int i = 0; i < nd; ++i) {
        Dev& D = dev[(size_t)i];
        if (D.rc && !rc) { rc = D.rc; set_error(D.err); }
        if (D.e) (void)hipSetDevice(D.e->device);
        if (D.d_a && !D.borrowed) (void)hipFree(D.d_a);
        if (D.d_b && !D.borrowed) (void)hipFree(D.d_b);
        D.edges.release(); D.labels.release();
        for (int q = 0; q < 6; ++q) { D.gather[q].release(); D.exp[q].release(); }
        ksp_engine_destroy(D.e);
    }
    if (nd > 1 || merged.p) { (void)hipSetDevice(devices[0]); merged.release(); }
    if (rc && *out_edges) { ksp_free(*out_edges); *out_edges = nullptr; *n_edges = 0; }
    return rc;
}

#ifdef KSP_FKTIME
int ksp_debug_fktime(unsigned long long* out16, int reset) {
    KSP_HIP(hipDeviceSynchronize());
    KSP_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(ksp::fk_time), 16 * 8));
    if (reset) { unsigned long long z[16] = {0}; KSP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ksp::fk_time), z, 16 * 8)); }
    return KSP_OK;
}
int ksp_debug_sttime(unsigned long long* out64, int reset) {
    KSP_HIP(hipDeviceSynchronize());
    KSP_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(ksp::st_time), 64 * 8));
    if (reset) { unsigned long long z[64] = {0}; z[63] = reset == 2; KSP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ksp::st_time), z, 64 * 8)); }   // (2: probe waits on)
    return KSP_OK;
}
#endif

}  // extern "C"
int ksp::pairwise_postings_multi_cc(const uint64_t* key_off, const uint32_t* sources, const uint32_t* key_weights, uint32_t n_keys,
                                    uint32_t n_sources, const int* devices, int n_devices, ksp_edge** out_edges, uint64_t* n_edges,
                                    ksp_stats* stats, CcRequest* cc) {
    if (!out_edges || !n_edges || !devices || (n_keys && (!key_off || !sources))) { set_error("pairwise_postings_host: NULL argument"); return KSP_E_ARG; }
    const u64 n = n_keys ? key_off[n_keys] : 0;
    for (u64 i = 0; i < n; ++i)
        if (sources[i] >= n_sources) { set_error("pairwise_postings_host: source index out of range"); return KSP_E_ARG; }
    MultiJob job;
    job.postings = true;
    job.key_off = key_off; job.sources = sources; job.key_weights = key_weights; job.n_keys = n_keys; job.n_sources = n_sources;
    job.cc = cc;
    return run_multi(job, devices, n_devices, out_edges, n_edges, stats);
}
extern "C" {

int ksp_pairwise_host_multi(const uint64_t* keys, const uint32_t* weights, const uint64_t* offsets, uint32_t n_sources,
                            const int* devices, int n_devices, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats) {
    if (!offsets || !out_edges || !n_edges || !devices) { set_error("pairwise_host: NULL argument"); return KSP_E_ARG; }
    MultiJob job;
    job.keys = keys; job.weights = weights; job.offsets = offsets; job.n_sources = n_sources;
    return run_multi(job, devices, n_devices, out_edges, n_edges, stats);
}

int ksp_pairwise_postings_host_multi(const uint64_t* key_off, const uint32_t* sources, const uint32_t* key_weights,
                                     uint32_t n_keys, uint32_t n_sources, const int* devices, int n_devices,
                                     ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats) {
    return ksp::pairwise_postings_multi_cc(key_off, sources, key_weights, n_keys, n_sources, devices, n_devices, out_edges, n_edges, stats, nullptr);
}

int ksp_pairwise_host(const uint64_t* keys, const uint32_t* weights, const uint64_t* offsets, uint32_t n_sources,
                      int device, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats) {
    return ksp_pairwise_host_multi(keys, weights, offsets, n_sources, &device, 1, out_edges, n_edges, stats);
}

int ksp_pairwise_postings_host(const uint64_t* key_off, const uint32_t* sources, const uint32_t* key_weights,
                               uint32_t n_keys, uint32_t n_sources, int device, ksp_edge** out_edges,
                               uint64_t* n_edges, ksp_stats* stats) {
    return ksp_pairwise_postings_host_multi(key_off, sources, key_weights, n_keys, n_sources, &device, 1, out_edges, n_edges, stats);
}

}  // extern "C"
