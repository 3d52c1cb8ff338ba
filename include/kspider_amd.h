/* kspider_amd — C ABI of the MI355X-native pairwise containment engine.
 *
 * This is the drop-in boundary for ONE path of dib-lab/kSpider: kSpider::pairwise()
 * (reference: include/kSpider.hpp:11, src/pairwise.cpp:123-276, SWIG wrapper
 * src/swig_interfaces/kSpider_internal.i:11).  Plain pointers and sizes only; no
 * torch / C++ types.  Every function returns 0 on success, a KSP_E_* code otherwise;
 * ksp_last_error() gives the message of the calling thread's last failure.
 *
 * There is NO CPU fallback behind these entry points: without a visible gfx950
 * device (or without the HIP runtime) they fail with KSP_E_HIP.
 */
#ifndef KSPIDER_AMD_H
#define KSPIDER_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    KSP_OK = 0,
    KSP_E_ARG = 1,      /* bad argument */
    KSP_E_HIP = 2,      /* HIP runtime / device failure */
    KSP_E_IO = 3,       /* missing / malformed index file */
    KSP_E_OVERFLOW = 4, /* edge buffer too small: *count holds the required size */
    KSP_E_LIMIT = 5     /* input exceeds an engine limit */
};

/* One row of the reference's PAIRS_COUNTER (src/pairwise.cpp:22-27):
 * key pair<uint32,uint32> with source_1 < source_2, value uint64 shared k-mers. */
typedef struct ksp_edge {
    uint32_t source_1;
    uint32_t source_2;
    uint64_t shared;
} ksp_edge;

typedef struct ksp_engine ksp_engine;

typedef struct ksp_stats {
    uint64_t n_sources;
    uint64_t n_entries;     /* sum of sketch sizes                                  */
    uint64_t n_blocks;      /* blocks of 128 source slots: ceil(n_sources / 128), or up to half as many again when the
                               source reordering keeps clusters inside blocks (slots without a source stay empty) */
    uint64_t n_block_keys;  /* distinct keys summed over blocks                     */
    uint64_t n_tiles;       /* n_blocks (n_blocks + 1) / 2                          */
    uint64_t last_tiles;    /* tiles joined by the last ksp_engine_join             */
    uint64_t last_pairs;    /* source pairs covered by the last ksp_engine_join     */
    uint64_t last_stream_bytes; /* bytes the join kernel streamed (model) */
    uint64_t last_edges;    /* non-zero pairs found by the last join                */
    float ms_build;         /* HIP-event time of the last build_blocks (all kernels) */
    float ms_join;          /* HIP-event time of the last join kernel launch        */
    int weighted;
    int key_bits;
    uint64_t n_active_tiles;    /* tiles with something to count (block pairs sharing a key, diagonal
                                   tiles of blocks with a multi-source key); = n_tiles in dense mode */
    uint64_t last_active_tiles; /* ... among the tiles of the last ksp_engine_join                   */
    uint64_t sort_entries;      /* entries of the global radix sort of the last build (rocPRIM onesweep:
                                   8-byte key + 4/8-byte tag read and written once per 8-bit pass)      */
    float ms_sort;              /* its HIP-event time (all passes + histogram)                        */
    int sort_bits;              /* key bits it sorted on                                              */
    int partition_kind;         /* how the last build brought equal keys together: 0 nothing to do / postings
                                   input, 1 rocPRIM radix partition or sort, 2 the hand-written paged
                                   partition, 3 the segment partition: level 1 read off the sorted runs
                                   (partition_kernels.hip.h)                                             */
    int partition_fallback;     /* 0, or why a hand-written partition handed the build on: 1 page table / pool
                                   full (keys far from uniform; -> rocPRIM), 2 internal count mismatch, 3 page
                                   wait timed out (2 and 3 are defects, never expected), 4 / 5 a tile / a bucket
                                   of the segment partition overflowed (-> the paged partition)           */
    uint64_t n_match_records;   /* match-list join: (key, block pair) records stage 1 handed to the join (0: the
                                   join searches the block lists)                                         */
    uint64_t n_join_workgroups; /* shares of the work list (workgroups of a join over all tiles)           */
    uint64_t n_kept_entries;    /* entries whose key is held by at least two sources (the others are pruned) */
    uint64_t n_kept_keys;       /* distinct keys among them                                                */
    int stage1_kind;            /* middle of stage 1: 1 bucket-resident (grouping + emit + labels in one kernel, group
                                   records straight to rank order: fused_kernels.hip.h), 0 pass by pass      */
    int reserved_;
} ksp_stats;

const char* ksp_last_error(void);
int ksp_device_count(int* count);

/* ---- the hot path on device-resident data ------------------------------------------
 * Replaces the accumulate region of the reference, src/pairwise.cpp:194-237
 * (Combo::combinations + PAIRS_COUNTER::try_emplace_l), for sketches laid out as sorted
 * uint64 runs in HBM.  Source ids in the edges are dense indices 0..n_sources-1.      */
int ksp_engine_create(int device, ksp_engine** out);
void ksp_engine_destroy(ksp_engine* e);

/* Stage 1: merge the sorted runs of every block of 128 sources into one sorted
 * posting list per block (device).  d_keys: device pointer, concatenated sorted-unique
 * uint64 runs; d_weights: device pointer (one uint32 per key entry; the colour weight
 * w_c of src/pairwise.cpp:221) or NULL for weight 1; h_offsets: HOST pointer,
 * n_sources+1 element offsets; key_bits: significant bits of the largest key, 0 = find
 * out on the device.  stream: hipStream_t (NULL = default stream).                    */
int ksp_engine_build_blocks(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights,
                            const uint64_t* h_offsets, uint32_t n_sources, int key_bits, void* stream);

/* Tiles of the block-pair upper triangle in row-major order: tile t <-> (I, J), I <= J. */
uint64_t ksp_engine_num_tiles(const ksp_engine* e);
/* Source pairs covered by tiles [tile_begin, tile_end): the worst-case edge count.     */
uint64_t ksp_engine_tile_pairs(const ksp_engine* e, uint64_t tile_begin, uint64_t tile_end);
/* Tighter bound on the edges tiles [tile_begin, tile_end) can produce: the source pairs of the tiles
 * that share a key at all (stage 1 knows them; equals ksp_engine_tile_pairs in dense mode).       */
uint64_t ksp_engine_edge_bound(const ksp_engine* e, uint64_t tile_begin, uint64_t tile_end);
/* Tile ranges of equal estimated work for `nparts` GPUs: rank p joins tiles [cuts[p], cuts[p+1]).
 * cuts has nparts + 1 entries; every rank computes the same cuts from the same block lists.       */
int ksp_engine_balanced_cuts(const ksp_engine* e, uint32_t nparts, uint64_t* cuts);

/* Stage 2: join tiles [tile_begin, tile_end) and append every pair with shared > 0 to
 * d_edges (device buffer of `capacity` edges; order unspecified).  *h_count receives the
 * number of non-zero pairs; if it exceeds capacity the surplus was dropped and the call
 * returns KSP_E_OVERFLOW.  Synchronises `stream` before returning.                      */
int ksp_engine_join(ksp_engine* e, uint64_t tile_begin, uint64_t tile_end, ksp_edge* d_edges, uint64_t capacity,
                    uint64_t* h_count, void* stream);
/* The same in two halves, for callers that pipeline: _launch queues the join on `stream` and returns; _wait blocks
 * until it has finished and reports the count (one launched join per engine at a time).  Work queued on the same
 * stream after _launch — the next ksp_engine_build_blocks on this engine included — runs behind the join, so the
 * device does not idle while the host collects the count and hands the edges on (bench.py does exactly that). */
int ksp_engine_join_launch(ksp_engine* e, uint64_t tile_begin, uint64_t tile_end, ksp_edge* d_edges, uint64_t capacity, void* stream);
int ksp_engine_join_wait(ksp_engine* e, uint64_t* h_count);
/* A pipelined caller cannot retry an overflowed join: KSP_E_OVERFLOW is only reported by _wait, and by then the next
 * build has replaced the lists (and the tile numbering) the join ran on.  Size the buffer from ksp_engine_edge_bound
 * before _launch — the bound is exact enough to allocate by — and treat KSP_E_OVERFLOW from _wait as "this step's
 * result is incomplete, run the step again".                                                                     */

/* One step of a pipelined caller in one call: ksp_engine_build_blocks, the tile range of rank `part` of `nparts`
 * (ksp_engine_balanced_cuts; returned in range[0..1]), its edge bound (*bound) and ksp_engine_join_launch on that range
 * into d_edges — nothing between the build and the launch of its join but the cutting of the work list.
 * KSP_E_OVERFLOW: *bound + 1 > capacity, nothing was launched (grow the buffer, then ksp_engine_join_launch).
 * A join launched on this engine before the call (the previous step's) is collected on the way — it ran in front of
 * this build on the stream: *prev_count / *prev_status are what ksp_engine_join_wait would have returned for it,
 * *prev_ms_join (may be NULL) its kernel time.
 * Collect the join launched here with ksp_engine_join_wait, or with the next ksp_engine_step_launch.
 * A step launched this way puts NO timing events into the stream (an event record is a ~6 us bubble between two
 * kernels; a step had eleven): ksp_stats.ms_build / ms_sort / ms_join and *prev_ms_join of such a step read 0 unless
 * ksp_engine_set_profiling(e, 1) is on.  The host learns of the read-backs, of the work list's inputs and of the
 * join's count through sequence numbers in pinned memory that it polls.                                              */
int ksp_engine_step_launch(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights, const uint64_t* h_offsets,
                           uint32_t n_sources, int key_bits, uint32_t part, uint32_t nparts, ksp_edge* d_edges,
                           uint64_t capacity, uint64_t range[2], uint64_t* bound, uint64_t* prev_count, int* prev_status,
                           float* prev_ms_join, void* stream);

/* Join tiles [tile_begin, tile_end) and deliver the edges in HOST memory (h_edges: `capacity` edges, pinned memory
 * for full PCIe rate): the range is cut into pieces by the edge bound, piece k + 1 is joined while piece k is copied
 * on a stream of the engine's own.  For results of hundreds of MB (100k genomes: 45 M pairs), whose copy would
 * otherwise follow the join.  *h_count = pairs found; KSP_E_OVERFLOW if that exceeds capacity (*h_count is still the
 * full count: allocate and call again).  Returns when everything has arrived.  The pair map of the reference lives
 * in host memory throughout (src/pairwise.cpp:191); this is the step that gets the device's result there.          */
int ksp_engine_join_to_host(ksp_engine* e, uint64_t tile_begin, uint64_t tile_end, ksp_edge* h_edges, uint64_t capacity,
                            uint64_t* h_count, void* stream);

int ksp_engine_get_stats(const ksp_engine* e, ksp_stats* out);
/* Per-phase HIP-event times of stage 1 (the reference's own phase timers, src/pairwise.cpp:131-133,155,181,
 * 239, print wall-clock seconds per phase; this is their device-side counterpart).  set_profiling(1) makes
 * every later build record one event per phase start; phase_times returns the phases of the last build:
 * names[i] (static strings) and ms[i], at most cap of them; the return value is the count.               */
int ksp_engine_set_profiling(ksp_engine* e, int on);
int ksp_engine_phase_times(const ksp_engine* e, const char** names, float* ms, int cap);

/* Stage 1 from an inverted index instead of sketches: key k is held by the sources
 * d_sources[h_key_off[k] .. h_key_off[k+1]) (dense source indices, distinct inside a key, at least two per
 * key; keys in any order) and weighs d_key_weights[k] (NULL: 1).  This is exactly the reference's input —
 * `_color_to_sources.bin` + `_color_count.bin`, src/pairwise.cpp:128-170 — so the drop-in path hands its
 * colours over without transposing them into per-source runs and skips the engine's sort and prune.
 * The caller guarantees that every source's weights sum to less than 2^32.  Then ksp_engine_join as usual. */
int ksp_engine_build_postings(ksp_engine* e, const uint64_t* h_key_off, const uint32_t* d_sources,
                              const uint32_t* d_key_weights, uint32_t n_keys, uint32_t n_sources, void* stream);
/* One SLICE of such an index: any subset of its keys, every key with all its holders (same arguments, the offsets
 * starting at 0; fewer than 2^30 memberships per slice).  Stops at the source labels; from there the calls of a
 * key-range slice of sketches apply (ksp_engine_slice_labels, MIN over the slices, ksp_engine_slice_finish, _sizes,
 * _export, ksp_engine_assemble).  The reference loads an index of any size (src/pairwise.cpp:95-111): kspider_pairwise
 * cuts one of 2^30 memberships or more into such slices by itself, and with $KSPIDER_DEVICES every device builds the
 * slice of 1 / n of the colours.                                                                                  */
int ksp_engine_build_postings_slice(ksp_engine* e, const uint64_t* h_key_off, const uint32_t* d_sources,
                                    const uint32_t* d_key_weights, uint32_t n_keys, uint32_t n_sources, void* stream);

/* ---- stage 1 sharded over GPUs (one process per GPU) ---------------------------------------
 * Every rank holds the full sketch set; rank `part` of `nparts` sorts and prunes the keys of its
 * 1/nparts share of the hash range only (ksp_engine_build_slice).  The engine orders the sources by a
 * label derived from the shared keys, so the ranks first combine their labels (ksp_engine_slice_labels,
 * element-wise MIN all-reduce of n_sources uint32 over RCCL, ksp_engine_slice_finish) and then build
 * their slices of the block lists in that common order.  The slices are exchanged by the caller
 * (all-gather: kspider_amd/dist.py) and every rank turns the gathered slices into the full block lists
 * (ksp_engine_assemble) before ksp_engine_join.
 *   slice_labels: copies the slice's n_sources labels into a device buffer.
 *   slice_finish: d_labels = the combined labels (device; NULL keeps the slice's own: single slice
 *                tests only — every rank must use the same labels).
 *   slice_sizes: out[0] padded list length L (uint32 entries of d_brk / d_info / d_bw),
 *                out[1] distinct keys, out[2] 128-bit posting masks, out[3] block keys.
 *   slice_export: copies the slice into caller buffers (device pointers): d_brk/d_info[/d_bw] L
 *                entries, d_blk_raw/d_blk_pos n_blocks+1 entries, d_big out[2] x 16 bytes.
 *   assemble:    h_sizes = nparts x 4 values as reported by slice_sizes (host); the *_all buffers
 *                hold the parts back to back with strides lstride (entries) / n_blocks+1 /
 *                bigstride (16-byte masks); must run on an engine that built one of the slices. */
int ksp_engine_build_slice(ksp_engine* e, const uint64_t* d_keys, const uint32_t* d_weights,
                           const uint64_t* h_offsets, uint32_t n_sources, int key_bits, uint32_t part,
                           uint32_t nparts, void* stream);
int ksp_engine_slice_labels(ksp_engine* e, uint32_t* d_labels, void* stream);
int ksp_engine_slice_finish(ksp_engine* e, const uint32_t* d_labels, void* stream);
int ksp_engine_slice_sizes(const ksp_engine* e, uint64_t out[4]);
int ksp_engine_slice_export(ksp_engine* e, uint32_t* d_brk, uint32_t* d_info, uint32_t* d_bw, uint32_t* d_blk_raw,
                            uint32_t* d_blk_pos, void* d_big, void* stream);
int ksp_engine_assemble(ksp_engine* e, uint32_t nparts, const uint64_t* h_sizes, const uint32_t* d_brk_all,
                        const uint32_t* d_info_all, const uint32_t* d_bw_all, uint64_t lstride,
                        const uint32_t* d_blk_raw_all, const uint32_t* d_blk_pos_all, const void* d_big_all,
                        uint64_t bigstride, void* stream);

/* ---- thin device-memory helpers so that FFI callers need no HIP binding ------------- */
int ksp_device_malloc(int device, uint64_t bytes, void** d_ptr);
int ksp_device_free(void* d_ptr);
int ksp_memcpy_h2d(void* d_dst, const void* h_src, uint64_t bytes);
int ksp_memcpy_d2h(void* h_dst, const void* d_src, uint64_t bytes);

/* ---- host-buffer convenience (H2D + both stages + D2H), edges sorted by (s1, s2) ---- */
int ksp_pairwise_host(const uint64_t* keys, const uint32_t* weights, const uint64_t* offsets, uint32_t n_sources,
                      int device, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats);
/* Same for an inverted index in host memory (see ksp_engine_build_postings). */
int ksp_pairwise_postings_host(const uint64_t* key_off, const uint32_t* sources, const uint32_t* key_weights,
                               uint32_t n_keys, uint32_t n_sources, int device, ksp_edge** out_edges,
                               uint64_t* n_edges, ksp_stats* stats);
void ksp_free(void* p);
/* The same two jobs on several GPUs of one node, one host thread and one engine per device (devices[] may name a
 * device twice: two engines share it — used by the single-GPU tests).  This is what $KSPIDER_DEVICES selects
 * behind kspider_pairwise() / kSpider::pairwise() (north star: the N x N pair space sharded over the GPUs of a
 * node): sketch input is built in hash-range slices, one per device, exchanged device to device; an inverted
 * index is built on every device; every device joins a tile range of equal estimated work; the edges are
 * gathered to devices[0] (peer copies over xGMI), sorted there and returned in pinned host memory.        */
int ksp_pairwise_host_multi(const uint64_t* keys, const uint32_t* weights, const uint64_t* offsets, uint32_t n_sources,
                            const int* devices, int n_devices, ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats);
int ksp_pairwise_postings_host_multi(const uint64_t* key_off, const uint32_t* sources, const uint32_t* key_weights,
                                     uint32_t n_keys, uint32_t n_sources, const int* devices, int n_devices,
                                     ksp_edge** out_edges, uint64_t* n_edges, ksp_stats* stats);

/* ---- the reference entry point ------------------------------------------------------
 * Same contract as kSpider::pairwise(string index_prefix, int user_threads)
 * (include/kSpider.hpp:11): reads PREFIX_color_to_sources.bin, PREFIX_color_count.bin,
 * PREFIX_groupID_to_kmerCount.bin, writes PREFIX_kSpider_seqToKmersNo.tsv and
 * PREFIX_kSpider_pairwise.tsv.  Device = $KSPIDER_DEVICE (default 0); $KSPIDER_DEVICES=0,1,...
 * shards the job over several GPUs (ksp_pairwise_postings_host_multi) — same output files.      */
int kspider_pairwise(const char* index_prefix, int user_threads);

/* ---- direct sketch inputs ----------------------------------
 * kspider_pairwise_sigs: sourmash signatures (every .sig file of DIR) -> pairwise TSVs in one go; what
 *   kSpider::sourmash_sigs_indexing(sigs_dir, kSize) (include/kSpider.hpp:18,
 *   src/sourmash_indexing.cpp:52) followed by kSpider::pairwise() computes, with the same group-ID
 *   assignment (glob order), first signature with ksize == kSize, PREFIX.namesMap format.
 * kspider_pairwise_bins: sketches stored as phmap::flat_hash_set<uint64_t> dumps, every .bin file of DIR (the
 *   input format of kSpider::bins_indexing, src/bins_indexing.cpp:98-182).
 * out_prefix NULL/"" -> basename(DIR) in the current directory, as the reference does.
 * Writes PREFIX.namesMap, PREFIX_kSpider_seqToKmersNo.tsv, PREFIX_kSpider_pairwise.tsv.        */
int kspider_pairwise_sigs(const char* sigs_dir, int kSize, const char* out_prefix, int user_threads);
int kspider_pairwise_bins(const char* bins_dir, const char* out_prefix, int user_threads);

/* ---- clustering -------------------------------------------------------
 * kspider_cluster: what `kSpider cluster -i PREFIX -d DIST -c CUTOFF` does (pykSpider/kSpider2/
 *   ks_clustering.py:63-137, 150-163): reads PREFIX.namesMap, PREFIX_kSpider_seqToKmersNo.tsv and
 *   PREFIX_kSpider_pairwise.tsv (dist_type "ani": also PREFIX_kSpider_pairwise.ani_col.tsv), keeps the rows
 *   whose column dist_type ("min_cont" 3, "avg_cont" 4, "max_cont" 5; NULL = "max_cont") times 100 is not
 *   below cutoff * 100 (cutoff in [0, 1]), finds the connected components ON THE GPU and writes one line of
 *   comma-separated names per component to PREFIX_kSpider_clusters_<cutoff*100>%.tsv.  Components are
 *   written in order of their smallest node, names in node order (the reference: rustworkx set order).
 * ksp_components: the device part alone — connected components of an undirected edge list (host arrays
 *   of node indices < n_nodes); h_label[v] = smallest node index of v's component.                      */
int kspider_cluster(const char* index_prefix, const char* dist_type, double cutoff);
int ksp_components(int device, uint32_t n_nodes, const uint32_t* h_a, const uint32_t* h_b, uint64_t n_edges,
                   uint32_t* h_label);
/* Clustering from HBM — what SURVEY 8f N4 is for: the pairs never leave the device as text.
 * ksp_components_edges: components straight over the join's edge records.  d_edges: `n_edges` ksp_edge records in
 *   DEVICE memory (as ksp_engine_join leaves them; node = source index), d_kmer_counts[v] = k-mer count of source v
 *   (device memory).  An edge counts when its containment column dist_col (3 min, 4 avg, 5 max; single-precision maths
 *   of src/pairwise.cpp:260-264) passes the reference's test: text of the float with 6 significant digits -> float ->
 *   x 100 -> not below cutoff x 100 (ks_clustering.py:101-105; a NaN passes).  That test is monotone in the float, so
 *   the device compares against the one critical float found on the host — the same rows pass, digit for digit.
 *   h_label[v] = smallest source index of v's component.
 * kspider_pairwise_and_cluster: `kSpider pairwise` followed by `kSpider cluster` (ks_clustering.py:63-137) in ONE
 *   device pass: writes PREFIX_kSpider_seqToKmersNo.tsv and PREFIX_kSpider_pairwise.tsv exactly as kspider_pairwise
 *   and PREFIX_kSpider_clusters_<cutoff*100>%.tsv exactly as kspider_cluster would from that TSV — but the components
 *   come from the edges while they are in HBM (the TSV is never read back).  dist_type: "min_cont", "avg_cont",
 *   "max_cont" (NULL / ""); "ani" needs the separate ANI column file and stays with kspider_cluster.  Reads
 *   PREFIX.namesMap like kspider_cluster.                                                                          */
int ksp_components_edges(int device, uint32_t n_nodes, const ksp_edge* d_edges, uint64_t n_edges, const uint32_t* d_kmer_counts,
                         int dist_col, double cutoff, uint32_t* h_label);
int kspider_pairwise_and_cluster(const char* index_prefix, int user_threads, const char* dist_type, double cutoff);

/* ---- host-only diagnostics (no GPU needed) -------------------------------------------
 * ksp_index_info: parse the three index files and report what the reader detected:
 * out[0] colours, out[1] groups, out[2] colour-count entries, out[3] sum of sources over
 * colours, out[4] phmap Group::kWidth (16/8), out[5] 1 if a growth_left trailer is present.
 * ksp_format_float: text of a float as `std::ostream << float` prints it
 * (src/pairwise.cpp:266-273); buf must hold 32 bytes; returns the length.             */
int ksp_index_info(const char* index_prefix, uint64_t out[6]);
int ksp_format_float(float value, char* buf);

#ifdef __cplusplus
}
#endif
#endif /* KSPIDER_AMD_H */
