/* ORACLE — TEST INFRASTRUCTURE ONLY (see ref_pairwise.cpp header).
 *
 * phmap raw-table dump layout, restated from parallel-hashmap's phmap_dump.h
 * (un-vendored dependency of the reference; lib/parallel-hashmap is an empty
 * submodule, no pinned version recoverable — WIRE FORMAT PARITY UNPINNED):
 *
 *   u64 size; u64 capacity;                     (capacity = 2^k - 1)
 *   if size > 0:
 *     int8 ctrl[capacity + kWidth + 1];         (full slot <=> ctrl >= 0; empty -128,
 *                                                deleted -2, sentinel -1; kWidth = 16
 *                                                with SSE2, 8 otherwise)
 *     slot  slots[capacity];                    (4 B flat_hash_set<u32>; 8 B {u32,u32};
 *                                                16 B {u64,u64})
 *     u64 growth_left;                          (newer phmap releases only: "trailer")
 *
 * _color_to_sources.bin = u64 C, then C x { u64 colour; dump of flat_hash_set<u32> }
 * (writers: /root/reference/src/index.cpp:336-363; readers: src/pairwise.cpp:95-121,166-170).
 */
#ifndef KSPIDER_ORACLE_H
#define KSPIDER_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_edge {
    uint32_t source_1, source_2;
    uint64_t shared;
} oracle_edge;

const char* oracle_last_error(void);

/* Restatement of kSpider::pairwise (src/pairwise.cpp:123-276) on index files. */
int oracle_ref_pairwise(const char* index_prefix, int user_threads, int kwidth, int trailer, int sorted_rows,
                        double* secs_accumulate, uint64_t* n_edges, uint64_t* n_updates);

/* Same accumulation (src/pairwise.cpp:194-237) on an in-memory colour CSR. */
int oracle_accumulate_mem(const uint32_t* color_off, const uint32_t* sources, const uint32_t* color_w,
                          uint32_t n_colors, int user_threads, double* secs_accumulate, uint64_t* n_edges,
                          uint64_t* n_updates, oracle_edge* out_edges, uint64_t out_capacity);

/* Colour index from sorted-unique sketches (semantics of src/index.cpp:189-331:
 * one colour per distinct source-membership set, colour weight = #k-mers with
 * that membership).  Group IDs are group_ids[s] (NULL -> s+1).  Output CSR is
 * malloc'ed; free with oracle_free(). */
int oracle_build_colors(const uint64_t* keys, const uint64_t* offsets, uint32_t n_sources,
                        const uint32_t* group_ids, uint32_t** color_off, uint32_t** sources,
                        uint32_t** color_w, uint32_t* n_colors);
void oracle_free(void* p);

/* Write the three .bin files + .namesMap in the (restated) phmap dump layout. */
int oracle_write_index(const char* index_prefix, const uint32_t* color_off, const uint32_t* sources,
                       const uint32_t* color_w, uint32_t n_colors, const uint32_t* group_ids,
                       const uint32_t* kmer_counts, uint32_t n_sources, int kwidth, int trailer,
                       uint64_t slot_seed);

/* One sketch as a phmap::flat_hash_set<uint64_t> dump (".bin", src/bins_indexing.cpp:178-180). */
int oracle_write_bin_sketch(const char* path, const uint64_t* hashes, uint64_t n, int kwidth, int trailer,
                            uint64_t slot_seed);

/* Brute force, the semantics of test/generate_golden_files.py:40-49:
 * shared = |A ∩ B| for every a < b, non-zero pairs only, sorted by (a, b).
 * IDs are dense indices 0..n-1.  Returns the number of edges (or -1 if the
 * buffer is too small). */
int64_t oracle_brute_pairs(const uint64_t* keys, const uint64_t* offsets, uint32_t n_sources, oracle_edge* out,
                           uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif
