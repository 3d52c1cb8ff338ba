// Internal declarations shared by the engine (engine.hip) and the host-side
// reference surface (pairwise_host.cpp).  Not part of the C ABI.
#ifndef KSPIDER_ENGINE_INTERNAL_H
#define KSPIDER_ENGINE_INTERNAL_H
#include <cstdint>
#include <string>

#include "../../include/kspider_amd.h"

#include <vector>

namespace ksp {
void set_error(const std::string& s);
// GPUs of a drop-in call: $KSPIDER_DEVICES ("0,1,2,..."; a device may appear twice) or the one of $KSPIDER_DEVICE (0)
std::vector<int> devices_from_env();

// ---- clustering straight from the join's edges (cluster.hip) ----
// the smallest float the reference's threshold test lets through (mode 1: only NaN rows pass)
void cc_critical(double cutoff, float* vcrit, int* mode);
// connected components of the kept edges among d_edges (device memory of the CURRENT device): h_label[v] = smallest
// node of v's component; d_cnt[v] = k-mer count of node v; col 3 / 4 / 5 = min / avg / max containment
int cc_edges_on_device(uint32_t n_nodes, const ksp_edge* d_edges, uint64_t n_edges, const uint32_t* d_cnt, int col, double cutoff,
                       uint32_t* h_label, uint64_t* n_kept);
void read_names_map(const std::string& prefix, std::vector<std::string>& name_of);
void write_cluster_file(const std::string& prefix, double threshold, const std::vector<uint32_t>& label,
                        const std::vector<std::string>& name_of);
// a drop-in call that also wants the components of its result, taken from the edges while they are in HBM
struct CcRequest {
    const uint32_t* kmer_counts = nullptr;   // per (dense) source index
    int col = 0;                             // 3 / 4 / 5
    double cutoff = 0;
    std::vector<uint32_t>* labels = nullptr; // out: per source index, the smallest index of its component
    uint64_t n_kept = 0;                     // out: edges that passed the cut
};
int pairwise_postings_multi_cc(const uint64_t* key_off, const uint32_t* sources, const uint32_t* key_weights, uint32_t n_keys,
                               uint32_t n_sources, const int* devices, int n_devices, ksp_edge** out_edges, uint64_t* n_edges,
                               ksp_stats* stats, CcRequest* cc);
}

extern "C" {
/* test/diagnostic hook: distinct-key offsets of the block lists (nb + 1 values). */
int ksp_engine_block_key_counts(const ksp_engine* e, uint32_t* h_blk_off);
int ksp_engine_source_order(const ksp_engine* e, uint32_t* h_newidx);   // (diagnostics) engine index of every source
}
#endif
