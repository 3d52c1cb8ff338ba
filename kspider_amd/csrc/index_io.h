// Reader for the reference's on-disk index (phmap binary dumps) and writers for its two
// TSV outputs.  Host-only C++.
#ifndef KSPIDER_INDEX_IO_H
#define KSPIDER_INDEX_IO_H
#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace ksp {

struct IndexData {
    // colour -> sources after the reference's uint32 narrowing + insert_or_assign
    // (src/pairwise.cpp:103,109), in first-insertion (file) order
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> colors;
    // colour -> #k-mers, both narrowed to uint32 (src/pairwise.cpp:119): a direct table when the ids are
    // small (the usual case: colour ids are handed out consecutively), a hash map otherwise
    struct CountTable {
        std::vector<uint32_t> direct;      // value per id
        std::vector<uint8_t> present;      // 1 where the id exists
        std::unordered_map<uint32_t, uint32_t> sparse;
        size_t n = 0;
        bool find(uint32_t k, uint32_t& v) const {
            if (!direct.empty()) {
                if (k >= direct.size() || !present[k]) return false;
                v = direct[k];
                return true;
            }
            auto it = sparse.find(k);
            if (it == sparse.end()) return false;
            v = it->second;
            return true;
        }
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
    } colors_count;
    // groupID -> k-mer count in table slot order (iteration order of src/pairwise.cpp:175)
    std::vector<std::pair<uint32_t, uint32_t>> kmer_slots;
    int kwidth = 16;       // detected phmap Group::kWidth
    bool trailer = false;  // detected trailing growth_left word
};

// Throws std::runtime_error (message names the file) on missing / malformed input.
void load_index(const std::string& prefix, IndexData& out);

// PREFIX_kSpider_seqToKmersNo.tsv  (src/pairwise.cpp:173-180)
void write_seq_to_kmers(const std::string& prefix, const IndexData& ix);

struct EdgeRow {
    uint32_t source_1, source_2;
    uint64_t shared;
};
// PREFIX_kSpider_pairwise.tsv (src/pairwise.cpp:242-275); rows must already be in the
// order they are to be written.  n1/n2 = k-mer counts of the two sources.
void write_pairwise_tsv(const std::string& prefix, const std::vector<EdgeRow>& rows,
                        const std::unordered_map<uint32_t, uint32_t>& kmer_count, int threads);

// One phmap::flat_hash_set<uint64_t> dump (a sketch ".bin": sigs_to_bins.cpp:113-136,
// src/bins_indexing.cpp:178-180) -> its hashes in slot order.
void load_u64_set(const std::string& path, std::vector<uint64_t>& out);

// "%g"-style text of a float exactly as `std::ostream << float` prints it.
int format_float(char* buf, float v);

}  // namespace ksp
#endif
