// kSpider::pairwise() drop-in: host side.
//
// Mirrors /root/reference/src/pairwise.cpp:123-276 phase by phase:
//   :127-129  load colour -> sources            -> ksp::load_index
//   :136-137  load colour counts                -> ksp::load_index
//   :166-181  load k-mer counts, write seqToKmersNo.tsv
//   :194-237  accumulate shared k-mers per pair -> MI355X engine (engine.hip)
//   :242-275  write pairwise.tsv (float maths + formatting on the host)
// The same progress lines go to stdout.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kSpider.hpp"
#include "../../include/kspider_amd.h"
#include "engine_internal.h"
#include "index_io.h"

namespace ksp {
std::vector<int> devices_from_env() {
    std::vector<int> out;
    if (const char* ds = std::getenv("KSPIDER_DEVICES")) {
        const char* p = ds;
        while (*p) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            out.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') break;
        }
    }
    if (out.empty()) {
        int device = 0;
        if (const char* d = std::getenv("KSPIDER_DEVICE")) device = std::atoi(d);
        out.push_back(device);
    }
    return out;
}
}  // namespace ksp

namespace {

typedef std::chrono::high_resolution_clock Clock;
double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// dist_type != nullptr: also cluster (kSpider cluster, ks_clustering.py:63-137) from the edges while they are on the device
int run_pairwise(const std::string& prefix, int user_threads, const char* dist_type = nullptr, double cutoff = 0) {
    int cc_col = 0;
    if (dist_type) {
        const std::string dt = *dist_type ? dist_type : "max_cont";
        cc_col = dt == "min_cont" ? 3 : dt == "avg_cont" ? 4 : dt == "max_cont" ? 5 : 0;
        if (!cc_col) {
            ksp::set_error("kspider_pairwise_and_cluster: distance '" + dt + "' is not min_cont, avg_cont or max_cont (ani needs the separate ANI column file: run kspider_cluster)");
            return KSP_E_ARG;
        }
    }
    auto t0 = Clock::now();
    ksp::IndexData ix;
    ksp::load_index(prefix, ix);
    std::cout << "mapping colors to groups: " << since(t0) << " secs" << std::endl;
    t0 = Clock::now();
    std::cout << "parsing index colors: " << since(t0) << " secs" << std::endl;
    t0 = Clock::now();
    ksp::write_seq_to_kmers(prefix, ix);
    std::unordered_map<uint32_t, uint32_t> kmer_count;
    for (auto& s : ix.kmer_slots) kmer_count[s.first] = s.second;
    std::cout << "kmer counting: " << since(t0) << " secs" << std::endl;

    t0 = Clock::now();
    // dense source index = rank of the group ID, so that index order == ID order and the
    // engine's (i < j) is the reference's ascending(source_1, source_2) (:73-78, :218)
    uint32_t max_id = 0;
    for (auto& c : ix.colors)
        for (uint32_t g : c.second) max_id = std::max(max_id, g);
    std::vector<uint32_t> ids;
    std::vector<uint32_t> dense_of;          // direct table when the ID space is small enough
    if (max_id < (1u << 28)) {
        std::vector<uint8_t> seen((size_t)max_id + 1, 0);
        for (auto& c : ix.colors)
            for (uint32_t g : c.second) seen[g] = 1;
        dense_of.assign((size_t)max_id + 1, 0);
        for (uint32_t g = 0; g <= max_id; ++g)
            if (seen[g]) { dense_of[g] = (uint32_t)ids.size(); ids.push_back(g); }
        if (ix.colors.empty()) ids.clear();
    } else {
        for (auto& c : ix.colors) ids.insert(ids.end(), c.second.begin(), c.second.end());
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    }
    const uint32_t N = (uint32_t)ids.size();
    auto dense = [&](uint32_t g) -> uint32_t {
        return dense_of.empty() ? (uint32_t)(std::lower_bound(ids.begin(), ids.end(), g) - ids.begin()) : dense_of[g];
    };

    // The colour index IS an inverted index (colour -> sources, src/pairwise.cpp:128-170): hand it to the
    // engine as postings — no transposition into per-source runs, no sort and prune on the device.
    std::vector<std::pair<uint32_t, uint32_t>> zero_pairs;   // pairs touched only through weight-0 colours
    std::vector<uint64_t> wsum((size_t)N, 0);
    std::vector<uint64_t> key_off;
    std::vector<uint32_t> post_src, key_w;
    key_off.push_back(0);
    {
        size_t total = 0;
        for (auto& c : ix.colors) total += c.second.size() >= 2 ? c.second.size() : 0;
        post_src.reserve(total);
        key_off.reserve(ix.colors.size() + 1);
        key_w.reserve(ix.colors.size());
    }
    for (size_t ci = 0; ci < ix.colors.size(); ++ci) {
        auto& c = ix.colors[ci];
        uint32_t w = 0;
        if (!ix.colors_count.find(c.first, w)) w = 0;   // colorsCount[item.first] (:221): 0 when absent
        if (c.second.size() < 2) continue;          // a colour with one source produces no pair
        if (w == 0) {
            // the reference still creates the pair entries (with += 0): remember them
            for (size_t x = 0; x < c.second.size(); ++x)
                for (size_t y = x + 1; y < c.second.size(); ++y) {
                    uint32_t a = c.second[x], b = c.second[y];
                    if (a > b) std::swap(a, b);
                    if (a != b) zero_pairs.emplace_back(a, b);
                }
            continue;
        }
        for (uint32_t g : c.second) {
            const uint32_t di = dense(g);
            post_src.push_back(di);
            wsum[di] += w;
        }
        key_off.push_back(post_src.size());
        key_w.push_back(w);
    }
    for (uint32_t s = 0; s < N; ++s)
        if (wsum[s] >= (1ull << 32))
            throw std::runtime_error("kspider_amd: colour weights of group " + std::to_string(ids[s]) +
                                     " sum to >= 2^32 (32-bit pair counters would overflow)");
    const uint64_t E = post_src.size();
    // (a source cannot repeat inside a colour: flat_hash_set; two colours narrowing to the same uint32 id
    //  were resolved by load_index the way insert_or_assign does)

    const double t_transpose = since(t0);
    const std::vector<int> devices = ksp::devices_from_env();
    ksp_edge* edges = nullptr;
    uint64_t n_edges = 0;
    ksp_stats st;
    auto t1 = Clock::now();
    ksp::CcRequest cc;
    std::vector<uint32_t> cc_counts, cc_labels;
    if (cc_col) {
        cc_counts.resize(N);
        for (uint32_t i = 0; i < N; ++i) {
            auto it = kmer_count.find(ids[i]);
            cc_counts[i] = it == kmer_count.end() ? 0u : it->second;   // (a missing group counts 0 k-mers, as operator[] of the reference yields)
        }
        cc.kmer_counts = cc_counts.data(); cc.col = cc_col; cc.cutoff = cutoff; cc.labels = &cc_labels;
    }
    int rc = ksp::pairwise_postings_multi_cc(key_off.data(), post_src.data(), key_w.data(), (uint32_t)key_w.size(), N,
                                             devices.data(), (int)devices.size(), &edges, &n_edges, &st, cc_col ? &cc : nullptr);
    const double t_device = since(t1);
    if (rc != KSP_OK) return rc;
    std::vector<ksp::EdgeRow> rows;
    rows.reserve(n_edges + zero_pairs.size());
    for (uint64_t i = 0; i < n_edges; ++i)
        rows.push_back(ksp::EdgeRow{ids[edges[i].source_1], ids[edges[i].source_2], edges[i].shared});
    ksp_free(edges);
    if (!zero_pairs.empty()) {
        std::sort(zero_pairs.begin(), zero_pairs.end());
        zero_pairs.erase(std::unique(zero_pairs.begin(), zero_pairs.end()), zero_pairs.end());
        const size_t nreal = rows.size();
        for (auto& zp : zero_pairs) {
            auto it = std::lower_bound(rows.begin(), rows.begin() + nreal, zp,
                                       [](const ksp::EdgeRow& r, const std::pair<uint32_t, uint32_t>& k) {
                                           return r.source_1 != k.first ? r.source_1 < k.first : r.source_2 < k.second;
                                       });
            if (it == rows.begin() + nreal || it->source_1 != zp.first || it->source_2 != zp.second)
                rows.push_back(ksp::EdgeRow{zp.first, zp.second, 0});
        }
        std::sort(rows.begin(), rows.end(), [](const ksp::EdgeRow& a, const ksp::EdgeRow& b) {
            return a.source_1 != b.source_1 ? a.source_1 < b.source_1 : a.source_2 < b.source_2;
        });
    }
    std::cout << "pairwise hashmap construction: " << since(t0) << " secs" << std::endl;
    if (std::getenv("KSPIDER_VERBOSE"))
        std::cout << "kspider_amd: postings from the colour index " << t_transpose << " s, device round trip " << t_device
                  << " s (stage 1 " << st.ms_build << " ms, join " << st.ms_join << " ms)" << std::endl;
    std::cout << "writing pairwise matrix to " << prefix << "_kSpider_pairwise.tsv" << std::endl;
    ksp::write_pairwise_tsv(prefix, rows, kmer_count, user_threads);
    if (std::getenv("KSPIDER_VERBOSE"))
        std::cout << "kspider_amd: sources=" << N << " colour-entries=" << E << " pairs=" << rows.size() << std::endl;
    if (cc_col) {
        // the cluster file of `kSpider cluster` from the components the device found on the join's own edge records
        std::vector<std::string> name_of;
        ksp::read_names_map(prefix, name_of);
        const uint64_t NN = name_of.size();
        if (cc_labels.size() != N) cc_labels.assign(N, 0);   // (no edges at all: the device pass did not run)
        if (n_edges == 0) for (uint32_t i = 0; i < N; ++i) cc_labels[i] = i;
        // rows that only exist with shared_kmers = 0 (colours of weight 0) are rows of the TSV too: the same test, on the host
        if (!zero_pairs.empty()) {
            float vcrit = 0;
            int mode = 0;
            ksp::cc_critical(cutoff, &vcrit, &mode);
            std::vector<uint32_t> parent(cc_labels);
            auto find = [&](uint32_t v) { while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; } return v; };
            bool merged = false;
            for (auto& zp : zero_pairs) {
                auto it = std::lower_bound(rows.begin(), rows.end(), zp, [](const ksp::EdgeRow& r, const std::pair<uint32_t, uint32_t>& k) {
                    return r.source_1 != k.first ? r.source_1 < k.first : r.source_2 < k.second;
                });
                if (it == rows.end() || it->source_1 != zp.first || it->source_2 != zp.second || it->shared != 0) continue;   // (the pair also shares a weighted colour: an ordinary row)
                const uint32_t a = dense(zp.first), b = dense(zp.second);
                const float n1 = (float)cc_counts[a], n2 = (float)cc_counts[b];
                const float c12 = 0.0f / n2, c21 = 0.0f / n1;
                const float v = cc_col == 3 ? std::min(c12, c21) : cc_col == 5 ? std::max(c12, c21) : (float)((c12 + c21) / 2.0);
                const bool kept = mode ? v != v : !(v < vcrit);
                if (!kept) continue;
                const uint32_t ra = find(a), rb = find(b);
                if (ra != rb) { parent[std::max(ra, rb)] = std::min(ra, rb); merged = true; }
            }
            if (merged) for (uint32_t i = 0; i < N; ++i) cc_labels[i] = find(i);
        }
        std::vector<uint32_t> node_label((size_t)NN);
        for (uint64_t v = 0; v < NN; ++v) node_label[v] = (uint32_t)v;
        for (uint32_t i = 0; i < N; ++i) {
            if (cc_labels[i] == i) continue;   // (a root, or a source without a kept edge)
            const uint64_t a = ids[i], b = ids[cc_labels[i]];
            if (a < 1 || b < 1 || a > NN || b > NN)
                throw std::runtime_error("pairwise row names node " + std::to_string(std::max(a, b)) + " but .namesMap has " + std::to_string(NN) + " rows (ids must be 1..N)");
            node_label[a - 1] = (uint32_t)(b - 1);
        }
        ksp::write_cluster_file(prefix, cutoff * 100.0, node_label, name_of);
        if (std::getenv("KSPIDER_VERBOSE")) std::cout << "kspider_amd: clusters from " << cc.n_kept << " edges that pass the cut" << std::endl;
    }
    return KSP_OK;
}

}  // namespace

extern "C" int kspider_pairwise(const char* index_prefix, int user_threads) {
    if (!index_prefix) {
        ksp::set_error("kspider_pairwise: index_prefix is NULL");
        return KSP_E_ARG;
    }
    try {
        return run_pairwise(index_prefix, user_threads < 1 ? 1 : user_threads);
    } catch (const std::bad_alloc&) {
        ksp::set_error("kspider_pairwise: out of host memory");
        return KSP_E_LIMIT;
    } catch (const std::exception& e) {
        ksp::set_error(e.what());
        const std::string m = e.what();
        return m.find("2^32") != std::string::npos ? KSP_E_LIMIT : KSP_E_IO;
    }
}

extern "C" int kspider_pairwise_and_cluster(const char* index_prefix, int user_threads, const char* dist_type, double cutoff) {
    if (!index_prefix) {
        ksp::set_error("kspider_pairwise_and_cluster: index_prefix is NULL");
        return KSP_E_ARG;
    }
    try {
        return run_pairwise(index_prefix, user_threads < 1 ? 1 : user_threads, dist_type ? dist_type : "", cutoff);
    } catch (const std::bad_alloc&) {
        ksp::set_error("kspider_pairwise_and_cluster: out of host memory");
        return KSP_E_LIMIT;
    } catch (const std::exception& e) {
        ksp::set_error(e.what());
        const std::string m = e.what();
        return m.find("2^32") != std::string::npos ? KSP_E_LIMIT : KSP_E_IO;
    }
}

extern "C" int ksp_index_info(const char* index_prefix, uint64_t out[6]) {
    if (!index_prefix || !out) {
        ksp::set_error("ksp_index_info: NULL argument");
        return KSP_E_ARG;
    }
    try {
        ksp::IndexData ix;
        ksp::load_index(index_prefix, ix);
        uint64_t m = 0;
        for (auto& c : ix.colors) m += c.second.size();
        out[0] = ix.colors.size();
        out[1] = ix.kmer_slots.size();
        out[2] = ix.colors_count.size();
        out[3] = m;
        out[4] = (uint64_t)ix.kwidth;
        out[5] = ix.trailer ? 1 : 0;
        return KSP_OK;
    } catch (const std::exception& e) {
        ksp::set_error(e.what());
        return KSP_E_IO;
    }
}

extern "C" int ksp_format_float(float value, char* buf) { return buf ? ksp::format_float(buf, value) : 0; }

namespace kSpider {
void pairwise(std::string index_prefix, int user_threads) {
    if (kspider_pairwise(index_prefix.c_str(), user_threads) != KSP_OK) throw std::runtime_error(ksp_last_error());
}
}  // namespace kSpider
