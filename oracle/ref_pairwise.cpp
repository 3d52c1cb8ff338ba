// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// CPU restatement of the reference hot path kSpider::pairwise()
// (/root/reference/src/pairwise.cpp:123-276) without phmap / Boost, which are
// absent from the reference snapshot (empty submodules) — the reference file
// itself cannot be compiled here.
//
// PARITY PIN: the pair semantics (shared_kmers = |A ∩ B|, non-zero pairs only,
// containment definitions) are pinned against golden vectors produced by the
// reference's own test oracle test/generate_golden_files.py, executed unmodified
// in the build container by tests/golden/make_golden.py (fixtures under
// tests/golden/).  The phmap binary dump layout read below is restated from the
// published parallel-hashmap `phmap_dump.h`; no reference fixture pins it:
// WIRE FORMAT PARITY UNPINNED.
//
// Deliberately NOT improved relative to the reference: inverted-index walk,
// contiguous static colour slices per thread, whole C(m,2) pair list
// materialised per colour, 4096-shard mutex-protected pair map.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "oracle.h"

namespace {

typedef std::chrono::high_resolution_clock Clock;

// ---------------------------------------------------------------------------
// phmap raw-table dump reader (restated; see oracle.h for the layout).
// ---------------------------------------------------------------------------
struct Reader {
    std::vector<unsigned char> buf;
    size_t pos = 0;
    explicit Reader(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("oracle: cannot open " + path);
        f.seekg(0, std::ios::end);
        buf.resize((size_t)f.tellg());
        f.seekg(0);
        f.read((char*)buf.data(), (std::streamsize)buf.size());
    }
    template <class T> T get() {
        if (pos + sizeof(T) > buf.size()) throw std::runtime_error("oracle: short file");
        T v;
        std::memcpy(&v, buf.data() + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
    const unsigned char* take(size_t n) {
        if (pos + n > buf.size()) throw std::runtime_error("oracle: short file");
        const unsigned char* p = buf.data() + pos;
        pos += n;
        return p;
    }
};

// One raw_hash_set dump with SLOT-byte slots; calls fn(slot_ptr) for each full
// slot in slot order (= the container's iteration order after phmap_load).
template <class Fn>
void read_raw_table(Reader& r, size_t slot_bytes, int kwidth, bool trailer, Fn fn) {
    uint64_t size = r.get<uint64_t>();
    uint64_t cap = r.get<uint64_t>();
    if (size == 0) return;
    if (((cap + 1) & cap) != 0 || size > cap) throw std::runtime_error("oracle: bad table header");
    const signed char* ctrl = (const signed char*)r.take(cap + kwidth + 1);
    const unsigned char* slots = r.take(slot_bytes * cap);
    if (trailer) (void)r.get<uint64_t>();  // growth_left
    uint64_t seen = 0;
    for (uint64_t i = 0; i < cap; ++i)
        if (ctrl[i] >= 0) { fn(slots + i * slot_bytes); ++seen; }
    if (seen != size) throw std::runtime_error("oracle: ctrl/size mismatch");
}

// ---------------------------------------------------------------------------
// PAIRS_COUNTER restatement (src/pairwise.cpp:22-27): 4096 submaps, one
// std::mutex each, key pair<u32,u32>, value u64.
// ---------------------------------------------------------------------------
// Open-addressing table, u64 key -> u64 value: power-of-two capacity, linear probing, grown at a load of 7/8 —
// the container CLASS of phmap's flat maps (one contiguous slot array, no per-node allocation), without its
// SSE control-byte groups.  Not an algorithmic improvement over the reference: same updates, same locking.
// (Round 2 used node-based std::unordered_map here, which BASELINE.md section 2 did not describe.)
struct FlatMap {
    static constexpr uint64_t kEmpty = ~0ull;   // (no pair (2^32-1, 2^32-1): source_1 < source_2; no colour key 2^64-1)
    std::vector<uint64_t> keys, vals;
    size_t used = 0, mask = 0;
    static uint64_t mix(uint64_t h) {   // phmap mixes the user's hash before use (128-bit multiply, folded)
        const unsigned __int128 m = (unsigned __int128)h * 0xde5fb9d2630458e9ull;
        return (uint64_t)m + (uint64_t)(m >> 64);
    }
    void grow() {
        const size_t cap = keys.empty() ? 16 : keys.size() * 2;
        std::vector<uint64_t> k(cap, kEmpty), v(cap, 0);
        for (size_t i = 0; i < keys.size(); ++i)
            if (keys[i] != kEmpty) {
                size_t j = (size_t)(mix(keys[i]) >> 12) & (cap - 1);
                while (k[j] != kEmpty) j = (j + 1) & (cap - 1);
                k[j] = keys[i];
                v[j] = vals[i];
            }
        keys.swap(k);
        vals.swap(v);
        mask = cap - 1;
    }
    // value slot of `key`, inserted with `init` when absent (`mixed` = mix(key), computed once by the caller)
    uint64_t& at(uint64_t key, uint64_t mixed, uint64_t init, bool& fresh) {
        if ((used + 1) * 8 > keys.size() * 7) grow();
        size_t j = (size_t)(mixed >> 12) & mask;
        while (keys[j] != kEmpty && keys[j] != key) j = (j + 1) & mask;
        fresh = keys[j] == kEmpty;
        if (fresh) { keys[j] = key; vals[j] = init; ++used; }
        return vals[j];
    }
    const uint64_t* find(uint64_t key) const {
        if (keys.empty()) return nullptr;
        size_t j = (size_t)(mix(key) >> 12) & mask;
        while (keys[j] != kEmpty && keys[j] != key) j = (j + 1) & mask;
        return keys[j] == kEmpty ? nullptr : &vals[j];
    }
};

struct ShardedPairs {
    static constexpr int kShards = 4096;  // N = 12
    struct Shard {
        std::mutex mu;
        FlatMap m;   // key = source_1 << 32 | source_2
    };
    std::vector<Shard> shards{(size_t)kShards};
    // try_emplace_l(key, [c](v){ v.second += c; }, c)   (src/pairwise.cpp:221-225)
    void add(const std::pair<uint32_t, uint32_t>& k, uint32_t c) {
        const uint64_t key = ((uint64_t)k.first << 32) | k.second;
        const uint64_t h = FlatMap::mix(key);   // (boost::hash<pair> in the reference: decides shard and slot only)
        Shard& s = shards[h & (kShards - 1)];   // the low 12 bits pick the submap, the bits above them the slot
        std::lock_guard<std::mutex> g(s.mu);
        bool fresh;
        uint64_t& v = s.m.at(key, h, (uint64_t)c, fresh);
        if (!fresh) v += c;
    }
    uint64_t size() const {
        uint64_t n = 0;
        for (auto& s : shards) n += s.m.used;
        return n;
    }
    template <class Fn> void for_each(Fn fn) const {   // fn(source_1, source_2, shared), table order
        for (auto& s : shards)
            for (size_t i = 0; i < s.m.keys.size(); ++i)
                if (s.m.keys[i] != FlatMap::kEmpty) fn((uint32_t)(s.m.keys[i] >> 32), (uint32_t)s.m.keys[i], s.m.vals[i]);
    }
};

// Combo (src/pairwise.cpp:39-70): all index pairs (j-1, i-1), i = n..2, j = i-1..1.
struct Combo {
    std::vector<std::pair<uint32_t, uint32_t>> combs;
    void combinations(int n) {
        combs.clear();
        for (int i = n; i >= 2; --i)
            for (int j = i - 1; j >= 1; --j) combs.emplace_back((uint32_t)(j - 1), (uint32_t)(i - 1));
    }
};

struct Index {
    // color_to_ids after insert_or_assign with the colour narrowed to uint32
    // (src/pairwise.cpp:103,109); kept in first-insertion order.
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> colors;
    FlatMap colors_count;                                           // :113-121 int_int_map: flat, both narrowed to 32 bits
    std::vector<std::pair<uint32_t, uint32_t>> kmer_count_slots;    // slot order (:175-179)
    std::unordered_map<uint32_t, uint32_t> kmer_count;
};

void load_index(const std::string& prefix, int kwidth, bool trailer, Index& ix) {
    {   // load_colors_to_sources (:95-111)
        Reader r(prefix + "_color_to_sources.bin");
        uint64_t n = r.get<uint64_t>();
        std::unordered_map<uint32_t, size_t> where;
        while (n--) {
            uint64_t k = r.get<uint64_t>();
            std::vector<uint32_t> v;
            read_raw_table(r, 4, kwidth, trailer, [&](const unsigned char* s) {
                uint32_t x;
                std::memcpy(&x, s, 4);
                v.push_back(x);
            });
            uint32_t k32 = (uint32_t)k;
            auto it = where.find(k32);
            if (it == where.end()) {
                where.emplace(k32, ix.colors.size());
                ix.colors.emplace_back(k32, std::move(v));
            } else {
                ix.colors[it->second].second = std::move(v);  // insert_or_assign
            }
        }
        if (r.pos != r.buf.size()) throw std::runtime_error("oracle: trailing bytes in color_to_sources");
    }
    {   // load_colors_count (:113-121)
        Reader r(prefix + "_color_count.bin");
        read_raw_table(r, 16, kwidth, trailer, [&](const unsigned char* s) {
            uint64_t k, v;
            std::memcpy(&k, s, 8);
            std::memcpy(&v, s + 8, 8);
            bool fresh;
            ix.colors_count.at((uint32_t)k, FlatMap::mix((uint32_t)k), 0, fresh) = (uint32_t)v;
        });
        if (!ix.colors_count.used) throw std::runtime_error("oracle: empty color_count (assert :117)");
    }
    {   // groupID_to_kmerCount (:166-170)
        Reader r(prefix + "_groupID_to_kmerCount.bin");
        read_raw_table(r, 8, kwidth, trailer, [&](const unsigned char* s) {
            uint32_t k, v;
            std::memcpy(&k, s, 4);
            std::memcpy(&v, s + 4, 4);
            ix.kmer_count_slots.emplace_back(k, v);
            ix.kmer_count[k] = v;
        });
        if (ix.kmer_count.empty()) throw std::runtime_error("oracle: empty kmerCount (assert :170)");
    }
}

// The timed region of the reference (src/pairwise.cpp:200-239).
double accumulate(Index& ix, int user_threads, ShardedPairs& edges) {
    int n = (int)ix.colors.size();
    int T = std::max(1, user_threads);
    auto t0 = Clock::now();
    auto work = [&](int thread_num) {
        int start = (int)((long long)thread_num * n / T);
        int end = (int)((long long)(thread_num + 1) * n / T);
        for (int vec_i = start; vec_i != end; ++vec_i) {
            auto item = ix.colors[vec_i];  // whole-item copy, as :210
            Combo combo;
            combo.combinations((int)item.second.size());
            for (uint32_t i = 0; i < combo.combs.size(); i++) {
                auto const& seq_pair = combo.combs[i];
                uint32_t s1 = item.second[seq_pair.first];
                uint32_t s2 = item.second[seq_pair.second];
                if (s1 > s2) std::swap(s1, s2);  // ascending() :73-78
                uint32_t ccount = 0;             // colorsCount[item.first] (:221), 0 when absent
                if (const uint64_t* cc = ix.colors_count.find(item.first)) ccount = (uint32_t)*cc;
                edges.add(std::make_pair(s1, s2), ccount);
            }
        }
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    return std::chrono::duration<double>(Clock::now() - t0).count();
}

void write_outputs(const std::string& prefix, Index& ix, ShardedPairs& edges, bool sorted) {
    {   // :173-180
        std::ofstream f(prefix + "_kSpider_seqToKmersNo.tsv");
        f << "ID\tseq\tkmers\n";
        uint64_t counter = 0;
        for (auto& it : ix.kmer_count_slots) f << ++counter << '\t' << it.first << '\t' << it.second << '\n';
    }
    std::vector<std::pair<std::pair<uint32_t, uint32_t>, uint64_t>> rows;
    edges.for_each([&](uint32_t a, uint32_t b, uint64_t v) { rows.push_back({{a, b}, v}); });
    if (sorted) std::sort(rows.begin(), rows.end());
    std::ofstream myfile(prefix + "_kSpider_pairwise.tsv");
    myfile << "source_1"
           << "\tsource_2"
           << "\tshared_kmers"
           << "\tmin_containment"
           << "\tavg_containment"
           << "\tmax_containment" << '\n';
    for (auto& edge : rows) {  // :253-274
        uint64_t shared_kmers = edge.second;
        uint32_t source_1 = edge.first.first;
        uint32_t source_2 = edge.first.second;
        uint32_t source_1_kmers = ix.kmer_count[source_1];
        uint32_t source_2_kmers = ix.kmer_count[source_2];
        float cont_1_in_2 = (float)shared_kmers / source_2_kmers;
        float cont_2_in_1 = (float)shared_kmers / source_1_kmers;
        float min_containment = std::min(cont_1_in_2, cont_2_in_1);
        float avg_containment = (cont_1_in_2 + cont_2_in_1) / 2.0;
        float max_containment = std::max(cont_1_in_2, cont_2_in_1);
        myfile << source_1 << '\t' << source_2 << '\t' << shared_kmers << '\t' << min_containment << '\t'
               << avg_containment << '\t' << max_containment << '\n';
    }
}

thread_local std::string g_err;

}  // namespace

extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

int oracle_ref_pairwise(const char* index_prefix, int user_threads, int kwidth, int trailer, int sorted_rows,
                        double* secs_accumulate, uint64_t* n_edges, uint64_t* n_updates) {
    try {
        Index ix;
        load_index(index_prefix, kwidth, trailer != 0, ix);
        ShardedPairs edges;
        double secs = accumulate(ix, user_threads, edges);
        if (secs_accumulate) *secs_accumulate = secs;
        if (n_edges) *n_edges = edges.size();
        if (n_updates) {
            uint64_t u = 0;
            for (auto& c : ix.colors) u += (uint64_t)c.second.size() * (c.second.size() - 1) / 2;
            *n_updates = u;
        }
        std::cout << "pairwise hashmap construction: " << secs << " secs" << std::endl;
        write_outputs(index_prefix, ix, edges, sorted_rows != 0);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

// In-memory variant for bench.py's cpu_baseline leg: same accumulate(), colours
// handed over as CSR (colour c -> sources[color_off[c]..color_off[c+1]) with
// weight color_w[c]).  Returns the accumulate-only wall time.
int oracle_accumulate_mem(const uint32_t* color_off, const uint32_t* sources, const uint32_t* color_w,
                          uint32_t n_colors, int user_threads, double* secs_accumulate, uint64_t* n_edges,
                          uint64_t* n_updates, oracle_edge* out_edges, uint64_t out_capacity) {
    try {
        Index ix;
        ix.colors.reserve(n_colors);
        uint64_t upd = 0;
        for (uint32_t c = 0; c < n_colors; ++c) {
            std::vector<uint32_t> v(sources + color_off[c], sources + color_off[c + 1]);
            upd += (uint64_t)v.size() * (v.size() - 1) / 2;
            ix.colors.emplace_back(c + 1, std::move(v));
            bool fresh;
            ix.colors_count.at(c + 1, FlatMap::mix(c + 1), 0, fresh) = color_w[c];
        }
        ShardedPairs edges;
        double secs = accumulate(ix, user_threads, edges);
        if (secs_accumulate) *secs_accumulate = secs;
        uint64_t ne = edges.size();
        if (n_edges) *n_edges = ne;
        if (n_updates) *n_updates = upd;
        if (out_edges) {
            if (ne > out_capacity) throw std::runtime_error("oracle: edge buffer too small");
            uint64_t i = 0;
            edges.for_each([&](uint32_t a, uint32_t b, uint64_t v) { out_edges[i++] = oracle_edge{a, b, v}; });
            std::sort(out_edges, out_edges + ne, [](const oracle_edge& a, const oracle_edge& b) {
                return a.source_1 != b.source_1 ? a.source_1 < b.source_1 : a.source_2 < b.source_2;
            });
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

}  // extern "C"
