// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_pairwise.cpp header).
//
// (1) Colour index from sketches — the *semantics* of the reference indexers
//     (/root/reference/src/index.cpp:189-331, src/sourmash_indexing.cpp:190-260):
//     every k-mer ends up with exactly one colour, a colour is a distinct set of
//     source IDs, colorsCount[colour] = #k-mers carrying it, singleton colours
//     reuse the group ID.  The reference's incremental string-keyed merge is
//     not reproduced (colour numbering does not influence pairwise output).
// (2) Writer for the three .bin files in the restated phmap dump layout
//     (oracle.h) + .namesMap (src/index.cpp:372-378).  Slots are scattered over
//     the table so that "slot order" differs from insertion order, as in a real
//     hash table.  Real phmap could iterate these files but not look keys up
//     (we do not know its hash): reader-side test data only.
// (3) Brute-force |A ∩ B| over all pairs (test/generate_golden_files.py:40-49).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "oracle.h"

namespace {

uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct RawTable {
    size_t slot_bytes;
    std::vector<std::vector<unsigned char>> items;
    explicit RawTable(size_t sb) : slot_bytes(sb) {}
    template <class A, class B> void put2(A a, B b) {
        std::vector<unsigned char> v(sizeof(A) + sizeof(B));
        std::memcpy(v.data(), &a, sizeof(A));
        std::memcpy(v.data() + sizeof(A), &b, sizeof(B));
        items.push_back(std::move(v));
    }
    template <class A> void put1(A a) {
        std::vector<unsigned char> v(sizeof(A));
        std::memcpy(v.data(), &a, sizeof(A));
        items.push_back(std::move(v));
    }
    void dump(std::ofstream& f, int kwidth, bool trailer, uint64_t seed) const {
        uint64_t size = items.size();
        uint64_t cap = 0;
        if (size) {
            cap = 1;
            while (cap - cap / 8 < size) cap = cap * 2 + 1;  // 2^k - 1, max load 7/8
        }
        f.write((const char*)&size, 8);
        f.write((const char*)&cap, 8);
        if (!size) return;
        std::vector<signed char> ctrl(cap + kwidth + 1, (signed char)-128);
        std::vector<unsigned char> slots(cap * slot_bytes, 0xCD);
        for (uint64_t i = 0; i < size; ++i) {
            uint64_t h = mix(seed ^ mix(i));
            uint64_t p = (h >> 7) & cap;
            while (p >= cap || ctrl[p] >= 0) p = (p + 1) & cap;  // cap == all-ones mask; slot index < cap
            ctrl[p] = (signed char)(h & 0x7F);
            std::memcpy(slots.data() + p * slot_bytes, items[i].data(), slot_bytes);
        }
        ctrl[cap] = -1;  // sentinel
        for (int i = 0; i < kwidth; ++i)
            ctrl[cap + 1 + i] = (uint64_t)i < cap ? ctrl[i] : (signed char)-128;  // cloned bytes
        f.write((const char*)ctrl.data(), (std::streamsize)ctrl.size());
        f.write((const char*)slots.data(), (std::streamsize)slots.size());
        if (trailer) {
            uint64_t growth_left = cap - cap / 8 - size;
            f.write((const char*)&growth_left, 8);
        }
    }
};

thread_local std::string g_err2;

}  // namespace

extern "C" {

void oracle_free(void* p) { std::free(p); }

int oracle_build_colors(const uint64_t* keys, const uint64_t* offsets, uint32_t n_sources,
                        const uint32_t* group_ids, uint32_t** color_off, uint32_t** sources,
                        uint32_t** color_w, uint32_t* n_colors) {
    try {
        uint64_t total = offsets[n_sources];
        std::vector<std::pair<uint64_t, uint32_t>> ent;
        ent.reserve(total);
        for (uint32_t s = 0; s < n_sources; ++s) {
            uint32_t gid = group_ids ? group_ids[s] : s + 1;
            for (uint64_t e = offsets[s]; e < offsets[s + 1]; ++e) ent.emplace_back(keys[e], gid);
        }
        std::sort(ent.begin(), ent.end());
        ent.erase(std::unique(ent.begin(), ent.end()), ent.end());
        std::map<std::vector<uint32_t>, uint32_t> color_of;  // membership -> colour index
        std::vector<const std::vector<uint32_t>*> members;
        std::vector<uint32_t> weight;
        std::vector<uint32_t> cur;
        size_t i = 0;
        while (i < ent.size()) {
            size_t j = i;
            cur.clear();
            while (j < ent.size() && ent[j].first == ent[i].first) cur.push_back(ent[j++].second);
            auto it = color_of.find(cur);
            if (it == color_of.end()) {
                it = color_of.emplace(cur, (uint32_t)weight.size()).first;
                members.push_back(&it->first);
                weight.push_back(0);
            }
            weight[it->second]++;
            i = j;
        }
        uint32_t C = (uint32_t)weight.size();
        uint64_t tot_m = 0;
        for (auto* m : members) tot_m += m->size();
        uint32_t* off = (uint32_t*)std::malloc(sizeof(uint32_t) * ((size_t)C + 1));
        uint32_t* src = (uint32_t*)std::malloc(sizeof(uint32_t) * std::max<uint64_t>(1, tot_m));
        uint32_t* w = (uint32_t*)std::malloc(sizeof(uint32_t) * std::max<uint32_t>(1, C));
        uint32_t o = 0;
        for (uint32_t c = 0; c < C; ++c) {
            off[c] = o;
            for (uint32_t g : *members[c]) src[o++] = g;
            w[c] = weight[c];
        }
        off[C] = o;
        *color_off = off;
        *sources = src;
        *color_w = w;
        *n_colors = C;
        return 0;
    } catch (const std::exception& e) {
        g_err2 = e.what();
        return 1;
    }
}

int oracle_write_index(const char* index_prefix, const uint32_t* color_off, const uint32_t* sources,
                       const uint32_t* color_w, uint32_t n_colors, const uint32_t* group_ids,
                       const uint32_t* kmer_counts, uint32_t n_sources, int kwidth, int trailer,
                       uint64_t slot_seed) {
    try {
        std::string prefix(index_prefix);
        uint32_t max_gid = 0;
        for (uint32_t s = 0; s < n_sources; ++s) max_gid = std::max(max_gid, group_ids ? group_ids[s] : s + 1);
        // colour ids: singleton colours reuse the group ID, the rest follow max_gid.
        std::vector<uint64_t> color_id(n_colors);
        uint64_t next = (uint64_t)max_gid + 1;
        std::vector<char> has_singleton(max_gid + 1, 0);
        for (uint32_t c = 0; c < n_colors; ++c) {
            uint32_t m = color_off[c + 1] - color_off[c];
            if (m == 1) {
                color_id[c] = sources[color_off[c]];
                has_singleton[sources[color_off[c]]] = 1;
            } else {
                color_id[c] = next++;
            }
        }
        {
            std::ofstream f(prefix + "_groupID_to_kmerCount.bin", std::ios::binary);
            RawTable t(8);
            for (uint32_t s = 0; s < n_sources; ++s)
                t.put2<uint32_t, uint32_t>(group_ids ? group_ids[s] : s + 1, kmer_counts[s]);
            t.dump(f, kwidth, trailer != 0, slot_seed ^ 0x11);
        }
        {
            std::ofstream f(prefix + "_color_to_sources.bin", std::ios::binary);
            uint64_t C = n_colors;
            f.write((const char*)&C, 8);
            // colours in a scattered (hash-like) order
            std::vector<uint32_t> order(n_colors);
            for (uint32_t c = 0; c < n_colors; ++c) order[c] = c;
            std::sort(order.begin(), order.end(),
                      [&](uint32_t a, uint32_t b) { return mix(slot_seed ^ a) < mix(slot_seed ^ b); });
            for (uint32_t c : order) {
                f.write((const char*)&color_id[c], 8);
                RawTable t(4);
                for (uint32_t o = color_off[c]; o < color_off[c + 1]; ++o) t.put1<uint32_t>(sources[o]);
                t.dump(f, kwidth, trailer != 0, slot_seed ^ (0x22 + c));
            }
        }
        {
            std::ofstream f(prefix + "_color_count.bin", std::ios::binary);
            RawTable t(16);
            for (uint32_t c = 0; c < n_colors; ++c) t.put2<uint64_t, uint64_t>(color_id[c], color_w[c]);
            // colours that were allocated and emptied stay in colorsCount with 0
            // (src/index.cpp:159,275,289): sources without a singleton colour.
            for (uint32_t s = 0; s < n_sources; ++s) {
                uint32_t g = group_ids ? group_ids[s] : s + 1;
                if (!has_singleton[g]) t.put2<uint64_t, uint64_t>(g, 0);
            }
            t.dump(f, kwidth, trailer != 0, slot_seed ^ 0x33);
        }
        {
            std::ofstream f(prefix + ".namesMap");
            f << n_sources << "\n";
            for (uint32_t s = 0; s < n_sources; ++s) {
                uint32_t g = group_ids ? group_ids[s] : s + 1;
                f << g << " src" << g << "\n";
            }
        }
        return 0;
    } catch (const std::exception& e) {
        g_err2 = e.what();
        return 1;
    }
}

// One sketch as a phmap::flat_hash_set<uint64_t> dump (the ".bin" files of
// sigs_to_bins.cpp:113-136 / src/bins_indexing.cpp:178-180), restated layout.
int oracle_write_bin_sketch(const char* path, const uint64_t* hashes, uint64_t n, int kwidth, int trailer,
                            uint64_t slot_seed) {
    try {
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("cannot write ") + path);
        RawTable t(8);
        for (uint64_t i = 0; i < n; ++i) t.put1<uint64_t>(hashes[i]);
        t.dump(f, kwidth, trailer != 0, slot_seed);
        return 0;
    } catch (const std::exception& e) {
        g_err2 = e.what();
        return 1;
    }
}

int64_t oracle_brute_pairs(const uint64_t* keys, const uint64_t* offsets, uint32_t n_sources, oracle_edge* out,
                           uint64_t capacity) {
    uint64_t ne = 0;
    for (uint32_t a = 0; a < n_sources; ++a) {
        for (uint32_t b = a + 1; b < n_sources; ++b) {
            uint64_t i = offsets[a], ie = offsets[a + 1], j = offsets[b], je = offsets[b + 1], c = 0;
            while (i < ie && j < je) {
                if (keys[i] < keys[j]) ++i;
                else if (keys[j] < keys[i]) ++j;
                else { ++c; ++i; ++j; }
            }
            if (c) {
                if (ne >= capacity) return -1;
                out[ne++] = oracle_edge{a, b, c};
            }
        }
    }
    return (int64_t)ne;
}

}  // extern "C"
