// Index reader / TSV writers for the kSpider::pairwise() drop-in.
//
// Wire format (restated from parallel-hashmap's phmap_dump.h — the reference snapshot
// ships lib/parallel-hashmap as an empty submodule, so the layout is NOT pinned by any
// reference fixture; see DESIGN.md "wire format: parity unpinned"):
//
//   raw table dump:  u64 size; u64 capacity (2^k - 1);
//                    if size > 0: int8 ctrl[capacity + kWidth + 1]  (full slot <=> ctrl >= 0,
//                                 ctrl[capacity] = -1 sentinel, then kWidth cloned bytes),
//                                 slot slots[capacity], [u64 growth_left in newer releases]
//   PREFIX_groupID_to_kmerCount.bin  one dump, slot = {u32 groupID, u32 kmers}   (src/index.cpp:336-342)
//   PREFIX_color_count.bin           one dump, slot = {u64 colour, u64 count}    (src/index.cpp:362-363)
//   PREFIX_color_to_sources.bin      u64 C, then C x {u64 colour; dump with u32 slots} (src/index.cpp:353-359)
//
// kWidth (16 with SSE2, 8 without) and the optional trailer are detected from the file
// sizes and the cloned control bytes; anything inconsistent is an error, never a guess.
#include "index_io.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>

namespace ksp {
namespace {

struct FileBuf {
    std::string path;
    std::vector<unsigned char> b;
    explicit FileBuf(const std::string& p) : path(p) {
        std::ifstream f(p, std::ios::binary);
        if (!f) throw std::runtime_error("kspider_amd: cannot open " + p);
        f.seekg(0, std::ios::end);
        std::streamoff n = f.tellg();
        if (n < 0) throw std::runtime_error("kspider_amd: cannot size " + p);
        b.resize((size_t)n);
        f.seekg(0);
        if (n && !f.read((char*)b.data(), n)) throw std::runtime_error("kspider_amd: short read on " + p);
    }
};

struct Layout {
    int kwidth;
    bool trailer;
};
const Layout kLayouts[4] = {{16, true}, {16, false}, {8, true}, {8, false}};

// Parses one raw table at `pos`.  Returns false (without throwing) when the bytes do not
// form a valid table under `lay`; on success advances pos and reports full slots in order.
template <class Fn>
bool parse_table(const std::vector<unsigned char>& b, size_t& pos, size_t slot_bytes, Layout lay, int* clone_ok,
                 Fn fn) {
    if (pos + 8 > b.size()) return false;
    uint64_t size, cap;
    std::memcpy(&size, &b[pos], 8);
    if (size == 0 && lay.trailer) {
        // releases that write growth_left stop after `size` for an empty table
        pos += 8;
        return true;
    }
    if (pos + 16 > b.size()) return false;
    std::memcpy(&cap, &b[pos + 8], 8);
    size_t p = pos + 16;
    if (size == 0) {
        // older releases: size and capacity, nothing else
        pos = p;
        return true;
    }
    if (((cap + 1) & cap) != 0 || size > cap || cap > (1ull << 40)) return false;
    const size_t nctrl = (size_t)cap + lay.kwidth + 1;
    if (p + nctrl < p || p + nctrl > b.size()) return false;
    const signed char* ctrl = (const signed char*)&b[p];
    p += nctrl;
    if (slot_bytes && cap > (b.size() - p) / slot_bytes) return false;
    const unsigned char* slots = &b[p];
    p += slot_bytes * (size_t)cap;
    if (lay.trailer) {
        if (p + 8 > b.size()) return false;
        uint64_t growth;
        std::memcpy(&growth, &b[p], 8);
        if (growth > cap) return false;
        p += 8;
    }
    if (ctrl[cap] != -1) return false;  // sentinel
    uint64_t full = 0;
    for (uint64_t i = 0; i < cap; ++i) full += ctrl[i] >= 0;
    if (full != size) return false;
    if (clone_ok) {
        bool ok = true;
        for (int i = 0; i < lay.kwidth && (uint64_t)i < cap; ++i) ok = ok && ctrl[cap + 1 + i] == ctrl[i];
        *clone_ok += ok ? 1 : 0;
    }
    for (uint64_t i = 0; i < cap; ++i)
        if (ctrl[i] >= 0) fn(slots + i * slot_bytes);
    pos = p;
    return true;
}

// Whole-file parse of a single-table file under `lay`; score > 0 iff it parses to EOF.
template <class Fn>
int try_single(const FileBuf& f, size_t slot_bytes, Layout lay, Fn fn) {
    size_t pos = 0;
    int clone = 0;
    if (!parse_table(f.b, pos, slot_bytes, lay, &clone, fn)) return 0;
    if (pos != f.b.size()) return 0;
    return 1 + 2 * clone;   // matching cloned control bytes outweigh the kWidth-16 preference
}

}  // namespace

// Position just behind the raw table at `pos` from its header alone (no look at control bytes or slots).
static bool skip_table(const std::vector<unsigned char>& b, size_t& pos, size_t slot_bytes, Layout lay) {
    if (pos + 8 > b.size()) return false;
    uint64_t size, cap;
    std::memcpy(&size, &b[pos], 8);
    if (size == 0 && lay.trailer) { pos += 8; return true; }
    if (pos + 16 > b.size()) return false;
    std::memcpy(&cap, &b[pos + 8], 8);
    size_t p = pos + 16;
    if (size == 0) { pos = p; return true; }
    if (((cap + 1) & cap) != 0 || size > cap || cap > (1ull << 40)) return false;
    const size_t bytes = (size_t)cap + lay.kwidth + 1 + slot_bytes * (size_t)cap + (lay.trailer ? 8 : 0);
    if (p + bytes < p || p + bytes > b.size()) return false;
    pos = p + bytes;
    return true;
}

void load_index(const std::string& prefix, IndexData& out) {
    FileBuf fk(prefix + "_groupID_to_kmerCount.bin");
    FileBuf fc(prefix + "_color_count.bin");
    FileBuf fs(prefix + "_color_to_sources.bin");

    // The three files come from one phmap build: pick the layout under which all of them parse.  The
    // nested file is walked header by header to its end (table positions), and a prefix of its tables is
    // checked in full (control bytes, cloned bytes); every table is checked in full again when it is read.
    int best = -1, best_score = 0;
    std::vector<size_t> best_pos;
    for (int li = 0; li < 4; ++li) {
        auto nop = [](const unsigned char*) {};
        int s1 = try_single(fk, 8, kLayouts[li], nop);
        int s2 = try_single(fc, 16, kLayouts[li], nop);
        if (!s1 || !s2) continue;
        size_t pos = 0;
        int clone = 0;
        bool ok = fs.b.size() >= 8;
        uint64_t C = 0;
        std::vector<size_t> tpos;
        if (ok) {
            std::memcpy(&C, fs.b.data(), 8);
            pos = 8;
            if (C > fs.b.size()) ok = false;
            if (ok) tpos.reserve((size_t)C + 1);
            for (uint64_t c = 0; ok && c < C; ++c) {
                if (pos + 8 > fs.b.size()) { ok = false; break; }
                tpos.push_back(pos);
                pos += 8;
                if (c < 4096) ok = parse_table(fs.b, pos, 4, kLayouts[li], &clone, nop);
                else ok = skip_table(fs.b, pos, 4, kLayouts[li]);
            }
            ok = ok && pos == fs.b.size();
        }
        if (!ok) continue;
        int score = 1 + s1 + s2 + (clone > 0 ? 2 : 0) + (kLayouts[li].kwidth == 16 ? 1 : 0);   // clone bytes decide ties
        if (score > best_score) { best_score = score; best = li; best_pos.swap(tpos); }
    }
    if (best < 0)
        throw std::runtime_error("kspider_amd: " + prefix +
                                 "_{groupID_to_kmerCount,color_count,color_to_sources}.bin are not a consistent "
                                 "set of phmap dumps (tried kWidth 16/8, with/without growth_left)");
    const Layout lay = kLayouts[best];
    out = IndexData();
    out.kwidth = lay.kwidth;
    out.trailer = lay.trailer;

    try_single(fk, 8, lay, [&](const unsigned char* s) {
        uint32_t k, v;
        std::memcpy(&k, s, 4);
        std::memcpy(&v, s + 4, 4);
        out.kmer_slots.emplace_back(k, v);
    });
    if (out.kmer_slots.empty())   // assert(groupID_to_kmerCount.size()) src/pairwise.cpp:170
        throw std::runtime_error("kspider_amd: " + fk.path + " holds no groups");
    {   // colour -> count, insert_or_assign with both sides narrowed (:119)
        std::vector<std::pair<uint32_t, uint32_t>> kv;
        try_single(fc, 16, lay, [&](const unsigned char* s) {
            uint64_t k, v;
            std::memcpy(&k, s, 8);
            std::memcpy(&v, s + 8, 8);
            kv.emplace_back((uint32_t)k, (uint32_t)v);
        });
        uint32_t mx = 0;
        for (auto& e : kv) mx = std::max(mx, e.first);
        auto& ct = out.colors_count;
        if (!kv.empty() && (uint64_t)mx < 64ull * kv.size() + (1u << 20)) {
            ct.direct.assign((size_t)mx + 1, 0);
            ct.present.assign((size_t)mx + 1, 0);
            for (auto& e : kv) {
                ct.n += ct.present[e.first] ? 0 : 1;
                ct.present[e.first] = 1;
                ct.direct[e.first] = e.second;   // later slot wins, as insert_or_assign does
            }
        } else {
            for (auto& e : kv) ct.sparse[e.first] = e.second;
            ct.n = ct.sparse.size();
        }
    }
    if (out.colors_count.empty())  // assert(tmpMap.size()) src/pairwise.cpp:117
        throw std::runtime_error("kspider_amd: " + fc.path + " holds no colours");
    {
        const size_t C = best_pos.size();
        std::vector<std::pair<uint64_t, std::vector<uint32_t>>> raw(C);
        std::vector<int> bad(1, 0);
        unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        if (C < 4096) nt = 1;
        auto work = [&](size_t c0, size_t c1) {
            for (size_t c = c0; c < c1; ++c) {
                size_t pos = best_pos[c];
                uint64_t k;
                std::memcpy(&k, &fs.b[pos], 8);
                pos += 8;
                uint64_t size = 0;
                std::memcpy(&size, &fs.b[pos], 8);
                auto& v = raw[c].second;
                v.reserve((size_t)std::min<uint64_t>(size, 1u << 24));
                raw[c].first = k;
                if (!parse_table(fs.b, pos, 4, lay, nullptr, [&](const unsigned char* s) {
                        uint32_t x;
                        std::memcpy(&x, s, 4);
                        v.push_back(x);
                    }))
                    bad[0] = 1;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, C * t / nt, C * (t + 1) / nt);
        for (auto& t : th) t.join();
        if (bad[0]) throw std::runtime_error("kspider_amd: " + fs.path + " holds a malformed source table");
        // narrowing of src/pairwise.cpp:103,109 + insert_or_assign: a later colour with the same 32-bit id
        // replaces the sources of the earlier one (first-insertion order is kept).  Ids below 2^32 are the
        // map's own distinct keys, so the bookkeeping is only needed when some id is wider.
        bool wide = false;
        for (auto& r : raw) wide = wide || (r.first >> 32) != 0;
        out.colors.reserve(C);
        if (!wide) {
            for (auto& r : raw) out.colors.emplace_back((uint32_t)r.first, std::move(r.second));
        } else {
            std::unordered_map<uint32_t, size_t> where;
            for (auto& r : raw) {
                const uint32_t k32 = (uint32_t)r.first;
                auto it = where.find(k32);
                if (it == where.end()) {
                    where.emplace(k32, out.colors.size());
                    out.colors.emplace_back(k32, std::move(r.second));
                } else {
                    out.colors[it->second].second = std::move(r.second);
                }
            }
        }
    }
}

void load_u64_set(const std::string& path, std::vector<uint64_t>& out) {
    FileBuf f(path);
    int best = -1, best_score = 0;
    for (int li = 0; li < 4; ++li) {
        int sc = try_single(f, 8, kLayouts[li], [](const unsigned char*) {});
        if (sc) sc += kLayouts[li].kwidth == 16 ? 1 : 0;
        if (sc > best_score) { best_score = sc; best = li; }
    }
    if (best < 0) throw std::runtime_error("kspider_amd: " + path + " is not a phmap flat_hash_set<uint64_t> dump");
    out.clear();
    try_single(f, 8, kLayouts[best], [&](const unsigned char* s) {
        uint64_t v;
        std::memcpy(&v, s, 8);
        out.push_back(v);
    });
}

void write_seq_to_kmers(const std::string& prefix, const IndexData& ix) {
    const std::string path = prefix + "_kSpider_seqToKmersNo.tsv";
    std::string text = "ID\tseq\tkmers\n";
    uint64_t counter = 0;
    char line[96];
    for (auto& it : ix.kmer_slots) {
        int n = std::snprintf(line, sizeof line, "%llu\t%u\t%u\n", (unsigned long long)++counter, it.first, it.second);
        text.append(line, (size_t)n);
    }
    std::ofstream f(path, std::ios::binary);
    if (!f || !f.write(text.data(), (std::streamsize)text.size()))
        throw std::runtime_error("kspider_amd: cannot write " + path);
}

int format_float(char* buf, float v) {
    // libstdc++'s num_put prints a float by converting it to double and calling
    // vsnprintf with "%.*g" and the stream precision (6 by default).
    return std::snprintf(buf, 32, "%.6g", (double)v);
}

namespace {
void format_rows(const std::vector<EdgeRow>& rows, size_t lo, size_t hi,
                 const std::unordered_map<uint32_t, uint32_t>& kmer_count, std::string& out) {
    out.reserve((hi - lo) * 56);
    char buf[160];
    for (size_t i = lo; i < hi; ++i) {
        const EdgeRow& e = rows[i];
        uint32_t n1 = 0, n2 = 0;   // operator[] of the reference yields 0 for a missing group
        auto it1 = kmer_count.find(e.source_1);
        if (it1 != kmer_count.end()) n1 = it1->second;
        auto it2 = kmer_count.find(e.source_2);
        if (it2 != kmer_count.end()) n2 = it2->second;
        // src/pairwise.cpp:260-264, single precision throughout
        const float cont_1_in_2 = (float)e.shared / n2;
        const float cont_2_in_1 = (float)e.shared / n1;
        const float mn = std::min(cont_1_in_2, cont_2_in_1);
        const float av = (cont_1_in_2 + cont_2_in_1) / 2.0;
        const float mx = std::max(cont_1_in_2, cont_2_in_1);
        int n = std::snprintf(buf, sizeof buf, "%u\t%u\t%llu\t", e.source_1, e.source_2, (unsigned long long)e.shared);
        n += format_float(buf + n, mn);
        buf[n++] = '\t';
        n += format_float(buf + n, av);
        buf[n++] = '\t';
        n += format_float(buf + n, mx);
        buf[n++] = '\n';
        out.append(buf, (size_t)n);
    }
}
}  // namespace

void write_pairwise_tsv(const std::string& prefix, const std::vector<EdgeRow>& rows,
                        const std::unordered_map<uint32_t, uint32_t>& kmer_count, int threads) {
    const std::string path = prefix + "_kSpider_pairwise.tsv";
    const std::string tmp = path + ".partial";
    const size_t T = (size_t)std::max(1, std::min(threads, 64));
    const size_t chunk = 1 << 16;
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) throw std::runtime_error("kspider_amd: cannot write " + path);
        f << "source_1\tsource_2\tshared_kmers\tmin_containment\tavg_containment\tmax_containment\n";
        // format T chunks at a time in parallel, write them in order
        for (size_t base = 0; base < rows.size(); base += chunk * T) {
            std::vector<std::string> parts(T);
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; ++t) {
                size_t lo = std::min(rows.size(), base + t * chunk), hi = std::min(rows.size(), lo + chunk);
                if (lo >= hi) break;
                if (T == 1) format_rows(rows, lo, hi, kmer_count, parts[t]);
                else th.emplace_back(format_rows, std::cref(rows), lo, hi, std::cref(kmer_count), std::ref(parts[t]));
            }
            for (auto& x : th) x.join();
            for (auto& s : parts)
                if (!s.empty() && !f.write(s.data(), (std::streamsize)s.size()))
                    throw std::runtime_error("kspider_amd: write failed on " + path);
        }
        f.flush();
        if (!f) throw std::runtime_error("kspider_amd: write failed on " + path);
    }
    // never leave a partial TSV under the final name
    if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("kspider_amd: cannot rename " + tmp);
}

}  // namespace ksp
