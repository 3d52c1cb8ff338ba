// Direct sketch inputs for the pairwise engine ("next" rows N1 and N3 of SURVEY.md §8f).
//
// N1  kspider_pairwise_sigs(dir, k, ...)  — what `kSpider index --sourmash` followed by
//     `kSpider pairwise` computes (src/sourmash_indexing.cpp:52-350 + src/pairwise.cpp), without
//     the serial string-keyed colour merge in between: sourmash signatures -> sorted hash runs
//     -> MI355X engine.  Mirrored reference behaviour:
//       * group IDs 1.. in glob() order over DIR/* for files ending in .sig or .gz (:85-117),
//         group name = file name minus its last extension (:87-89);
//       * only ".sig" files are read (:152; gzip content is transparent, zstr::ifstream :154),
//         first top-level element, first signature whose "ksize" equals k (:158-167, break :273);
//       * k-mer count = number of "mins" entries (:187);
//       * PREFIX.namesMap: count line, then "<groupID> <groupName>" (:313-319);
//       * PREFIX = basename(DIR), files go to the current directory (:57, :280) unless the
//         caller passes an explicit prefix.
// N3  kspider_pairwise_bins(dir, ...)     — sketches stored as phmap::flat_hash_set<uint64_t>
//     dumps (*.bin, written by sigs_to_bins.cpp:113-136; read by src/bins_indexing.cpp:98-182):
//     group IDs in glob() order over DIR/* for files ending in .bin, k-mer count = set size.
//
// Both write PREFIX.namesMap, PREFIX_kSpider_seqToKmersNo.tsv and PREFIX_kSpider_pairwise.tsv
// (rows sorted by (source_1, source_2), identical maths/text to src/pairwise.cpp:242-275).
#include <glob.h>
#include <zlib.h>

#include <algorithm>
#include <exception>
#include <thread>
#include <mutex>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/kspider_amd.h"
#include "engine_internal.h"
#include "index_io.h"

namespace {

// fn(i) for i in [0, n) on up to `threads` host threads; the first exception is rethrown on the caller
template <class Fn>
void parallel_for(size_t n, int threads, Fn fn) {
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::max(1, threads), n, 64}));
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex mu;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&]() {
            try {
                for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> g(mu);
                if (!err) err = std::current_exception();
            }
        });
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}


std::vector<std::string> glob_dir(const std::string& dir) {
    glob_t g;
    std::memset(&g, 0, sizeof g);
    const std::string pat = dir + "/*";
    int rv = glob(pat.c_str(), GLOB_TILDE, nullptr, &g);   // same call as the reference's glob2()
    std::vector<std::string> out;
    if (rv == 0)
        for (size_t i = 0; i < g.gl_pathc; ++i) out.emplace_back(g.gl_pathv[i]);
    globfree(&g);
    if (rv != 0 && rv != GLOB_NOMATCH) throw std::runtime_error("kspider_amd: glob() failed on " + pat);
    return out;
}

std::string extension_of(const std::string& f) {
    size_t i = f.rfind('.');
    return i == std::string::npos ? "" : f.substr(i + 1);
}
std::string stem_of(const std::string& f) {   // path minus last extension, then basename
    std::string p = f.substr(0, f.find_last_of('.'));
    return p.substr(p.find_last_of("/\\") + 1);
}
std::string strip_slashes(std::string d) {
    while (!d.empty() && d.back() == '/') d.pop_back();
    return d;
}

std::string read_maybe_gz(const std::string& path) {
    gzFile f = gzopen(path.c_str(), "rb");   // transparent for plain files
    if (!f) throw std::runtime_error("kspider_amd: cannot open " + path);
    std::string out;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
    const bool bad = n < 0;
    const int rc = gzclose(f);   // (Z_BUF_ERROR: the compressed stream ends early)
    if (bad || rc != Z_OK) throw std::runtime_error("kspider_amd: read error on " + path + " (truncated or corrupt gzip?)");
    return out;
}

// ---- minimal JSON walker: enough to reach [0]["signatures"][i]{"ksize","mins"} ---------------
struct Json {
    const char* p;
    const char* end;
    const std::string& path;
    int depth = 0;   // nesting of skip(): untrusted input must not be able to exhaust the stack
    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error("kspider_amd: " + path + ": malformed JSON (" + what + ")");
    }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    void need(char c) { if (!eat(c)) fail("unexpected character"); }
    std::string str() {
        need('"');
        std::string s;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                ++p;
                if (p >= end) fail("escape");
                if (*p == 'u') { if (end - p < 5) fail("escape"); p += 4; s += '?'; } else s += *p;
            }
            else s += *p;
            ++p;
        }
        if (p >= end) fail("string");
        ++p;
        return s;
    }
    void skip() {   // any value
        ws();
        if (p >= end) fail("eof");
        if (*p == '"') { str(); return; }
        if (*p == '{' || *p == '[') {
            if (++depth > 512) fail("nesting too deep");
            const bool obj = *p == '{';
            ++p;
            if (!eat(obj ? '}' : ']')) {
                do {
                    if (obj) { str(); need(':'); }
                    skip();
                } while (eat(','));
                need(obj ? '}' : ']');
            }
            --depth;
            return;
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r') ++p;
    }
    // number as text (cpp-json keeps numbers as strings and converts with stoull: json.h:232-235)
    std::string num() {
        ws();
        const char* s = p;
        while (p < end && (std::isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) ++p;
        if (p == s) fail("number");
        return std::string(s, p);
    }
};

// first signature of the first top-level element whose ksize == k; false when there is none
bool parse_sig(const std::string& path, const std::string& text, int k, std::vector<uint64_t>& mins) {
    Json j{text.data(), text.data() + text.size(), path};
    j.need('[');
    j.need('{');
    bool found = false;
    if (!j.eat('}')) {
        do {
            std::string key = j.str();
            j.need(':');
            if (key != "signatures" || found) { j.skip(); continue; }
            j.need('[');
            if (j.eat(']')) continue;
            do {   // one signature object
                j.need('{');
                long long ks = -1;
                std::vector<uint64_t> m;
                bool have_mins = false;
                if (!j.eat('}')) {
                    do {
                        std::string f = j.str();
                        j.need(':');
                        if (f == "ksize") {
                            ks = std::atoll(j.num().c_str());
                        } else if (f == "mins" && !found) {
                            have_mins = true;
                            j.need('[');
                            if (!j.eat(']')) {
                                do {
                                    std::string t = j.num();
                                    errno = 0;
                                    m.push_back(std::strtoull(t.c_str(), nullptr, 10));
                                    if (errno) j.fail("hash out of range");
                                } while (j.eat(','));
                                j.need(']');
                            }
                        } else {
                            j.skip();
                        }
                    } while (j.eat(','));
                    j.need('}');
                }
                if (!found && ks == k) {
                    if (!have_mins) j.fail("signature without mins");
                    mins.swap(m);
                    found = true;
                }
            } while (j.eat(','));
            j.need(']');
        } while (j.eat(','));
        j.need('}');
    }
    return found;
}

struct Source {
    uint32_t id;
    uint32_t kmers;               // what the reference stores in groupID_to_kmerCount
    std::vector<uint64_t> run;    // sorted unique hashes
};

int run_sources(const std::string& prefix, const std::vector<std::pair<uint32_t, std::string>>& names,
                std::vector<Source>& src, int threads) {
    typedef std::chrono::high_resolution_clock Clock;
    {   // .namesMap (src/sourmash_indexing.cpp:313-319)
        std::ofstream f(prefix + ".namesMap");
        if (!f) throw std::runtime_error("kspider_amd: cannot write " + prefix + ".namesMap");
        f << names.size() << "\n";
        for (auto& n : names) f << n.first << " " << n.second << "\n";
    }
    ksp::IndexData ix;
    std::unordered_map<uint32_t, uint32_t> kmer_count;
    for (auto& s : src) {
        ix.kmer_slots.emplace_back(s.id, s.kmers);
        kmer_count[s.id] = s.kmers;
    }
    ksp::write_seq_to_kmers(prefix, ix);

    auto t0 = Clock::now();
    std::vector<uint64_t> offsets(src.size() + 1, 0);
    for (size_t i = 0; i < src.size(); ++i) offsets[i + 1] = offsets[i] + src[i].run.size();
    std::vector<uint64_t> keys(offsets.back());
    for (size_t i = 0; i < src.size(); ++i) std::copy(src[i].run.begin(), src[i].run.end(), keys.begin() + offsets[i]);
    const std::vector<int> devices = ksp::devices_from_env();
    ksp_edge* edges = nullptr;
    uint64_t n_edges = 0;
    int rc = ksp_pairwise_host_multi(keys.data(), nullptr, offsets.data(), (uint32_t)src.size(), devices.data(),
                                     (int)devices.size(), &edges, &n_edges, nullptr);
    if (rc != KSP_OK) return rc;
    std::vector<ksp::EdgeRow> rows;
    rows.reserve(n_edges);
    for (uint64_t i = 0; i < n_edges; ++i)   // sources are in ascending ID order, so rows stay sorted
        rows.push_back(ksp::EdgeRow{src[edges[i].source_1].id, src[edges[i].source_2].id, edges[i].shared});
    ksp_free(edges);
    std::cout << "pairwise hashmap construction: " << std::chrono::duration<double>(Clock::now() - t0).count()
              << " secs" << std::endl;
    std::cout << "writing pairwise matrix to " << prefix << "_kSpider_pairwise.tsv" << std::endl;
    ksp::write_pairwise_tsv(prefix, rows, kmer_count, threads);
    return KSP_OK;
}

std::string default_prefix(const std::string& dir) {
    std::string d = strip_slashes(dir);
    return d.substr(d.find_last_of("/\\") + 1);
}

template <class Fn>
int guarded(Fn fn) {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        ksp::set_error("kspider_amd: out of host memory");
        return KSP_E_LIMIT;
    } catch (const std::exception& e) {
        ksp::set_error(e.what());
        return KSP_E_IO;
    }
}

}  // namespace

extern "C" int kspider_pairwise_sigs(const char* sigs_dir, int kSize, const char* out_prefix, int user_threads) {
    if (!sigs_dir) { ksp::set_error("kspider_pairwise_sigs: sigs_dir is NULL"); return KSP_E_ARG; }
    return guarded([&]() {
        const std::string dir = strip_slashes(sigs_dir);
        const std::string prefix = (out_prefix && *out_prefix) ? out_prefix : default_prefix(dir);
        std::vector<std::pair<uint32_t, std::string>> names;
        std::unordered_map<std::string, uint32_t> id_of;
        std::vector<std::string> sig_files;
        uint32_t next_id = 1;
        for (auto& f : glob_dir(dir)) {
            const std::string ext = extension_of(f);
            if (ext != "sig" && ext != "gz") continue;
            const std::string name = stem_of(f);
            if (!id_of.count(name)) {
                id_of[name] = next_id;
                names.emplace_back(next_id, name);
                ++next_id;
            }
            if (ext == "sig") sig_files.push_back(f);   // pass 2 of the reference reads ".sig" only
        }
        if (names.empty()) throw std::runtime_error("kspider_amd: no .sig/.gz files in " + dir);
        // files are read, parsed and sorted by `user_threads` host threads; the reference's sequential
        // semantics (a later file of the same group replaces the earlier one) are applied in file order
        std::vector<Source> parsed(sig_files.size());
        std::vector<char> has(sig_files.size(), 0);
        parallel_for(sig_files.size(), user_threads, [&](size_t i) {
            const std::string& f = sig_files[i];
            std::vector<uint64_t> mins;
            if (!parse_sig(f, read_maybe_gz(f), kSize, mins)) return;   // no signature with that ksize
            Source& s = parsed[i];
            s.id = id_of.at(stem_of(f));
            s.kmers = (uint32_t)mins.size();
            std::sort(mins.begin(), mins.end());
            mins.erase(std::unique(mins.begin(), mins.end()), mins.end());
            s.run.swap(mins);
            has[i] = 1;
        });
        std::unordered_map<uint32_t, Source> by_id;
        for (size_t i = 0; i < sig_files.size(); ++i)
            if (has[i]) by_id[parsed[i].id] = std::move(parsed[i]);
        if (by_id.empty()) throw std::runtime_error("kspider_amd: no signature with ksize " + std::to_string(kSize));
        std::vector<Source> src;
        for (auto& kv : by_id) src.push_back(std::move(kv.second));
        std::sort(src.begin(), src.end(), [](const Source& a, const Source& b) { return a.id < b.id; });
        return run_sources(prefix, names, src, user_threads < 1 ? 1 : user_threads);
    });
}

extern "C" int kspider_pairwise_bins(const char* bins_dir, const char* out_prefix, int user_threads) {
    if (!bins_dir) { ksp::set_error("kspider_pairwise_bins: bins_dir is NULL"); return KSP_E_ARG; }
    return guarded([&]() {
        const std::string dir = strip_slashes(bins_dir);
        const std::string prefix = (out_prefix && *out_prefix) ? out_prefix : default_prefix(dir);
        std::vector<std::pair<uint32_t, std::string>> names;
        std::vector<Source> src;
        std::vector<std::string> files;
        std::unordered_map<std::string, uint32_t> seen;
        uint32_t next_id = 1;
        for (auto& f : glob_dir(dir)) {
            if (extension_of(f) != "bin") continue;
            const std::string name = stem_of(f);
            if (seen.count(name)) continue;
            seen[name] = next_id;
            names.emplace_back(next_id, name);
            Source s;
            s.id = next_id++;
            src.push_back(std::move(s));
            files.push_back(f);
        }
        parallel_for(files.size(), user_threads, [&](size_t i) {
            Source& s = src[i];
            ksp::load_u64_set(files[i], s.run);
            s.kmers = (uint32_t)s.run.size();
            std::sort(s.run.begin(), s.run.end());
            s.run.erase(std::unique(s.run.begin(), s.run.end()), s.run.end());
        });
        if (src.empty()) throw std::runtime_error("kspider_amd: no .bin files in " + dir);
        return run_sources(prefix, names, src, user_threads < 1 ? 1 : user_threads);
    });
}
