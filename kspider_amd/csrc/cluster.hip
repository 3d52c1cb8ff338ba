// kspider_cluster(): the reference's `kSpider cluster` — pykSpider/kSpider2/ks_clustering.py:63-137 — with the
// connected components computed on the GPU.
//
// What the reference does (restated, nothing copied):
//   * nodes: one per row of PREFIX.namesMap (:56-61; first line skipped, "<id> <name>" split on blanks);
//     an edge names its nodes by INDEX = source id - 1 (:99-100), the output names a node by id = index + 1
//     (:135) — so the ids have to be 1..N, as the reference's indexers write them;
//   * edges: every row of PREFIX_kSpider_pairwise.tsv whose column dist_col (min_cont 3, avg_cont 4,
//     max_cont 5; ani: the one column of PREFIX_kSpider_pairwise.ani_col.tsv) parsed as a float and
//     multiplied by 100 is NOT below cutoff * 100 (:101-105; a NaN is never below: kept);
//   * rustworkx.connected_components, one output line per component: the names joined by ',' (:121-137),
//     singletons included, into PREFIX_kSpider_clusters_<cutoff*100>%.tsv (the number printed as Python
//     prints a float: shortest round-trip digits, ".0" on integers).
// Two deliberate differences, both stated in INTEGRATION.md: the reference loses every 10 000 001st kept edge
// (:107-113: the edge that finds the batch full is dropped with the flush — which edge that is depends on its
// hash-map row order); all edges are kept here.  And the order of the lines / of the names inside a line is
// rustworkx's set order there; here components come in order of their smallest node, members ascending.
//
// Device side: min-label hooking + pointer jumping (every parent[] only ever decreases, parent[v] <= v): when
// nothing changes any more every tree is a star whose root is the smallest node of its component.
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/kspider_amd.h"
#include "engine_internal.h"

typedef uint32_t u32;
typedef uint64_t u64;

namespace {

__global__ void k_cc_init(u32* __restrict__ parent, u32 n) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) parent[v] = v;
}
// one pass over the edges: the larger of the two labels is lowered to the smaller one
__global__ void k_cc_hook(const u32* __restrict__ a, const u32* __restrict__ b, u64 m, u32* __restrict__ parent,
                          u32* __restrict__ changed) {
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (u64)gridDim.x * blockDim.x) {
        const u32 pu = parent[a[e]], pv = parent[b[e]];
        if (pu == pv) continue;
        const u32 hi = pu > pv ? pu : pv, lo = pu > pv ? pv : pu;
        if (atomicMin(&parent[hi], lo) > lo) *changed = 1;
    }
}
__global__ void k_cc_jump(u32* __restrict__ parent, u32 n, u32* __restrict__ changed) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const u32 p = parent[v], gp = parent[p];
    if (gp != p) { parent[v] = gp; *changed = 1; }
}

// The same pass straight over the join's edge records (ksp_edge, device memory): an edge counts when its containment
// column — single-precision maths of the pairwise writer, index_io.cpp::format_rows = src/pairwise.cpp:260-264 — is not
// below the cut.  `vcrit` is the smallest float the reference's test (text of the float with 6 significant digits ->
// Python float -> x 100 -> not below cutoff x 100, ks_clustering.py:101-105) lets through: that test is monotone in the
// float, so one compare against the critical value found on the host (ksp::cc_critical) IS that test, digit for digit.
// mode 1: no finite value passes, only NaN rows do (a NaN is never "below": kept, as in the reference).
__device__ inline bool cc_edge_kept(const ksp_edge& x, const u32* __restrict__ cnt, const int col, const float vcrit, const int mode) {
    const float n1 = (float)cnt[x.source_1], n2 = (float)cnt[x.source_2];
    const float c12 = (float)x.shared / n2, c21 = (float)x.shared / n1;
    float v;
    if (col == 3) v = c21 < c12 ? c21 : c12;        // std::min(c12, c21)
    else if (col == 5) v = c12 < c21 ? c21 : c12;   // std::max(c12, c21)
    else v = (float)((double)(c12 + c21) / 2.0);
    if (mode) return v != v;
    return !(v < vcrit);
}
__global__ void k_cc_hook_edges(const ksp_edge* __restrict__ ed, u64 m, const u32* __restrict__ cnt, const int col, const float vcrit,
                                const int mode, u32* __restrict__ parent, u32* __restrict__ changed, unsigned long long* __restrict__ kept) {
    unsigned long long mine = 0;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (u64)gridDim.x * blockDim.x) {
        const ksp_edge x = ed[e];
        if (!cc_edge_kept(x, cnt, col, vcrit, mode)) continue;
        ++mine;
        const u32 pu = parent[x.source_1], pv = parent[x.source_2];
        if (pu == pv) continue;
        const u32 hi = pu > pv ? pu : pv, lo = pu > pv ? pv : pu;
        if (atomicMin(&parent[hi], lo) > lo) *changed = 1;
    }
    if (kept && mine) atomicAdd(kept, mine);
}

#define CL_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t err__ = (call);                                                       \
        if (err__ != hipSuccess) {                                                       \
            ksp::set_error(std::string(#call) + ": " + hipGetErrorString(err__));        \
            rc = KSP_E_HIP;                                                              \
            goto done;                                                                   \
        }                                                                                \
    } while (0)

// text of a double as Python's repr() prints it (the reference builds the output file name with an f-string)
std::string py_float_repr(double v) {
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, r.ptr);
    if (s.find_first_of(".enai") == std::string::npos) s += ".0";
    return s;
}

bool split_tabs(const std::string& line, std::vector<std::string>& out) {
    out.clear();
    size_t b = 0;
    while (true) {
        const size_t e = line.find('\t', b);
        out.emplace_back(line.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return true;
}
std::string strip(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}
bool parse_id(const std::string& t, long long& v) {   // int(text): optional sign, digits, surrounding blanks
    const std::string s = strip(t);
    if (s.empty()) return false;
    char* end = nullptr;
    errno = 0;
    v = std::strtoll(s.c_str(), &end, 10);
    return !errno && end && *end == 0;
}
bool parse_float(const std::string& t, double& v) {   // float(text): decimal, inf, nan
    const std::string s = strip(t);
    if (s.empty()) return false;
    char* end = nullptr;
    v = std::strtod(s.c_str(), &end);
    return end && *end == 0;
}

}  // namespace

extern "C" int ksp_components(int device, uint32_t n_nodes, const uint32_t* h_a, const uint32_t* h_b, uint64_t n_edges,
                              uint32_t* h_label) {
    if ((n_edges && (!h_a || !h_b)) || (n_nodes && !h_label)) { ksp::set_error("ksp_components: NULL argument"); return KSP_E_ARG; }
    for (u64 e = 0; e < n_edges; ++e)
        if (h_a[e] >= n_nodes || h_b[e] >= n_nodes) { ksp::set_error("ksp_components: node index out of range"); return KSP_E_ARG; }
    int rc = KSP_OK;
    u32 *d_a = nullptr, *d_b = nullptr, *d_parent = nullptr, *d_changed = nullptr;
    u32 h_changed = 1;
    int ndev = 0;
    CL_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { ksp::set_error("ksp_components: no such device"); return KSP_E_HIP; }
    CL_HIP(hipSetDevice(device));
    if (n_nodes == 0) return KSP_OK;
    CL_HIP(hipMalloc((void**)&d_parent, (size_t)n_nodes * 4));
    CL_HIP(hipMalloc((void**)&d_changed, 4));
    if (n_edges) {
        CL_HIP(hipMalloc((void**)&d_a, n_edges * 4));
        CL_HIP(hipMalloc((void**)&d_b, n_edges * 4));
        CL_HIP(hipMemcpy(d_a, h_a, n_edges * 4, hipMemcpyHostToDevice));
        CL_HIP(hipMemcpy(d_b, h_b, n_edges * 4, hipMemcpyHostToDevice));
    }
    {
        const unsigned gn = (n_nodes + 255) / 256;
        const unsigned ge = (unsigned)std::min<u64>((n_edges + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(k_cc_init, dim3(gn), dim3(256), 0, nullptr, d_parent, n_nodes);
        // every round at least halves the depth of every tree and merges what an edge connects: O(log n) rounds;
        // the bound only guards against a defect
        for (int round = 0; n_edges && h_changed && round < 10000; ++round) {
            CL_HIP(hipMemsetAsync(d_changed, 0, 4, nullptr));
            hipLaunchKernelGGL(k_cc_hook, dim3(ge), dim3(256), 0, nullptr, d_a, d_b, n_edges, d_parent, d_changed);
            hipLaunchKernelGGL(k_cc_jump, dim3(gn), dim3(256), 0, nullptr, d_parent, n_nodes, d_changed);
            hipLaunchKernelGGL(k_cc_jump, dim3(gn), dim3(256), 0, nullptr, d_parent, n_nodes, d_changed);
            CL_HIP(hipMemcpy(&h_changed, d_changed, 4, hipMemcpyDeviceToHost));
        }
        if (n_edges && h_changed) { ksp::set_error("ksp_components: did not converge"); rc = KSP_E_HIP; goto done; }
        CL_HIP(hipMemcpy(h_label, d_parent, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
    }
done:
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (d_parent) (void)hipFree(d_parent);
    if (d_changed) (void)hipFree(d_changed);
    return rc;
}

namespace ksp {
// The reference keeps a row when float(text of the column) * 100 is not below cutoff * 100 (ks_clustering.py:101-105),
// the text being the float printed with 6 significant digits (src/pairwise.cpp:266-273 = ksp::format_float).  Printing,
// parsing and the multiplication are all monotone, so the rows kept are exactly those whose float is not below ONE
// critical float: found here by bisection over the non-negative floats (their bit patterns are ordered).
// mode 0: keep v when !(v < *vcrit);  mode 1: no finite value and no infinity passes — only NaN rows are kept.
void cc_critical(const double cutoff, float* vcrit, int* mode) {
    const double threshold = cutoff * 100.0;
    auto passes = [&](const uint32_t bits) {
        float v;
        std::memcpy(&v, &bits, 4);
        char buf[64];
        const int n = ksp_format_float(v, buf);
        buf[n] = 0;
        const double d = std::strtod(buf, nullptr) * 100.0;
        return !(d < threshold);
    };
    const uint32_t inf_bits = 0x7F800000u;
    *mode = 0;
    if (!passes(inf_bits)) { *mode = 1; *vcrit = 0; return; }
    uint32_t lo = 0, hi = inf_bits;   // the smallest pattern that passes lies in [lo, hi]; hi passes
    if (passes(0)) hi = 0;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (passes(mid)) hi = mid; else lo = mid + 1;
    }
    std::memcpy(vcrit, &hi, 4);
}

// connected components of the kept edges among `d_edges` (device memory, on the current device); see ksp_components_edges
int cc_edges_on_device(uint32_t n_nodes, const ksp_edge* d_edges, uint64_t n_edges, const uint32_t* d_cnt, int col, double cutoff,
                       uint32_t* h_label, uint64_t* n_kept) {
    int rc = KSP_OK;
    u32 *d_parent = nullptr, *d_changed = nullptr;
    unsigned long long* d_kept = nullptr;
    u32 h_changed = 1;
    float vcrit = 0;
    int mode = 0;
    cc_critical(cutoff, &vcrit, &mode);
    if (n_kept) *n_kept = 0;
    if (n_nodes == 0) return KSP_OK;
    CL_HIP(hipMalloc((void**)&d_parent, (size_t)n_nodes * 4));
    CL_HIP(hipMalloc((void**)&d_changed, 16));
    d_kept = reinterpret_cast<unsigned long long*>(d_changed + 2);
    {
        const unsigned gn = (n_nodes + 255) / 256;
        const unsigned ge = (unsigned)std::min<u64>((n_edges + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(k_cc_init, dim3(gn), dim3(256), 0, nullptr, d_parent, n_nodes);
        for (int round = 0; n_edges && h_changed && round < 10000; ++round) {
            CL_HIP(hipMemsetAsync(d_changed, 0, 16, nullptr));
            hipLaunchKernelGGL(k_cc_hook_edges, dim3(ge), dim3(256), 0, nullptr, d_edges, n_edges, d_cnt, col, vcrit, mode, d_parent, d_changed,
                               round == 0 ? d_kept : nullptr);
            hipLaunchKernelGGL(k_cc_jump, dim3(gn), dim3(256), 0, nullptr, d_parent, n_nodes, d_changed);
            hipLaunchKernelGGL(k_cc_jump, dim3(gn), dim3(256), 0, nullptr, d_parent, n_nodes, d_changed);
            CL_HIP(hipMemcpy(&h_changed, d_changed, 4, hipMemcpyDeviceToHost));
            if (round == 0 && n_kept) { unsigned long long k = 0; CL_HIP(hipMemcpy(&k, d_kept, 8, hipMemcpyDeviceToHost)); *n_kept = k; }
        }
        if (n_edges && h_changed) { set_error("components: did not converge"); rc = KSP_E_HIP; goto done; }
        CL_HIP(hipMemcpy(h_label, d_parent, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
    }
done:
    if (d_parent) (void)hipFree(d_parent);
    if (d_changed) (void)hipFree(d_changed);
    return rc;
}

// one line per component — in order of their smallest node, members ascending — into PREFIX_kSpider_clusters_<cutoff*100>%.tsv
// (ks_clustering.py:121-137, 150-163); label[v] = smallest node of v's component, names by node index
void write_cluster_file(const std::string& prefix, const double threshold, const std::vector<u32>& label,
                        const std::vector<std::string>& name_of) {
    const u64 N = label.size();
    std::vector<u32> count((size_t)N + 1, 0), order((size_t)N);
    for (u64 v = 0; v < N; ++v) ++count[label[v] + 1];
    for (u64 v = 0; v < N; ++v) count[v + 1] += count[v];
    {
        std::vector<u32> cur(count.begin(), count.end() - 1);
        for (u64 v = 0; v < N; ++v) order[cur[label[v]]++] = (u32)v;
    }
    const std::string out = prefix + "_kSpider_clusters_" + py_float_repr(threshold) + "%.tsv";
    const std::string tmp = out + ".partial";
    {
        std::ofstream f(tmp);
        if (!f) throw std::runtime_error("cannot write " + tmp);
        for (u64 r = 0; r < N; ++r) {
            if (count[r + 1] == count[r]) continue;
            for (u32 i = count[r]; i < count[r + 1]; ++i) {
                if (i != count[r]) f << ',';
                f << name_of[order[i]];
            }
            f << '\n';
        }
        f.flush();
        if (!f) { std::remove(tmp.c_str()); throw std::runtime_error("write error on " + tmp); }
    }
    if (std::rename(tmp.c_str(), out.c_str()) != 0) { std::remove(tmp.c_str()); throw std::runtime_error("cannot rename " + tmp); }
}

// PREFIX.namesMap -> name of node index v = id - 1 (ks_clustering.py:56-61); ids must be 1..N
void read_names_map(const std::string& prefix, std::vector<std::string>& name_of) {
    std::unordered_map<long long, std::string> names;   // id -> name (a later row of the same id replaces the earlier)
    std::ifstream f(prefix + ".namesMap");
    if (!f) throw std::runtime_error("cannot open " + prefix + ".namesMap");
    std::string line;
    std::getline(f, line);   // the count line
    while (std::getline(f, line)) {
        const std::string s = strip(line);
        size_t sp = 0;
        while (sp < s.size() && !std::isspace((unsigned char)s[sp])) ++sp;
        size_t nb = sp;
        while (nb < s.size() && std::isspace((unsigned char)s[nb])) ++nb;
        size_t ne = nb;
        while (ne < s.size() && !std::isspace((unsigned char)s[ne])) ++ne;
        long long id;
        if (!parse_id(s.substr(0, sp), id) || ne == nb) throw std::runtime_error("malformed row in " + prefix + ".namesMap");
        names[id] = s.substr(nb, ne - nb);
    }
    const u64 N = names.size();
    if (N >= (1ull << 32)) throw std::runtime_error("more than 2^32 names");
    name_of.assign((size_t)N, std::string());
    for (u64 v = 1; v <= N; ++v) {
        auto it = names.find((long long)v);
        if (it == names.end()) throw std::runtime_error(".namesMap has no id " + std::to_string(v) + " (ids must be 1..N)");
        name_of[(size_t)v - 1] = it->second;
    }
}
}  // namespace ksp

extern "C" int ksp_components_edges(int device, uint32_t n_nodes, const ksp_edge* d_edges, uint64_t n_edges, const uint32_t* d_kmer_counts,
                                    int dist_col, double cutoff, uint32_t* h_label) {
    if ((n_edges && (!d_edges || !d_kmer_counts)) || (n_nodes && !h_label)) { ksp::set_error("ksp_components_edges: NULL argument"); return KSP_E_ARG; }
    if (dist_col < 3 || dist_col > 5) { ksp::set_error("ksp_components_edges: dist_col is 3 (min), 4 (avg) or 5 (max containment)"); return KSP_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { ksp::set_error("ksp_components_edges: no such device"); return KSP_E_HIP; }
    if (hipSetDevice(device) != hipSuccess) { ksp::set_error("ksp_components_edges: hipSetDevice"); return KSP_E_HIP; }
    return ksp::cc_edges_on_device(n_nodes, d_edges, n_edges, d_kmer_counts, dist_col, cutoff, h_label, nullptr);
}

extern "C" int kspider_cluster(const char* index_prefix, const char* dist_type, double cutoff) {
    if (!index_prefix) { ksp::set_error("kspider_cluster: index_prefix is NULL"); return KSP_E_ARG; }
    const std::string prefix = index_prefix, dt = dist_type && *dist_type ? dist_type : "max_cont";
    int col;
    if (dt == "min_cont") col = 3;
    else if (dt == "avg_cont") col = 4;
    else if (dt == "max_cont") col = 5;
    else if (dt == "ani") col = 6;
    else { ksp::set_error("kspider_cluster: unknown distance '" + dt + "' (min_cont, avg_cont, max_cont, ani)"); return KSP_E_ARG; }
    const double threshold = cutoff * 100.0;   // (ks_clustering.py: cutoff = float(cutoff) * 100)
    try {
        std::string line;
        {   // _kSpider_seqToKmersNo.tsv must be there and well-formed (load_seq_to_kmers, :48-53); its values are not used
            std::ifstream f(prefix + "_kSpider_seqToKmersNo.tsv");
            if (!f) throw std::runtime_error("cannot open " + prefix + "_kSpider_seqToKmersNo.tsv");
            std::getline(f, line);
            std::vector<std::string> p;
            while (std::getline(f, line)) {
                split_tabs(strip(line), p);
                long long a, b;
                if (p.size() != 3 || !parse_id(p[1], a) || !parse_id(p[2], b))
                    throw std::runtime_error("malformed row in " + prefix + "_kSpider_seqToKmersNo.tsv");
            }
        }
        std::vector<std::string> name_of;
        ksp::read_names_map(prefix, name_of);
        const u64 N = name_of.size();
        std::vector<u32> ea, eb;
        {
            std::ifstream f(prefix + "_kSpider_pairwise.tsv");
            if (!f) throw std::runtime_error("cannot open " + prefix + "_kSpider_pairwise.tsv");
            std::ifstream ani;
            if (col == 6) {
                ani.open(prefix + "_kSpider_pairwise.ani_col.tsv");
                if (!ani) throw std::runtime_error("ANI was selected, but " + prefix + "_kSpider_pairwise.ani_col.tsv was not found");
                std::getline(ani, line);
            }
            std::getline(f, line);   // header
            std::vector<std::string> p;
            std::string aline;
            while (std::getline(f, line)) {
                split_tabs(strip(line), p);
                long long a, b;
                double d;
                if (p.size() < 2 || !parse_id(p[0], a) || !parse_id(p[1], b)) throw std::runtime_error("malformed row in " + prefix + "_kSpider_pairwise.tsv");
                if (col == 6) {
                    if (!std::getline(ani, aline) || !parse_float(aline, d)) throw std::runtime_error("malformed / short " + prefix + "_kSpider_pairwise.ani_col.tsv");
                } else if ((int)p.size() <= col || !parse_float(p[(size_t)col], d)) {
                    throw std::runtime_error("malformed row in " + prefix + "_kSpider_pairwise.tsv");
                }
                d *= 100.0;
                if (d < threshold) continue;   // (a NaN is not below anything: kept, as in the reference)
                if (a < 1 || b < 1 || (u64)a > N || (u64)b > N)
                    throw std::runtime_error("pairwise row names node " + std::to_string(std::max(a, b)) + " but .namesMap has " + std::to_string(N) + " rows (ids must be 1..N)");
                ea.push_back((u32)(a - 1));
                eb.push_back((u32)(b - 1));
            }
        }
        std::vector<u32> label((size_t)N);
        int device = 0;
        if (const char* dv = std::getenv("KSPIDER_DEVICE")) device = std::atoi(dv);
        const int rc = ksp_components(device, (u32)N, ea.data(), eb.data(), ea.size(), label.data());
        if (rc) return rc;
        ksp::write_cluster_file(prefix, threshold, label, name_of);
        return KSP_OK;
    } catch (const std::bad_alloc&) {
        ksp::set_error("kspider_cluster: out of host memory");
        return KSP_E_LIMIT;
    } catch (const std::exception& e) {
        ksp::set_error(std::string("kspider_cluster: ") + e.what());
        return KSP_E_IO;
    }
}
