"""ctypes binding of the C ABI in ``include/kspider_amd.h``.

Everything here goes through ``libkspider_amd.so`` (hand-written HIP for gfx950).
There is no CPU fallback: if the library is missing or no MI355X is visible the
calls raise.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KSPIDER_AMD_LIB") or os.path.join(_HERE, "lib", "libkspider_amd.so")   # (override: A/B timing of two builds)

EDGE_DTYPE = np.dtype([("source_1", "<u4"), ("source_2", "<u4"), ("shared", "<u8")])

KSP_OK, KSP_E_ARG, KSP_E_HIP, KSP_E_IO, KSP_E_OVERFLOW, KSP_E_LIMIT = range(6)

#: every symbol include/kspider_amd.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ksp_last_error", "ksp_device_count", "ksp_engine_create", "ksp_engine_destroy",
    "ksp_engine_build_blocks", "ksp_engine_num_tiles", "ksp_engine_tile_pairs", "ksp_engine_join",
    "ksp_engine_join_launch", "ksp_engine_join_wait", "ksp_engine_join_to_host", "ksp_engine_step_launch",
    "ksp_engine_get_stats", "ksp_device_malloc", "ksp_device_free", "ksp_memcpy_h2d", "ksp_memcpy_d2h",
    "ksp_pairwise_host", "ksp_free", "kspider_pairwise", "ksp_index_info", "ksp_format_float",
    "kspider_pairwise_sigs", "kspider_pairwise_bins",
    "ksp_engine_build_slice", "ksp_engine_slice_sizes", "ksp_engine_slice_export", "ksp_engine_assemble",
    "ksp_engine_edge_bound", "ksp_engine_slice_labels", "ksp_engine_slice_finish", "ksp_engine_balanced_cuts",
    "ksp_engine_build_postings", "ksp_engine_build_postings_slice", "ksp_pairwise_postings_host",
    "ksp_engine_set_profiling", "ksp_engine_phase_times",
    "ksp_pairwise_host_multi", "ksp_pairwise_postings_host_multi",
    "kspider_cluster", "ksp_components", "ksp_components_edges", "kspider_pairwise_and_cluster",
]


class Stats(ctypes.Structure):
    _fields_ = [
        ("n_sources", ctypes.c_uint64), ("n_entries", ctypes.c_uint64), ("n_blocks", ctypes.c_uint64),
        ("n_block_keys", ctypes.c_uint64), ("n_tiles", ctypes.c_uint64), ("last_tiles", ctypes.c_uint64),
        ("last_pairs", ctypes.c_uint64), ("last_stream_bytes", ctypes.c_uint64), ("last_edges", ctypes.c_uint64),
        ("ms_build", ctypes.c_float), ("ms_join", ctypes.c_float), ("weighted", ctypes.c_int),
        ("key_bits", ctypes.c_int), ("n_active_tiles", ctypes.c_uint64), ("last_active_tiles", ctypes.c_uint64),
        ("sort_entries", ctypes.c_uint64), ("ms_sort", ctypes.c_float), ("sort_bits", ctypes.c_int),
        ("partition_kind", ctypes.c_int), ("partition_fallback", ctypes.c_int),
        ("n_match_records", ctypes.c_uint64), ("n_join_workgroups", ctypes.c_uint64),
        ("n_kept_entries", ctypes.c_uint64), ("n_kept_keys", ctypes.c_uint64),
        ("stage1_kind", ctypes.c_int), ("reserved_", ctypes.c_int),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class KspError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"kspider_amd error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libkspider_amd.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP extension is mandatory, there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        L.ksp_last_error.restype = ctypes.c_char_p
        L.ksp_engine_num_tiles.restype = ctypes.c_uint64
        L.ksp_engine_num_tiles.argtypes = [ctypes.c_void_p]
        L.ksp_engine_tile_pairs.restype = ctypes.c_uint64
        L.ksp_engine_edge_bound.restype = ctypes.c_uint64
        L.ksp_engine_balanced_cuts.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
        L.ksp_engine_edge_bound.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.ksp_engine_tile_pairs.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.ksp_engine_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.ksp_engine_destroy.argtypes = [ctypes.c_void_p]
        L.ksp_engine_destroy.restype = None
        L.ksp_engine_build_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        L.ksp_engine_join.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]
        L.ksp_engine_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
        L.ksp_engine_join_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.ksp_engine_join_wait.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ksp_engine_build_slice.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_void_p]
        L.ksp_engine_slice_sizes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.ksp_engine_slice_labels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.ksp_engine_slice_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.ksp_engine_slice_export.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 7
        L.ksp_engine_assemble.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.ksp_device_malloc.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
        L.ksp_device_free.argtypes = [ctypes.c_void_p]
        L.ksp_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        L.ksp_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        L.ksp_pairwise_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(Stats)]
        L.ksp_free.argtypes = [ctypes.c_void_p]
        L.ksp_pairwise_host_multi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                              ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                              ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(Stats)]
        L.ksp_pairwise_postings_host_multi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                       ctypes.c_uint32, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64),
                                                       ctypes.POINTER(Stats)]
        L.ksp_engine_build_postings.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        L.ksp_pairwise_postings_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                 ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                                 ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(Stats)]
        L.ksp_free.restype = None
        L.ksp_engine_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ksp_engine_phase_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p),
                                             ctypes.POINTER(ctypes.c_float), ctypes.c_int]
        L.kspider_cluster.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double]
        L.ksp_components.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                     ctypes.c_void_p]
        L.kspider_pairwise.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.ksp_index_info.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
        L.ksp_format_float.argtypes = [ctypes.c_float, ctypes.c_char_p]
        L.kspider_pairwise_sigs.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.kspider_pairwise_bins.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        _lib = L
    return _lib


def _check(rc):
    if rc != KSP_OK:
        raise KspError(rc, lib().ksp_last_error().decode(errors="replace"))


def device_count() -> int:
    n = ctypes.c_int(0)
    _check(lib().ksp_device_count(ctypes.byref(n)))
    return n.value


def pairwise_host(keys: np.ndarray, offsets: np.ndarray, weights: np.ndarray | None = None, device: int = 0,
                  devices: list | None = None):
    """Host sketches -> (edges sorted by (source_1, source_2), stats).  IDs are dense 0..N-1.
    `devices`: shard the job over several GPUs (ksp_pairwise_host_multi; a device may be named twice)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.uint32)
    n = offsets.size - 1
    out = ctypes.c_void_p()
    ne = ctypes.c_uint64(0)
    st = Stats()
    devs = (ctypes.c_int * len(devices))(*devices) if devices else (ctypes.c_int * 1)(device)
    _check(lib().ksp_pairwise_host_multi(keys.ctypes.data, w.ctypes.data if w is not None else None, offsets.ctypes.data,
                                         n, devs, len(devs), ctypes.byref(out), ctypes.byref(ne), ctypes.byref(st)))
    try:
        buf = (ctypes.c_char * (ne.value * EDGE_DTYPE.itemsize)).from_address(out.value) if ne.value else b""
        edges = np.frombuffer(buf, dtype=EDGE_DTYPE).copy()
    finally:
        lib().ksp_free(out)
    return edges, st.as_dict()


def pairwise(index_prefix: str, user_threads: int = 1) -> None:
    """kSpider::pairwise(index_prefix, user_threads) through the C ABI (reference: kSpider.hpp:11)."""
    _check(lib().kspider_pairwise(os.fsencode(index_prefix), int(user_threads)))


def pairwise_sigs(sigs_dir: str, kSize: int, out_prefix: str | None = None, user_threads: int = 1) -> None:
    """sourmash signatures -> pairwise TSVs (= sourmash_sigs_indexing(sigs_dir, kSize) + pairwise())."""
    _check(lib().kspider_pairwise_sigs(os.fsencode(sigs_dir), int(kSize),
                                       os.fsencode(out_prefix) if out_prefix else None, int(user_threads)))


def pairwise_bins(bins_dir: str, out_prefix: str | None = None, user_threads: int = 1) -> None:
    """phmap flat_hash_set<uint64> sketch dumps (*.bin) -> pairwise TSVs."""
    _check(lib().kspider_pairwise_bins(os.fsencode(bins_dir), os.fsencode(out_prefix) if out_prefix else None,
                                       int(user_threads)))


def pairwise_postings_host(key_off: np.ndarray, sources: np.ndarray, key_weights: np.ndarray | None, n_sources: int,
                           device: int = 0, devices: list | None = None):
    """Inverted index (key k held by sources[key_off[k]:key_off[k+1]], weight key_weights[k] or 1) ->
    (edges sorted by (source_1, source_2), stats).  What the drop-in path feeds the engine."""
    key_off = np.ascontiguousarray(key_off, dtype=np.uint64)
    sources = np.ascontiguousarray(sources, dtype=np.uint32)
    kw = None if key_weights is None else np.ascontiguousarray(key_weights, dtype=np.uint32)
    out = ctypes.c_void_p()
    n = ctypes.c_uint64(0)
    st = Stats()
    devs = (ctypes.c_int * len(devices))(*devices) if devices else (ctypes.c_int * 1)(device)
    _check(lib().ksp_pairwise_postings_host_multi(key_off.ctypes.data, sources.ctypes.data,
                                                  kw.ctypes.data if kw is not None else None, key_off.size - 1, n_sources,
                                                  devs, len(devs), ctypes.byref(out), ctypes.byref(n), ctypes.byref(st)))
    try:
        buf = (ctypes.c_char * (n.value * EDGE_DTYPE.itemsize)).from_address(out.value) if n.value else b""
        edges = np.frombuffer(buf, dtype=EDGE_DTYPE).copy()
    finally:
        lib().ksp_free(out)
    return edges, st.as_dict()


def cluster(index_prefix: str, dist_type: str = "max_cont", cutoff: float = 0.0) -> None:
    """`kSpider cluster -i PREFIX -d DIST -c CUTOFF` (ks_clustering.py:150-163); components on the GPU."""
    _check(lib().kspider_cluster(os.fsencode(index_prefix), dist_type.encode(), float(cutoff)))


def pairwise_and_cluster(index_prefix: str, user_threads: int = 1, dist_type: str = "max_cont", cutoff: float = 0.0) -> None:
    """`kSpider pairwise` + `kSpider cluster` in one device pass: both TSVs as the two calls would write them, the
    components taken from the edges while they are in HBM (the pairwise TSV is never read back)."""
    L = lib()
    L.kspider_pairwise_and_cluster.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_double]
    _check(L.kspider_pairwise_and_cluster(os.fsencode(index_prefix), int(user_threads), dist_type.encode(), float(cutoff)))


def components_edges(n_nodes: int, d_edges_ptr: int, n_edges: int, d_kmer_counts_ptr: int, dist_col: int, cutoff: float,
                     device: int = 0) -> np.ndarray:
    """Components over ksp_edge records in DEVICE memory: an edge counts when its containment column passes the
    reference's cut (include/kspider_amd.h); label[v] = smallest source index of v's component."""
    L = lib()
    L.ksp_components_edges.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_void_p]
    out = np.empty(max(1, n_nodes), dtype=np.uint32)
    _check(L.ksp_components_edges(device, n_nodes, d_edges_ptr or None, n_edges, d_kmer_counts_ptr or None, dist_col, float(cutoff),
                                  out.ctypes.data))
    return out[:n_nodes]


def components(n_nodes: int, a: np.ndarray, b: np.ndarray, device: int = 0) -> np.ndarray:
    """Connected components of an undirected edge list on the GPU: label[v] = smallest node of v's component."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    out = np.empty(n_nodes, dtype=np.uint32)
    _check(lib().ksp_components(device, n_nodes, a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data))
    return out


def index_info(index_prefix: str) -> dict:
    out = (ctypes.c_uint64 * 6)()
    _check(lib().ksp_index_info(os.fsencode(index_prefix), out))
    return dict(colors=out[0], groups=out[1], color_counts=out[2], sources=out[3], kwidth=out[4], trailer=bool(out[5]))


def format_float(v: float) -> str:
    buf = ctypes.create_string_buffer(32)
    n = lib().ksp_format_float(ctypes.c_float(v), buf)
    return buf.raw[:n].decode()


class Engine:
    """Device-resident engine: build_blocks() once per sketch set, join() per tile range."""

    def __init__(self, device: int = 0):
        self.device = device
        self._h = ctypes.c_void_p()
        _check(lib().ksp_engine_create(device, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib().ksp_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_blocks(self, d_keys_ptr: int, h_offsets: np.ndarray, d_weights_ptr: int = 0, key_bits: int = 0,
                     stream: int = 0):
        h_offsets = np.ascontiguousarray(h_offsets, dtype=np.uint64)
        self._off = h_offsets
        _check(lib().ksp_engine_build_blocks(self._h, d_keys_ptr or None, d_weights_ptr or None,
                                             h_offsets.ctypes.data, h_offsets.size - 1, key_bits, stream or None))

    def step_launch(self, d_keys_ptr: int, h_offsets: np.ndarray, part: int, nparts: int, d_edges_ptr: int, capacity: int,
                    stream: int = 0):
        """build_blocks + this rank's tile range + join_launch on it in one call (include/kspider_amd.h).  Returns
        (t0, t1, bound, launched, prev_count); launched False: the bound does not fit `capacity`, call join_launch(t0, t1, ...);
        prev_count: the count of the join that was pending on this engine (None: there was none)."""
        h_offsets = np.ascontiguousarray(h_offsets, dtype=np.uint64)
        self._off = h_offsets
        L = lib()
        L.ksp_engine_step_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                             ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        rng = (ctypes.c_uint64 * 2)()
        bound = ctypes.c_uint64(0)
        prev = ctypes.c_uint64(0)
        prev_rc = ctypes.c_int(0)
        prev_ms = ctypes.c_float(0)
        had = bool(self._join_in_flight) if hasattr(self, "_join_in_flight") else False
        rc = L.ksp_engine_step_launch(self._h, d_keys_ptr or None, None, h_offsets.ctypes.data, h_offsets.size - 1, 0, part, nparts,
                                      d_edges_ptr or None, capacity, rng, ctypes.byref(bound), ctypes.byref(prev), ctypes.byref(prev_rc),
                                      ctypes.byref(prev_ms), stream or None)
        self.prev_ms_join = float(prev_ms.value)
        _check(prev_rc.value)
        self._join_in_flight = rc == KSP_OK
        if rc != KSP_E_OVERFLOW:
            _check(rc)
        return int(rng[0]), int(rng[1]), int(bound.value), rc == KSP_OK, (int(prev.value) if had else None)

    def build_postings(self, h_key_off: np.ndarray, d_sources_ptr: int, d_key_weights_ptr: int, n_sources: int,
                       stream: int = 0):
        """Stage 1 from an inverted index: key k is held by d_sources[key_off[k]:key_off[k+1]] (device uint32)."""
        h_key_off = np.ascontiguousarray(h_key_off, dtype=np.uint64)
        self._off = h_key_off
        _check(lib().ksp_engine_build_postings(self._h, h_key_off.ctypes.data, d_sources_ptr or None,
                                               d_key_weights_ptr or None, h_key_off.size - 1, n_sources, stream or None))

    # ---- key-range sharded stage 1 (multi-GPU) ------------------------------------------------
    def build_slice(self, d_keys_ptr: int, h_offsets: np.ndarray, part: int, nparts: int, d_weights_ptr: int = 0,
                    key_bits: int = 0, stream: int = 0):
        h_offsets = np.ascontiguousarray(h_offsets, dtype=np.uint64)
        self._off = h_offsets
        _check(lib().ksp_engine_build_slice(self._h, d_keys_ptr or None, d_weights_ptr or None,
                                            h_offsets.ctypes.data, h_offsets.size - 1, key_bits, part, nparts,
                                            stream or None))

    def slice_labels(self, d_labels: int, stream: int = 0):
        """Copy the slice's source labels (n_sources uint32) into a device buffer."""
        _check(lib().ksp_engine_slice_labels(self._h, d_labels, stream or None))

    def slice_finish(self, d_labels: int = 0, stream: int = 0):
        """Second half of a slice build, in the source order given by the combined labels."""
        _check(lib().ksp_engine_slice_finish(self._h, d_labels or None, stream or None))

    def slice_sizes(self) -> np.ndarray:
        out = (ctypes.c_uint64 * 4)()
        _check(lib().ksp_engine_slice_sizes(self._h, out))
        return np.array(list(out), dtype=np.uint64)

    def slice_export(self, d_brk: int, d_info: int, d_bw: int, d_blk_raw: int, d_blk_pos: int, d_big: int,
                     stream: int = 0):
        _check(lib().ksp_engine_slice_export(self._h, d_brk or None, d_info or None, d_bw or None, d_blk_raw or None,
                                             d_blk_pos or None, d_big or None, stream or None))

    def assemble(self, h_sizes: np.ndarray, d_brk_all: int, d_info_all: int, d_bw_all: int, lstride: int,
                 d_blk_raw_all: int, d_blk_pos_all: int, d_big_all: int, bigstride: int, stream: int = 0):
        h_sizes = np.ascontiguousarray(h_sizes, dtype=np.uint64)
        nparts = h_sizes.size // 4
        _check(lib().ksp_engine_assemble(self._h, nparts, h_sizes.ctypes.data, d_brk_all or None, d_info_all or None,
                                         d_bw_all or None, lstride, d_blk_raw_all or None, d_blk_pos_all or None,
                                         d_big_all or None, bigstride, stream or None))

    @property
    def num_tiles(self) -> int:
        return lib().ksp_engine_num_tiles(self._h)

    def tile_pairs(self, t0: int, t1: int) -> int:
        return lib().ksp_engine_tile_pairs(self._h, t0, t1)

    def balanced_cuts(self, nparts: int) -> list:
        """Tile ranges of equal estimated work: rank p joins tiles [cuts[p], cuts[p + 1])."""
        out = (ctypes.c_uint64 * (nparts + 1))()
        _check(lib().ksp_engine_balanced_cuts(self._h, nparts, out))
        return [int(x) for x in out]

    def source_order(self, n_sources: int) -> np.ndarray:
        """(diagnostics) engine index (block x 128 + slot) of every source of the last build."""
        L = lib()
        L.ksp_engine_source_order.restype = ctypes.c_int
        L.ksp_engine_source_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        out = np.zeros(max(1, n_sources), dtype=np.uint32)
        _check(L.ksp_engine_source_order(self._h, out.ctypes.data_as(ctypes.c_void_p)))
        return out[:n_sources]

    def edge_bound(self, t0: int, t1: int) -> int:
        """Upper bound on the edges of tiles [t0, t1): source pairs of the tiles that share a key."""
        return lib().ksp_engine_edge_bound(self._h, t0, t1)

    def join_launch(self, t0: int, t1: int, d_edges_ptr: int, capacity: int, stream: int = 0) -> None:
        """Queue the join on `stream` and return; join_wait() collects the count (see include/kspider_amd.h)."""
        _check(lib().ksp_engine_join_launch(self._h, t0, t1, d_edges_ptr or None, capacity, ctypes.c_void_p(stream)))
        self._join_in_flight = True

    def join_wait(self) -> int:
        cnt = ctypes.c_uint64(0)
        self._join_in_flight = False
        _check(lib().ksp_engine_join_wait(self._h, ctypes.byref(cnt)))
        return int(cnt.value)

    def join_to_host(self, t0: int, t1: int, h_edges_ptr: int, capacity: int, stream: int = 0) -> int:
        """Join tiles [t0, t1) piece by piece, every piece copied to (pinned) host memory under the join of the next."""
        cnt = ctypes.c_uint64(0)
        L = lib()
        L.ksp_engine_join_to_host.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                              ctypes.c_void_p, ctypes.c_void_p]
        rc = L.ksp_engine_join_to_host(self._h, t0, t1, h_edges_ptr or None, capacity, ctypes.byref(cnt), stream or None)
        if rc == KSP_E_OVERFLOW:
            err = KspError(rc, L.ksp_last_error().decode())
            err.count = int(cnt.value)
            raise err
        _check(rc)
        return int(cnt.value)

    def join(self, t0: int, t1: int, d_edges_ptr: int, capacity: int, stream: int = 0) -> int:
        cnt = ctypes.c_uint64(0)
        _check(lib().ksp_engine_join(self._h, t0, t1, d_edges_ptr or None, capacity, ctypes.byref(cnt),
                                     stream or None))
        return cnt.value

    def set_profiling(self, on: bool = True):
        """Record one HIP event per phase start of every later build (see phase_times)."""
        _check(lib().ksp_engine_set_profiling(self._h, int(bool(on))))

    def phase_times(self) -> list:
        """[(phase name, ms)] of the last build made with profiling on."""
        names = (ctypes.c_char_p * 24)()
        ms = (ctypes.c_float * 24)()
        n = lib().ksp_engine_phase_times(self._h, names, ms, 24)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def stats(self) -> dict:
        st = Stats()
        _check(lib().ksp_engine_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def ms_join(self) -> float:
        """HIP-event time of the last collected join (one field of the stats, without building the dict)."""
        st = Stats()
        _check(lib().ksp_engine_get_stats(self._h, ctypes.byref(st)))
        return float(st.ms_join)


class DeviceBuffer:
    """hipMalloc'ed buffer through the C ABI (tests use it instead of torch)."""

    def __init__(self, nbytes: int, device: int = 0):
        self.ptr = ctypes.c_void_p()
        self.nbytes = int(nbytes)
        _check(lib().ksp_device_malloc(device, self.nbytes, ctypes.byref(self.ptr)))

    @classmethod
    def from_numpy(cls, a: np.ndarray, device: int = 0):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, device)
        _check(lib().ksp_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes))
        return b

    def to_numpy(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        _check(lib().ksp_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().ksp_device_free(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
