"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (``liboracle.so``: see
``ref_pairwise.cpp`` / ``ref_index.cpp``) plus a numpy brute force that follows
``/root/reference/test/generate_golden_files.py:40-49,76-82``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``kspider_amd``) never does.

Parity pin: semantics pinned by golden vectors from the reference's own test
oracle (``tests/golden/make_golden.py``); phmap wire format: parity unpinned.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Edge(ctypes.Structure):
    _fields_ = [("source_1", ctypes.c_uint32), ("source_2", ctypes.c_uint32), ("shared", ctypes.c_uint64)]


EDGE_DTYPE = np.dtype([("source_1", "<u4"), ("source_2", "<u4"), ("shared", "<u8")])


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ref_pairwise.cpp", "ref_index.cpp", "oracle.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_last_error.restype = ctypes.c_char_p
        L.oracle_brute_pairs.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().oracle_last_error().decode())


def brute_pairs(keys: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """All-pairs |A∩B| by two-pointer merge in C++ (dense ids 0..N-1, non-zero only, sorted)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    cap = max(1, n * (n - 1) // 2)
    out = np.zeros(cap, dtype=EDGE_DTYPE)
    ne = lib().oracle_brute_pairs(_p(keys, ctypes.c_uint64), _p(offsets, ctypes.c_uint64), ctypes.c_uint32(n),
                                  out.ctypes.data_as(ctypes.POINTER(Edge)), ctypes.c_uint64(cap))
    if ne < 0:
        raise RuntimeError("brute_pairs: buffer too small")
    return out[:ne].copy()


def brute_pairs_numpy(keys: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """Pure numpy/python restatement of generate_golden_files.py:40-49 (small inputs only)."""
    n = offsets.size - 1
    sets = [set(keys[int(offsets[s]):int(offsets[s + 1])].tolist()) for s in range(n)]
    rows = []
    for a in range(n):
        for b in range(a + 1, n):
            common = len(sets[a].intersection(sets[b]))
            if common:
                rows.append((a, b, common))
    return np.array(rows, dtype=EDGE_DTYPE)


def build_colors(keys: np.ndarray, offsets: np.ndarray, group_ids: np.ndarray | None = None):
    """Colour index (CSR colour->sources as group IDs, colour weights) from sketches."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    gid = None if group_ids is None else np.ascontiguousarray(group_ids, dtype=np.uint32)
    off_p = ctypes.POINTER(ctypes.c_uint32)()
    src_p = ctypes.POINTER(ctypes.c_uint32)()
    w_p = ctypes.POINTER(ctypes.c_uint32)()
    nc = ctypes.c_uint32(0)
    _check(lib().oracle_build_colors(_p(keys, ctypes.c_uint64), _p(offsets, ctypes.c_uint64), ctypes.c_uint32(n),
                                     _p(gid, ctypes.c_uint32) if gid is not None else None,
                                     ctypes.byref(off_p), ctypes.byref(src_p), ctypes.byref(w_p), ctypes.byref(nc)))
    C = nc.value
    color_off = np.ctypeslib.as_array(off_p, shape=(C + 1,)).copy()
    sources = np.ctypeslib.as_array(src_p, shape=(max(1, int(color_off[-1])),)).copy()[: int(color_off[-1])]
    color_w = np.ctypeslib.as_array(w_p, shape=(max(1, C),)).copy()[:C]
    for ptr in (off_p, src_p, w_p):
        lib().oracle_free(ptr)
    return color_off, sources, color_w


def write_index(prefix: str, color_off, sources, color_w, group_ids, kmer_counts, kwidth: int = 16,
                trailer: bool = True, slot_seed: int = 1234):
    """Write PREFIX_{color_to_sources,color_count,groupID_to_kmerCount}.bin + .namesMap."""
    color_off = np.ascontiguousarray(color_off, dtype=np.uint32)
    sources = np.ascontiguousarray(sources, dtype=np.uint32)
    color_w = np.ascontiguousarray(color_w, dtype=np.uint32)
    kc = np.ascontiguousarray(kmer_counts, dtype=np.uint32)
    gid = None if group_ids is None else np.ascontiguousarray(group_ids, dtype=np.uint32)
    _check(lib().oracle_write_index(prefix.encode(), _p(color_off, ctypes.c_uint32), _p(sources, ctypes.c_uint32),
                                    _p(color_w, ctypes.c_uint32), ctypes.c_uint32(color_w.size),
                                    _p(gid, ctypes.c_uint32) if gid is not None else None, _p(kc, ctypes.c_uint32),
                                    ctypes.c_uint32(kc.size), ctypes.c_int(kwidth), ctypes.c_int(int(trailer)),
                                    ctypes.c_uint64(slot_seed)))


def write_bin_sketch(path: str, hashes: np.ndarray, kwidth: int = 16, trailer: bool = True, slot_seed: int = 99):
    """One sketch as a phmap flat_hash_set<uint64_t> dump (.bin)."""
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    rc = lib().oracle_write_bin_sketch(path.encode(), _p(h, ctypes.c_uint64), ctypes.c_uint64(h.size),
                                       ctypes.c_int(kwidth), ctypes.c_int(int(trailer)), ctypes.c_uint64(slot_seed))
    if rc != 0:
        raise RuntimeError("write_bin_sketch failed")


def index_from_sketches(prefix: str, keys, offsets, group_ids=None, **kw):
    """Sketches -> colour index files (what `kSpider index` hands to `kSpider pairwise`)."""
    color_off, sources, color_w = build_colors(keys, offsets, group_ids)
    write_index(prefix, color_off, sources, color_w, group_ids, np.diff(np.asarray(offsets)).astype(np.uint32), **kw)
    return color_off, sources, color_w


def ref_pairwise(prefix: str, user_threads: int = 1, kwidth: int = 16, trailer: bool = True, sorted_rows: bool = True):
    """Restatement of kSpider::pairwise on index files; returns (secs_accumulate, n_edges, n_updates)."""
    secs = ctypes.c_double(0)
    ne = ctypes.c_uint64(0)
    nu = ctypes.c_uint64(0)
    _check(lib().oracle_ref_pairwise(prefix.encode(), ctypes.c_int(user_threads), ctypes.c_int(kwidth),
                                     ctypes.c_int(int(trailer)), ctypes.c_int(int(sorted_rows)), ctypes.byref(secs),
                                     ctypes.byref(ne), ctypes.byref(nu)))
    return secs.value, ne.value, nu.value


def accumulate_mem(color_off, sources, color_w, user_threads: int = 1, want_edges: bool = True):
    """Timed accumulation (src/pairwise.cpp:200-239 equivalent) on an in-memory colour CSR."""
    color_off = np.ascontiguousarray(color_off, dtype=np.uint32)
    sources = np.ascontiguousarray(sources, dtype=np.uint32)
    color_w = np.ascontiguousarray(color_w, dtype=np.uint32)
    secs = ctypes.c_double(0)
    ne = ctypes.c_uint64(0)
    nu = ctypes.c_uint64(0)
    if want_edges:
        m = np.diff(color_off.astype(np.int64))
        top = int(sources.max()) + 1 if sources.size else 1      # (no more pairs than C(#ids, 2) either)
        cap = int(max(1, min((m * (m - 1) // 2).sum(), top * (top - 1) // 2)))
        out = np.zeros(cap, dtype=EDGE_DTYPE)
        outp = out.ctypes.data_as(ctypes.POINTER(Edge))
    else:
        cap, out, outp = 0, None, None
    _check(lib().oracle_accumulate_mem(_p(color_off, ctypes.c_uint32), _p(sources, ctypes.c_uint32),
                                       _p(color_w, ctypes.c_uint32), ctypes.c_uint32(color_w.size),
                                       ctypes.c_int(user_threads), ctypes.byref(secs), ctypes.byref(ne),
                                       ctypes.byref(nu), outp, ctypes.c_uint64(cap)))
    return secs.value, ne.value, nu.value, (out[: ne.value].copy() if out is not None else None)


def containment_rows(edges: np.ndarray, kmer_count: dict) -> list:
    """Float maths of src/pairwise.cpp:260-264 in numpy float32 (for text comparisons)."""
    rows = []
    for e in edges:
        s = np.float32(np.uint64(e["shared"]))
        c12 = s / np.float32(kmer_count[int(e["source_2"])])
        c21 = s / np.float32(kmer_count[int(e["source_1"])])
        rows.append((int(e["source_1"]), int(e["source_2"]), int(e["shared"]), min(c12, c21),
                     np.float32((np.float64(c12 + c21)) / 2.0), max(c12, c21)))
    return rows
