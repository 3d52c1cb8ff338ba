"""The drop-in boundary on a real MI355X: index files -> kSpider::pairwise() surfaces ->
TSVs identical (rows sorted by (source_1, source_2), byte for byte) to the CPU oracle's,
and equal to the golden vectors of the reference's own test oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import load_golden_lens, load_golden_pairs, load_sig_set, read_pairwise_tsv
from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_tsvs(oracle, prefix, threads=2, **kw):
    oracle.ref_pairwise(prefix, threads, **kw)
    a = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
    b = open(prefix + "_kSpider_seqToKmersNo.tsv", "rb").read()
    os.remove(prefix + "_kSpider_pairwise.tsv")
    os.remove(prefix + "_kSpider_seqToKmersNo.tsv")
    return a, b


@pytest.mark.parametrize("tag", ["setA", "setB"])
def test_golden_fixture_sets(oracle_lib, tmp_path, tag):
    names, sk = load_sig_set(tag)
    golden = load_golden_pairs(tag)
    # raw sketches straight into the engine
    edges, _ = engine.pairwise_host(sk.keys, sk.offsets)
    got = {(names[e["source_1"]], names[e["source_2"]]): int(e["shared"]) for e in edges}
    assert got == {k: v[0] for k, v in golden.items()}
    # and through the index-file drop-in
    prefix = str(tmp_path / "sigs")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets)
    want_pw, want_sk = _oracle_tsvs(oracle_lib, prefix)
    engine.pairwise(prefix, 2)
    assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want_pw
    assert open(prefix + "_kSpider_seqToKmersNo.tsv", "rb").read() == want_sk
    rows = read_pairwise_tsv(prefix + "_kSpider_pairwise.tsv")
    assert {(names[r[0] - 1], names[r[1] - 1]): r[2] for r in rows} == {k: v[0] for k, v in golden.items()}
    assert len(rows) >= 90


@pytest.mark.parametrize("kwidth,trailer,threads", [(16, True, 1), (16, False, 4), (8, True, 2), (8, False, 3)])
def test_tsv_bytes_equal_oracle_for_every_dump_layout(oracle_lib, tmp_path, kwidth, trailer, threads):
    sk = synth.generate("C2", n_sources=333, mean_size=250, cluster_cap=25, seed=50 + kwidth + trailer)
    gids = (np.arange(sk.n_sources, dtype=np.uint32) * 3 + 11)[::-1].copy()   # arbitrary, unsorted group IDs
    prefix = str(tmp_path / "ix")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets, group_ids=gids, kwidth=kwidth, trailer=trailer)
    want_pw, want_sk = _oracle_tsvs(oracle_lib, prefix, kwidth=kwidth, trailer=trailer)
    engine.pairwise(prefix, threads)
    assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want_pw
    assert open(prefix + "_kSpider_seqToKmersNo.tsv", "rb").read() == want_sk
    assert not os.path.exists(prefix + "_kSpider_pairwise.tsv.partial")


def test_python_module_and_exe_are_drop_ins(oracle_lib, tmp_path):
    sk = synth.generate("C2", n_sources=200, mean_size=150, cluster_cap=16, seed=61)
    prefix = str(tmp_path / "ix")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets)
    want_pw, want_sk = _oracle_tsvs(oracle_lib, prefix)
    # SWIG-compatible module, keyword arguments as in test/kspider_run.py:4
    sys.path.insert(0, os.path.join(ROOT, "kspider_amd", "lib"))
    try:
        import _kSpider_internal as ks
    finally:
        sys.path.pop(0)
    assert ks.pairwise(index_prefix=prefix, user_threads=1) is None
    assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want_pw
    os.remove(prefix + "_kSpider_pairwise.tsv")
    # exe: `pairwise PREFIX THREADS` (pairwise.cpp:3-5); prints the reference's phase lines
    out = subprocess.run([os.path.join(ROOT, "kspider_amd", "lib", "pairwise"), prefix, "2"], check=True,
                         capture_output=True, text=True).stdout
    assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want_pw
    for phase in ("mapping colors to groups:", "parsing index colors:", "kmer counting:",
                  "pairwise hashmap construction:", "writing pairwise matrix to"):
        assert phase in out
    bad = subprocess.run([os.path.join(ROOT, "kspider_amd", "lib", "pairwise"), prefix + "_missing", "2"],
                         capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot open" in bad.stderr


def test_weighted_colour_mode_matches_reference_accumulation(oracle_lib):
    """Engine fed colour runs + weights (the form kSpider::pairwise hands over) vs
    the restated accumulate loop (src/pairwise.cpp:194-237)."""
    sk = synth.generate("C2", n_sources=420, mean_size=500, cluster_cap=48, seed=62, shuffle=False)
    co, src, w = oracle_lib.build_colors(sk.keys, sk.offsets)
    secs, ne, nu, ref = oracle_lib.accumulate_mem(co, src, w, 4)
    # transpose colour -> sources into per-source sorted colour runs with weights
    n = sk.n_sources
    m = np.diff(co.astype(np.int64))
    color_of_entry = np.repeat(np.arange(len(w), dtype=np.uint64) + 1, m)
    order = np.lexsort((color_of_entry, src))
    s_sorted = src[order] - 1
    keys = color_of_entry[order]
    wts = np.repeat(w, m)[order].astype(np.uint32)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(np.bincount(s_sorted, minlength=n))
    edges, st = engine.pairwise_host(keys, offsets, wts)
    assert st["weighted"] == 1
    e = edges.copy()
    e["source_1"] += 1
    e["source_2"] += 1
    assert len(e) == len(ref) and (e == ref).all()
    # and the unweighted raw-hash route gives the same matrix
    raw, _ = engine.pairwise_host(sk.keys, sk.offsets)
    assert (raw == edges).all()


def test_zero_weight_colour_rows(oracle_lib, tmp_path):
    co = np.array([0, 2, 5, 7], dtype=np.uint32)
    src = np.array([1, 2, 2, 3, 4, 1, 2], dtype=np.uint32)
    w = np.array([7, 0, 0], dtype=np.uint32)
    prefix = str(tmp_path / "z")
    oracle_lib.write_index(prefix, co, src, w, np.arange(1, 5, dtype=np.uint32), np.array([10, 20, 30, 40]))
    want_pw, _ = _oracle_tsvs(oracle_lib, prefix, threads=1)
    engine.pairwise(prefix, 1)
    assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want_pw
    assert b"\t0\t0\t0\t0\n" in want_pw


def test_uint32_narrowing_and_missing_kmer_counts(oracle_lib, tmp_path):
    """The reference narrows colour ids and counts to uint32 with insert_or_assign (src/pairwise.cpp:103,
    109,119) and divides by 0 when a group has no k-mer count (operator[] at :257-258): same rows, same text."""
    import struct
    prefix = str(tmp_path / "n")
    # colours 5 and 5 + 2^32 collide after narrowing (the later one wins); group 9 has no k-mer count
    co = np.array([0, 2, 4, 7], dtype=np.uint32)
    src = np.array([1, 2, 3, 4, 1, 3, 9], dtype=np.uint32)
    w = np.array([11, 22, 5], dtype=np.uint32)
    oracle_lib.write_index(prefix, co, src, w, np.array([1, 2, 3, 4], dtype=np.uint32), np.array([100, 200, 300, 400]))
    # rewrite the colour ids by hand: give the first two colours ids 5 and 5 + 2^32
    path = prefix + "_color_to_sources.bin"
    blob = bytearray(open(path, "rb").read())
    n = struct.unpack_from("<Q", blob, 0)[0]
    assert n == 3
    pos, ids_at = 8, []
    for _ in range(n):
        ids_at.append(pos)
        size, cap = struct.unpack_from("<QQ", blob, pos + 8)
        pos += 8 + 16 + (cap + 17) + 4 * cap + 8
    cur = [struct.unpack_from("<Q", blob, p)[0] for p in ids_at]
    order = np.argsort(cur)                      # file order is scattered; pick two colours deterministically
    struct.pack_into("<Q", blob, ids_at[order[0]], 5)
    struct.pack_into("<Q", blob, ids_at[order[1]], 5 + (1 << 32))
    open(path, "wb").write(bytes(blob))
    want_pw, want_sk = _oracle_tsvs(oracle_lib, prefix, threads=1)
    engine.pairwise(prefix, 1)
    got = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
    assert got == want_pw
    assert b"inf" in got                          # group 9: shared / 0
