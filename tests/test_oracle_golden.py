"""The oracle (CPU restatement) against the golden vectors produced by the reference's
own test oracle (tests/golden/make_golden.py ran /root/reference/test/generate_golden_files.py).
Runs on CPU; nothing here reads /root/reference."""
import os

import numpy as np
import pytest

from helpers import load_golden_lens, load_golden_pairs, load_sig_set, read_pairwise_tsv


@pytest.mark.parametrize("tag", ["setA", "setB"])
def test_brute_force_equals_golden(oracle_lib, tag):
    names, sk = load_sig_set(tag)
    lens = load_golden_lens(tag)
    assert {n: int(s) for n, s in zip(names, sk.sizes)} == lens          # generate_golden_files.py:23
    golden = load_golden_pairs(tag)
    for brute in (oracle_lib.brute_pairs, oracle_lib.brute_pairs_numpy):
        got = {(names[e["source_1"]], names[e["source_2"]]): int(e["shared"]) for e in brute(sk.keys, sk.offsets)}
        assert got == {k: v[0] for k, v in golden.items()}                # non-zero pairs only, exact


@pytest.mark.parametrize("tag", ["setA", "setB"])
@pytest.mark.parametrize("threads", [1, 3])
def test_ref_pairwise_restatement_equals_golden(oracle_lib, tmp_path, tag, threads):
    names, sk = load_sig_set(tag)
    golden = load_golden_pairs(tag)
    prefix = str(tmp_path / "sigs")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets)
    secs, n_edges, n_updates = oracle_lib.ref_pairwise(prefix, threads)
    rows = read_pairwise_tsv(prefix + "_kSpider_pairwise.tsv")
    assert len(rows) == n_edges == len(golden)
    lens = load_golden_lens(tag)
    for s1, s2, shared, mn, av, mx in rows:
        assert s1 < s2
        g_shared, g_cont = golden[(names[s1 - 1], names[s2 - 1])]
        assert shared == g_shared                                         # validate.py:100-108
        if g_cont is not None:
            # golden containments are float64 rounded to 3 decimals (generate_golden_files.py:76-82);
            # ours are float32 printed with 6 significant digits: agree to 1e-3
            n1, n2 = lens[names[s1 - 1]], lens[names[s2 - 1]]
            assert abs(float(mn) - shared / max(n1, n2)) < 1e-5
            assert abs(float(mx) - shared / min(n1, n2)) < 1e-5
            assert abs(float(mn) - g_cont[0]) <= 1.1e-3 and abs(float(av) - g_cont[1]) <= 1.1e-3
            assert abs(float(mx) - g_cont[2]) <= 1.1e-3
    # seqToKmersNo: one row per group, k-mer counts equal the golden lengths (validate.py:90-94)
    with open(prefix + "_kSpider_seqToKmersNo.tsv") as f:
        assert next(f) == "ID\tseq\tkmers\n"
        got = {}
        for i, line in enumerate(f):
            a, b, c = line.split("\t")
            assert int(a) == i + 1
            got[names[int(b) - 1]] = int(c)
    assert got == lens


def test_accumulate_mem_equals_brute(oracle_lib):
    from kspider_amd import synth
    sk = synth.generate("C2", n_sources=150, mean_size=300, cluster_cap=20, seed=9)
    co, src, w = oracle_lib.build_colors(sk.keys, sk.offsets)
    assert int(w.sum()) == np.unique(sk.keys).size                        # every k-mer has exactly one colour
    for threads in (1, 4):
        secs, ne, nu, edges = oracle_lib.accumulate_mem(co, src, w, threads)
        ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
        e = edges.copy()
        e["source_1"] -= 1
        e["source_2"] -= 1
        assert (e == ref).all()


def test_zero_weight_colour_creates_zero_row(oracle_lib, tmp_path):
    """colorsCount[c] missing/0 still inserts the pair (src/pairwise.cpp:221-225)."""
    co = np.array([0, 2, 5], dtype=np.uint32)
    src = np.array([1, 2, 2, 3, 4], dtype=np.uint32)
    w = np.array([7, 0], dtype=np.uint32)
    prefix = str(tmp_path / "z")
    oracle_lib.write_index(prefix, co, src, w, np.arange(1, 5, dtype=np.uint32), np.array([10, 10, 10, 10]))
    oracle_lib.ref_pairwise(prefix, 1)
    rows = read_pairwise_tsv(prefix + "_kSpider_pairwise.tsv")
    assert [(r[0], r[1], r[2]) for r in rows] == [(1, 2, 7), (2, 3, 0), (2, 4, 0), (3, 4, 0)]
