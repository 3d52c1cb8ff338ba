"""Shared helpers for the test-suite (fixtures on disk, TSV parsing)."""
import json
import os

import numpy as np

from kspider_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def load_sig_set(tag):
    """tests/golden/<tag>/sigs/*.sig -> (names sorted, SketchSet); group ID = index + 1
    (glob order, as src/sourmash_indexing.cpp:85-117 assigns them)."""
    d = os.path.join(GOLDEN, tag, "sigs")
    names = sorted(f[:-4] for f in os.listdir(d) if f.endswith(".sig"))
    runs = []
    for nm in names:
        with open(os.path.join(d, nm + ".sig")) as f:
            doc = json.load(f)
        runs.append(doc[0]["signatures"][0]["mins"])   # test/generate_golden_files.py:10
    return names, synth.from_runs(runs, tag)


def load_golden_pairs(tag):
    """golden_pairwise.tsv -> {(sig1, sig2): (shared, [min, avg, max] or None)}"""
    out = {}
    with open(os.path.join(GOLDEN, tag, "golden_pairwise.tsv")) as f:
        header = next(f).rstrip("\n").split("\t")
        for line in f:
            p = line.rstrip("\n").split("\t")
            cont = [float(x) for x in p[3:6]] if len(header) > 3 else None
            out[(p[0], p[1])] = (int(p[2]), cont)
    return out


def load_golden_lens(tag):
    with open(os.path.join(GOLDEN, tag, "golden_sig_to_len.tsv")) as f:
        return {a: int(b) for a, b in (l.split() for l in f)}


def read_pairwise_tsv(path):
    rows = []
    with open(path) as f:
        header = next(f)
        assert header == "source_1\tsource_2\tshared_kmers\tmin_containment\tavg_containment\tmax_containment\n"
        for line in f:
            p = line.rstrip("\n").split("\t")
            rows.append((int(p[0]), int(p[1]), int(p[2]), p[3], p[4], p[5]))
    return rows


def edges_as_tuples(edges):
    return [(int(e["source_1"]), int(e["source_2"]), int(e["shared"])) for e in edges]
