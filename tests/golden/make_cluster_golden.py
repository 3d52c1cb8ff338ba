#!/usr/bin/env python3
"""Golden vectors for the clustering row: the REFERENCE's own Clusters class
(/root/reference/pykSpider/kSpider2/ks_clustering.py, loaded from where it lies, unmodified) is run on the
pairwise TSVs of the two committed signature sets for several (distance, cutoff) pairs, with a pure-Python
stand-in for the three rustworkx calls it makes (tests/golden/_rx_standin).  Stored under
tests/golden/clusters/: the inputs the reference read (.namesMap, seqToKmersNo.tsv, pairwise.tsv — outputs of
the oracle's restated pairwise()) and the cluster files it wrote.  Nothing of the reference is copied.

Run from the repo root (needs /root/reference, click, tqdm):  python tests/golden/make_cluster_golden.py"""
import importlib.util
import os
import shutil
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(HERE, "_rx_standin"))
REF_PKG = "/root/reference/pykSpider/kSpider2"

import oracle  # noqa: E402
from helpers import load_sig_set  # noqa: E402

CASES = [("max_cont", 0.0), ("max_cont", 0.3), ("avg_cont", 0.25), ("min_cont", 0.07), ("min_cont", 0.5), ("max_cont", 1.0)]


def load_reference_clusters_class():
    """ks_clustering.py imports `from kSpider2.click_context import cli`; the package's __init__ would drag in the
    SWIG extension, so the three modules it needs are loaded by path under a bare `kSpider2` package."""
    pkg = types.ModuleType("kSpider2")
    pkg.__path__ = [REF_PKG]
    sys.modules["kSpider2"] = pkg
    for name in ("customLogger", "kSpider_version", "click_context", "ks_clustering"):
        spec = importlib.util.spec_from_file_location("kSpider2." + name, os.path.join(REF_PKG, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["kSpider2." + name] = mod
        spec.loader.exec_module(mod)
    return sys.modules["kSpider2.ks_clustering"].Clusters, sys.modules["kSpider2.customLogger"].Logger


def main():
    Clusters, Logger = load_reference_clusters_class()
    out_root = os.path.join(HERE, "clusters")
    shutil.rmtree(out_root, ignore_errors=True)
    for tag in ("setA", "setB"):
        names, sk = load_sig_set(tag)
        d = os.path.join(out_root, tag)
        os.makedirs(d)
        prefix = os.path.join(d, "sigs")
        oracle.index_from_sketches(prefix, sk.keys, sk.offsets)
        oracle.ref_pairwise(prefix, 2)
        with open(prefix + ".namesMap", "w") as f:      # src/sourmash_indexing.cpp:313-319: count, then "<id> <name>"
            f.write(f"{len(names)}\n")
            for i, nm in enumerate(names):
                f.write(f"{i + 1} {nm}\n")
        for f in os.listdir(d):
            if f.endswith(".bin"):
                os.remove(os.path.join(d, f))
        for dist, cutoff in CASES:
            Clusters.seq_to_kmers = dict()                # (class-level dicts in the reference)
            Clusters.names_map = dict()
            k = Clusters(logger_obj=Logger(True), index_prefix=prefix, cut_off_threshold=float(cutoff) * 100, dist_type=dist)
            k.construct_graph()
            k.cluster_graph()
            src = prefix + f"_kSpider_clusters_{float(cutoff) * 100}%.tsv"
            assert os.path.exists(src), src
            os.rename(src, os.path.join(d, f"ref_{dist}_{cutoff}.clusters"))
            print(tag, dist, cutoff, "->", os.path.basename(src), len(k.connected_components), "components")


if __name__ == "__main__":
    main()
