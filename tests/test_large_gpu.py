"""A sketch set of more than 2^30 key entries (the limit of one engine build: 32-bit entry positions) through
ksp_pairwise_host: the set is cut into hash-range slices, each built by an engine of its own on the one GPU and
assembled (the multi-GPU machinery with several workers per device).  Size-independent checks as in
tests/test_configs_gpu.py.  (The reference has no such limit — its pair map runs out of RAM first:
src/pairwise.cpp:22-27, ~25 B per non-zero pair.)"""
import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu


def test_more_than_2_30_entries():
    sk = synth.generate("C3", n_sources=220_000)
    n_entries = int(sk.offsets[-1])
    assert n_entries > (1 << 30)
    ev, st = engine.pairwise_host(sk.keys, sk.offsets)
    n = sk.n_sources
    assert st["n_entries"] == n_entries and len(ev) > 10_000_000
    assert (ev["source_1"] < ev["source_2"]).all() and int(ev["source_2"].max()) < n
    key = ev["source_1"].astype(np.int64) * (1 << 32) + ev["source_2"].astype(np.int64)
    assert (np.diff(key) > 0).all()
    sizes = sk.sizes
    assert (ev["shared"] > 0).all() and (ev["shared"] <= np.minimum(sizes[ev["source_1"]], sizes[ev["source_2"]])).all()
    uniq, counts = np.unique(sk.keys, return_counts=True)
    counts = counts.astype(np.int64)
    assert int(ev["shared"].sum(dtype=np.uint64)) == int((counts * (counts - 1) // 2).sum())
    row = np.bincount(ev["source_1"], weights=ev["shared"].astype(np.float64), minlength=n)
    row += np.bincount(ev["source_2"], weights=ev["shared"].astype(np.float64), minlength=n)
    rng = np.random.default_rng(9)
    for a in rng.choice(n, size=200, replace=False):
        assert int(row[a]) == int((counts[np.searchsorted(uniq, sk.run(int(a)))] - 1).sum())
    for i in rng.choice(len(ev), size=200, replace=False):
        a, b, s = int(ev["source_1"][i]), int(ev["source_2"][i]), int(ev["shared"][i])
        assert np.intersect1d(sk.run(a), sk.run(b), assume_unique=True).size == s
