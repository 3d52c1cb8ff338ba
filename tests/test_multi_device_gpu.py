"""Multi-GPU behind the reference's own entry point (one host thread + one engine per device, C++): on the
one-GPU box the device list names the card several times — two or three engines share it — and the results must
be byte-identical to the single-device run: TSVs of kspider_pairwise() under $KSPIDER_DEVICES, edge sets of the
sketch path (hash-range slices exchanged device to device + assemble) and of the postings path (replicated
deterministic build + tile-range shard)."""
import os
import subprocess

import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0, 0, 0, 0, 0]])
def test_sketch_path_slices_over_devices(oracle_lib, devices):
    sk = synth.generate("C2", n_sources=900, mean_size=700, cluster_cap=40, seed=808)
    one, _ = engine.pairwise_host(sk.keys, sk.offsets)
    many, st = engine.pairwise_host(sk.keys, sk.offsets, devices=devices)
    assert len(one) == len(many) > 1000 and (one == many).all()
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert (many == ref).all()
    assert st["last_edges"] == len(many)


def test_sketch_path_edge_cases_on_two_devices(oracle_lib):
    for runs in ([[], [], []], [[1, 2, 3]], [[5], [5]], [[1, 2, 3], [], [3], [], list(range(1000)), [999]]):
        sk = synth.from_runs(runs)
        one, _ = engine.pairwise_host(sk.keys, sk.offsets)
        two, _ = engine.pairwise_host(sk.keys, sk.offsets, devices=[0, 0])
        assert len(one) == len(two) and (one == two).all()
    # weighted sketches
    rng = np.random.default_rng(4)
    sk = synth.generate("C2", n_sources=300, mean_size=200, cluster_cap=20, seed=809)
    w = rng.integers(1, 50, size=sk.keys.size, dtype=np.uint32)
    uniq, inv = np.unique(sk.keys, return_inverse=True)
    w = rng.integers(1, 50, size=uniq.size, dtype=np.uint32)[inv]      # one weight per key value
    one, _ = engine.pairwise_host(sk.keys, sk.offsets, w)
    two, _ = engine.pairwise_host(sk.keys, sk.offsets, w, devices=[0, 0, 0])
    assert len(one) == len(two) > 0 and (one == two).all()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_postings_path_shards_tiles_over_devices(oracle_lib, devices):
    sk = synth.generate("C2", n_sources=1200, mean_size=500, cluster_cap=50, seed=810)
    co, src, w = oracle_lib.build_colors(sk.keys, sk.offsets)           # colour -> sources (ids = index + 1)
    keep = np.diff(co.astype(np.int64)) >= 2
    off = np.concatenate([[0], np.cumsum(np.diff(co.astype(np.int64))[keep])]).astype(np.uint64)
    sel = np.repeat(keep, np.diff(co.astype(np.int64)))
    sources = (src[sel] - 1).astype(np.uint32)
    one, _ = engine.pairwise_postings_host(off, sources, w[keep], sk.n_sources)
    many, _ = engine.pairwise_postings_host(off, sources, w[keep], sk.n_sources, devices=devices)
    assert len(one) == len(many) > 1000 and (one == many).all()
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert (many == ref).all()


def test_dropin_tsv_bytes_equal_under_kspider_devices(oracle_lib, tmp_path):
    """`pairwise PREFIX T` with KSPIDER_DEVICES=0,0 writes the same bytes as the single-device run and as the
    restated reference (kSpider::pairwise, include/kSpider.hpp:11)."""
    sk = synth.generate("C2", n_sources=500, mean_size=400, cluster_cap=30, seed=811)
    prefix = str(tmp_path / "ix")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets)
    oracle_lib.ref_pairwise(prefix, 2)
    want = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
    os.remove(prefix + "_kSpider_pairwise.tsv")
    exe = os.path.join(ROOT, "kspider_amd", "lib", "pairwise")
    for devs in (None, "0,0", "0,0,0"):
        env = dict(os.environ)
        env.pop("KSPIDER_DEVICES", None)
        if devs:
            env["KSPIDER_DEVICES"] = devs
        subprocess.run([exe, prefix, "2"], check=True, capture_output=True, env=env)
        assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want, devs
        os.remove(prefix + "_kSpider_pairwise.tsv")
    # a device that does not exist fails loudly
    env = dict(os.environ, KSPIDER_DEVICES="0,99")
    p = subprocess.run([exe, prefix, "2"], capture_output=True, env=env)
    assert p.returncode != 0 and not os.path.exists(prefix + "_kSpider_pairwise.tsv")


def test_forced_slices_on_one_device(oracle_lib, monkeypatch):
    """$KSP_SLICES: the sequential-slice path that lifts the 2^30-entry limit of one build, on a small set."""
    sk = synth.generate("C2", n_sources=700, mean_size=600, cluster_cap=40, seed=812)
    one, _ = engine.pairwise_host(sk.keys, sk.offsets)
    monkeypatch.setenv("KSP_SLICES", "3")
    sliced, _ = engine.pairwise_host(sk.keys, sk.offsets)
    assert len(one) == len(sliced) > 1000 and (one == sliced).all()


def _postings(oracle_lib, sk):
    co, src, w = oracle_lib.build_colors(sk.keys, sk.offsets)
    keep = np.diff(co.astype(np.int64)) >= 2
    off = np.concatenate([[0], np.cumsum(np.diff(co.astype(np.int64))[keep])]).astype(np.uint64)
    sel = np.repeat(keep, np.diff(co.astype(np.int64)))
    return off, (src[sel] - 1).astype(np.uint32), w[keep]


@pytest.mark.parametrize("slices", ["2", "3", "7"])
def test_postings_input_in_forced_slices_on_one_device(oracle_lib, monkeypatch, slices):
    """The machinery that lifts the 2^30-membership limit of the drop-in path (src/pairwise.cpp:95-111 has none): the
    colours cut into slices of whole keys, every slice built up to its labels (ksp_engine_build_postings_slice), labels
    MIN-combined, slices assembled.  Forced on a small index via $KSP_SLICES: edges equal the unsliced run and the
    brute-force oracle — weighted colours, and an index with fewer colours than slices (empty slices)."""
    sk = synth.generate("C2", n_sources=1100, mean_size=500, cluster_cap=50, seed=813)
    off, sources, w = _postings(oracle_lib, sk)
    one, _ = engine.pairwise_postings_host(off, sources, w, sk.n_sources)
    monkeypatch.setenv("KSP_SLICES", slices)
    sliced, st = engine.pairwise_postings_host(off, sources, w, sk.n_sources)
    assert len(one) == len(sliced) > 1000 and (one == sliced).all()
    assert (sliced == oracle_lib.brute_pairs(sk.keys, sk.offsets)).all()
    # unweighted (weights NULL) and a tiny index: 3 colours over up to 7 slices
    plain, _ = engine.pairwise_postings_host(off, sources, None, sk.n_sources)
    monkeypatch.delenv("KSP_SLICES")
    plain1, _ = engine.pairwise_postings_host(off, sources, None, sk.n_sources)
    assert (plain == plain1).all()
    monkeypatch.setenv("KSP_SLICES", slices)
    toff = np.array([0, 2, 5, 7], dtype=np.uint64)
    tsrc = np.array([0, 3, 1, 2, 3, 0, 2], dtype=np.uint32)
    tw = np.array([5, 1, 9], dtype=np.uint32)
    tiny, _ = engine.pairwise_postings_host(toff, tsrc, tw, 4)
    want = {(0, 3): 5, (1, 2): 1, (1, 3): 1, (2, 3): 1, (0, 2): 9}
    assert {(int(e["source_1"]), int(e["source_2"])): int(e["shared"]) for e in tiny} == want


def test_dropin_and_cluster_bytes_equal_with_sliced_index(oracle_lib, tmp_path):
    """`pairwise PREFIX T` on an index forced into 3 slices (and into 2 x 2: two devices named, four slices) writes the
    bytes of the unsliced run; kspider_pairwise_and_cluster on the sliced path writes the same cluster file."""
    sk = synth.generate("C2", n_sources=600, mean_size=400, cluster_cap=30, seed=814)
    prefix = str(tmp_path / "ix")
    oracle_lib.index_from_sketches(prefix, sk.keys, sk.offsets)
    with open(prefix + ".namesMap", "w") as f:
        f.write(f"{sk.n_sources}\n")
        for i in range(sk.n_sources):
            f.write(f"{i + 1} g{i + 1}\n")
    oracle_lib.ref_pairwise(prefix, 2)
    want = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
    os.remove(prefix + "_kSpider_pairwise.tsv")
    exe = os.path.join(ROOT, "kspider_amd", "lib", "pairwise")
    for devs, slices in ((None, "3"), ("0,0", "4"), ("0,0,0", None)):
        env = dict(os.environ)
        env.pop("KSPIDER_DEVICES", None)
        env.pop("KSP_SLICES", None)
        if devs:
            env["KSPIDER_DEVICES"] = devs
        if slices:
            env["KSP_SLICES"] = slices
        subprocess.run([exe, prefix, "2"], check=True, capture_output=True, env=env)
        assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want, (devs, slices)
        os.remove(prefix + "_kSpider_pairwise.tsv")
    engine.pairwise_and_cluster(prefix, 2, "max_cont", 0.25)
    from oracle import ref_cluster
    path = ref_cluster.output_path(prefix, 0.25)
    one = open(path, "rb").read()
    os.remove(path)
    os.environ["KSP_SLICES"] = "3"
    try:
        engine.pairwise_and_cluster(prefix, 2, "max_cont", 0.25)
    finally:
        del os.environ["KSP_SLICES"]
    assert open(path, "rb").read() == one and one.count(b"\n") > 1
    assert open(prefix + "_kSpider_pairwise.tsv", "rb").read() == want
