"""The hand-written two-level partition of stage 1 (kspider_amd/csrc/partition_kernels.hip.h) against the
brute-force oracle and against the rocPRIM partition it replaces: page boundaries, chunks that hold many
sources (and empty ones), one source spanning many chunks, key ranges that are not powers of two, full 64-bit
keys, fewer key values than buckets, and key distributions so far from uniform that the page tables
overflow and the build must fall back — same edges every way."""
import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu


def _edges(sk, monkeypatch, **env):
    for k in ("KSP_PARTITION", "KSP_PART_MIN", "KSP_DEBUG_BUCKET_MEAN", "KSP_SEG"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return engine.pairwise_host(sk.keys, sk.offsets)


def _both(sk, monkeypatch, oracle=None, expect_hand=True, expect_seg=None):
    """Paged partition (KSP_SEG=0), segment partition forced (KSP_SEG=1: level 1 read off the sorted runs; a tile or
    bucket that does not fit sends the build back to the paged one, partition_fallback 4 / 5) and the library's."""
    hand, st = _edges(sk, monkeypatch, KSP_PART_MIN="1", KSP_SEG="0")
    seg, st_seg = _edges(sk, monkeypatch, KSP_PART_MIN="1", KSP_SEG="1")
    lib, st_lib = _edges(sk, monkeypatch, KSP_PARTITION="rocprim")
    assert st_lib["partition_kind"] in (0, 1)
    assert st["partition_fallback"] in (0, 1), st      # (2 / 3 would be defects of the partition itself)
    if expect_hand:
        assert st["partition_kind"] == 2 and st["partition_fallback"] == 0, st
    assert st_seg["partition_fallback"] in (0, 1, 4, 5), st_seg
    if expect_seg is not None:
        assert (st_seg["partition_kind"] == 3 and st_seg["partition_fallback"] == 0) == expect_seg, st_seg
    assert len(seg) == len(lib) and (seg == lib).all()
    assert len(hand) == len(lib) and (hand == lib).all()
    if oracle is not None:
        ref = oracle.brute_pairs(sk.keys, sk.offsets)
        assert len(hand) == len(ref) and (hand == ref).all()
    return hand, st


def test_pages_chunks_and_ragged_sources(oracle_lib, monkeypatch):
    """~1.6 M entries: lists of several pages, 2 048-entry chunks holding dozens of tiny sources, empty
    sources in between, and one 300 000-entry source that spans ~150 chunks."""
    rng = np.random.default_rng(7)
    pool = rng.integers(0, (1 << 64) // 1000, size=400_000, dtype=np.uint64)
    runs = []
    for s in range(1200):
        if s % 17 == 0:
            runs.append(np.zeros(0, dtype=np.uint64))
        elif s == 601:
            runs.append(np.unique(np.concatenate([pool[:200_000], rng.integers(0, (1 << 64) // 1000, size=100_000, dtype=np.uint64)])))
        elif s % 3 == 0:
            runs.append(np.unique(pool[rng.integers(0, pool.size, size=int(rng.integers(1, 60)))]))
        else:
            fam = pool[(s % 7) * 50_000:(s % 7 + 1) * 50_000]
            runs.append(np.unique(np.concatenate([fam[rng.random(fam.size) < 0.03],
                                                  rng.integers(0, (1 << 64) // 1000, size=300, dtype=np.uint64)])))
    sk = synth.from_runs(runs)
    assert int(sk.offsets[-1]) > 1_500_000
    _both(sk, monkeypatch, oracle_lib)


@pytest.mark.parametrize("n_sources,size", [(3, 5), (40, 100), (513, 700), (2000, 2100)])
def test_small_and_medium_sets(oracle_lib, monkeypatch, n_sources, size):
    sk = synth.generate("C2", n_sources=n_sources, mean_size=size, cluster_cap=max(2, n_sources // 20), seed=900 + n_sources)
    _both(sk, monkeypatch, oracle_lib, expect_seg=True)


def test_full_width_keys_and_the_largest_key(oracle_lib, monkeypatch):
    rng = np.random.default_rng(8)
    top = np.uint64(0xFFFFFFFFFFFFFFFF)
    runs = []
    for s in range(300):
        wide = rng.integers(0, 1 << 64, size=500, dtype=np.uint64)
        extra = [top, np.uint64(0)] if s % 2 == 0 else [top - np.uint64(1), np.uint64(1) << np.uint64(63)]
        runs.append(np.unique(np.concatenate([wide, np.array(extra, dtype=np.uint64)])))
    _both(synth.from_runs(runs), monkeypatch, oracle_lib)


def test_fewer_key_values_than_buckets(oracle_lib, monkeypatch):
    """Keys 0..99 held by many sources each (a colour-id-like key space): every key is its own bucket."""
    rng = np.random.default_rng(9)
    runs = [np.unique(rng.integers(0, 100, size=int(rng.integers(5, 80)), dtype=np.uint64)) for _ in range(900)]
    _both(synth.from_runs(runs), monkeypatch, oracle_lib, expect_hand=False)   # (may end on the sort path: huge buckets)


def test_skewed_keys_overflow_the_page_tables_and_fall_back(oracle_lib, monkeypatch):
    """Nearly all keys inside 1/100 000 of the key range: one level-1 bucket would need hundreds of pages, the
    partition raises its overflow word and the same build is repeated with the library partition."""
    rng = np.random.default_rng(10)
    runs = []
    for s in range(700):
        low = rng.integers(0, 1 << 40, size=2500, dtype=np.uint64)
        high = rng.integers(0, 1 << 57, size=20, dtype=np.uint64)
        runs.append(np.unique(np.concatenate([low, high, np.arange(s % 5, 4000, 5, dtype=np.uint64)])))
    sk = synth.from_runs(runs)
    hand, st = _edges(sk, monkeypatch, KSP_PART_MIN="1")                             # (segment partition first: a tile overflows)
    assert st["partition_kind"] == 1 and st["partition_fallback"] == 1, st          # fell back: page tables full
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert len(hand) == len(ref) and (hand == ref).all()


@pytest.mark.parametrize("seg", ["0", "1"])
def test_rebuilds_on_one_engine_give_the_same_edges(monkeypatch, seg):
    """The page pools, cursors and page tables (the bucket cursors of the segment partition) are reset per build: ten
    builds on one engine, same edge set."""
    monkeypatch.setenv("KSP_SEG", seg)
    sk = synth.generate("C2", n_sources=1500, mean_size=1500, cluster_cap=60, seed=321)
    dk = engine.DeviceBuffer.from_numpy(sk.keys)
    e = engine.Engine(0)
    cap = 1 << 22
    de = engine.DeviceBuffer(cap * 16)
    first = None
    for _ in range(10):
        e.build_blocks(dk.ptr.value, sk.offsets)
        assert e.stats()["partition_kind"] == (3 if seg == "1" else 2)
        cnt = e.join(0, e.num_tiles, de.ptr.value, cap)
        ev = np.sort(de.to_numpy(engine.EDGE_DTYPE, cnt), order=["source_1", "source_2"])
        if first is None:
            first = ev
        assert len(ev) == len(first) and (ev == first).all()


def test_three_levels_when_there_are_more_than_65536_buckets(oracle_lib, monkeypatch):
    """Sets above 1.3e8 entries (C3, C4) get a middle level between the level-1 lists and the final scatter; tiny
    buckets force it on a set the brute-force oracle can check."""
    rng = np.random.default_rng(17)
    pool = rng.integers(0, (1 << 64) // 1000, size=300_000, dtype=np.uint64)
    runs = []
    for s in range(900):
        fam = pool[(s % 6) * 50_000:(s % 6 + 1) * 50_000]
        runs.append(np.unique(np.concatenate([fam[rng.random(fam.size) < 0.025],
                                              rng.integers(0, (1 << 64) // 1000, size=int(rng.integers(1, 900)), dtype=np.uint64)])))
    sk = synth.from_runs(runs)
    assert int(sk.offsets[-1]) > 1_200_000                      # / 16 per bucket: > 65 536 buckets
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    for mean in ("16", "9"):
        for seg, kind in (("0", 2), ("1", 3)):   # level 1 copied into pages (k_part1) / read off the sorted runs (k_seg_mid)
            hand, st = _edges(sk, monkeypatch, KSP_PART_MIN="1", KSP_DEBUG_BUCKET_MEAN=mean, KSP_SEG=seg)
            assert st["partition_kind"] == kind and st["partition_fallback"] == 0, st
            assert st["sort_bits"] > 16                                 # more than 2^16 buckets
            assert len(hand) == len(ref) and (hand == ref).all()


def test_segment_partition_shapes(oracle_lib, monkeypatch):
    """The segment partition on what it is for and around its edges: long sorted runs (chosen by the segment-length
    rule, no switch), groups cut at 512 sources (tiny runs), empty runs between long ones, a run that ends exactly on a
    window of the boundary scan, one bucket range that no key of a source falls into, and a hot key whose bucket
    outgrows its fixed places (falls back, same edges)."""
    rng = np.random.default_rng(23)
    hmax = (1 << 64) // 1000
    # (a) sourmash-like: 600 sources x ~4 000 hashes, clusters of 30
    sk = synth.generate("C2", n_sources=600, mean_size=4000, cluster_cap=30, seed=77)
    edges, st = _edges(sk, monkeypatch)
    assert st["partition_kind"] == 3 and st["partition_fallback"] == 0, st
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert len(edges) == len(ref) and (edges == ref).all()
    # (b) thousands of tiny runs (groups end at 512 sources), empty runs, runs of exactly 256 / 512 entries, and runs
    # whose keys all sit in the upper half of the range (boundaries crossed in one step)
    pool = rng.integers(0, hmax, size=60_000, dtype=np.uint64)
    runs = []
    for s in range(2600):
        if s % 11 == 0:
            runs.append(np.zeros(0, dtype=np.uint64))
        elif s % 13 == 0:
            r = np.unique(pool[rng.integers(0, pool.size, size=700)])[:256 * (1 + s % 2)]
            runs.append(r)
        elif s % 7 == 0:
            runs.append(np.unique(pool[pool > hmax // 2][rng.integers(0, 20_000, size=300)]))
        else:
            runs.append(np.unique(pool[rng.integers(0, pool.size, size=int(rng.integers(1, 40)))]))
    _both(synth.from_runs(runs), monkeypatch, oracle_lib)
    # (c) one key held by every source: its bucket needs 3 000 places, the others ~150
    hot = np.uint64(hmax // 3)
    runs = [np.unique(np.concatenate([rng.integers(0, hmax, size=600, dtype=np.uint64), pool[rng.integers(0, pool.size, size=40)],
                                      np.array([hot], dtype=np.uint64)])) for _ in range(3000)]
    sk = synth.from_runs(runs)
    edges, st = _edges(sk, monkeypatch, KSP_PART_MIN="1", KSP_SEG="1", KSP_DEBUG_BUCKET_MEAN="150")
    assert st["partition_kind"] == 2 and st["partition_fallback"] == 5, st        # a bucket overflowed: paged partition
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert len(edges) == len(ref) and (edges == ref).all()


def test_stable_split_of_the_group_records_up_to_1024_blocks(oracle_lib, monkeypatch):
    """24 000 sources of a dozen hashes: 283 blocks with the spare ones — above the 256 the hand-written split of the
    group records takes by default.  Library sort (default there), the split's 1 024-block tables (KSP_MS=1024: 16-bit
    counters, ten ballots, positional masks for the join) and KSP_MS=0 give the same edges as the restated reference."""
    sk = synth.generate("C5", n_sources=24000, seed=777)
    co, src, w = oracle_lib.build_colors(sk.keys, sk.offsets)
    _, n_edges, _, ref = oracle_lib.accumulate_mem(co, src, w, 8)
    ref = np.sort(ref, order=["source_1", "source_2"])
    for ms in (None, "1024", "0"):
        monkeypatch.delenv("KSP_MS", raising=False)
        if ms is not None:
            monkeypatch.setenv("KSP_MS", ms)
        edges, st = engine.pairwise_host(sk.keys, sk.offsets)
        assert st["n_blocks"] > 256, st
        assert len(edges) == n_edges == len(ref)
        assert (edges["source_1"] + 1 == ref["source_1"]).all() and (edges["source_2"] + 1 == ref["source_2"]).all()
        assert (edges["shared"] == ref["shared"]).all()
