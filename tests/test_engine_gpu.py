"""Parity of the HIP engine (through the C ABI) against the CPU oracle.

Bit-exact bar: integer pair counts, identical edge sets.
"""
import os

import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "no_reorder", "dense_walk", "collect", "lds_counters", "sort_grouping", "sort_lists",
                                      "rocprim_partition", "match_join", "plain_cuts", "library_split", "bucket_resident"])
def mode(request, monkeypatch):
    """Every case runs seven ways (the last two: keys grouped by the full sort instead of the hash buckets,
    KSP_HASH_GROUP=0; block lists by sorting the entries by block instead of key by key, KSP_KEY_GROUPS=0): as shipped (sources reordered by shared-key label, join over the
    work list of active tiles, accumulation chosen by the postings' sizes), with the caller's source
    order (KSP_REORDER=0), with the reordering but a plain walk over all tiles (KSP_NO_SCHED=1), and with
    the off-diagonal accumulation forced to the bit-sliced collect path / to the LDS counters."""
    for k in ("KSP_REORDER", "KSP_NO_SCHED", "KSP_COLLECT", "KSP_HASH_GROUP", "KSP_KEY_GROUPS", "KSP_PARTITION", "KSP_JOIN", "KSP_ALIGN", "KSP_MS", "KSP_FUSED"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "bucket_resident":   # the middle of stage 1 bucket-resident (k_fgroup / k_fkeys / k_fms_place) instead of pass by pass
        monkeypatch.setenv("KSP_FUSED", "1")
    if request.param == "library_split":   # the group records sorted by block with the library's radix sort instead of k_ms_*
        monkeypatch.setenv("KSP_MS", "0")
    if request.param == "plain_cuts":   # blocks cut every 128 sources of the label order (no spare blocks, no holes)
        monkeypatch.setenv("KSP_ALIGN", "0")
    if request.param == "match_join":   # off-diagonal tiles from stage 1's match records instead of searching the lists
        monkeypatch.setenv("KSP_JOIN", "matches")
    if request.param == "rocprim_partition":   # the library partition instead of partition_kernels.hip.h
        monkeypatch.setenv("KSP_PARTITION", "rocprim")
    if request.param == "sort_lists":
        monkeypatch.setenv("KSP_KEY_GROUPS", "0")
    if request.param == "sort_grouping":
        monkeypatch.setenv("KSP_HASH_GROUP", "0")
    if request.param == "no_reorder":
        monkeypatch.setenv("KSP_REORDER", "0")
    elif request.param == "dense_walk":
        monkeypatch.setenv("KSP_NO_SCHED", "1")
    elif request.param == "collect":
        monkeypatch.setenv("KSP_COLLECT", "1")
    elif request.param == "lds_counters":
        monkeypatch.setenv("KSP_COLLECT", "0")
    return request.param


def _check(sk, oracle, weights=None):
    edges, st = engine.pairwise_host(sk.keys, sk.offsets, weights)
    ref = oracle.brute_pairs(sk.keys, sk.offsets)
    assert edges.dtype == ref.dtype
    assert len(edges) == len(ref), (len(edges), len(ref), st)
    assert (edges == ref).all()
    return edges, st


@pytest.mark.parametrize("n,mean,cap,shuffle", [
    (1, 50, 4, True), (2, 50, 2, True), (37, 120, 8, True), (128, 200, 16, True), (129, 200, 16, False),
    (300, 400, 40, True), (300, 400, 40, False), (700, 900, 64, True),
])
def test_raw_sketches_match_brute_force(oracle_lib, n, mean, cap, shuffle):
    sk = synth.generate("C2", n_sources=n, mean_size=mean, cluster_cap=cap, shuffle=shuffle, seed=100 + n)
    _check(sk, oracle_lib)


def test_full_range_keys(oracle_lib):
    sk = synth.generate("C4", n_sources=260, mean_size=300, cluster_cap=30, seed=5)
    _check(sk, oracle_lib)


def test_extreme_key_values(oracle_lib):
    m = np.uint64(0xFFFFFFFFFFFFFFFF)
    runs = [[0, 1, 2, int(m)], [0, int(m)], [int(m)], [5], [], [1, 2, 3, 4, 5, int(m) - 1]]
    runs += [[i, i + 1, int(m) - i] for i in range(200)]
    sk = synth.from_runs(runs)
    _check(sk, oracle_lib)


def test_empty_and_ragged(oracle_lib):
    sk = synth.from_runs([[], [], []])
    edges, _ = engine.pairwise_host(sk.keys, sk.offsets)
    assert len(edges) == 0
    sk = synth.from_runs([[1, 2, 3], [], [3], [], list(range(1000)), [999]])
    _check(sk, oracle_lib)
    sk = synth.from_runs([])
    edges, _ = engine.pairwise_host(sk.keys, sk.offsets)
    assert len(edges) == 0


def test_identical_sketches_dense_overlap(oracle_lib):
    base = np.arange(1, 2000, dtype=np.uint64) * np.uint64(7919)
    runs = [base[: 1500 + (i % 7) * 50] for i in range(140)]
    sk = synth.from_runs(runs)
    edges, _ = _check(sk, oracle_lib)
    assert len(edges) == 140 * 139 // 2


@pytest.mark.parametrize("cut16", [True, False])
def test_huge_sketches_use_32bit_counters(oracle_lib, monkeypatch, cut16):
    """Pairs of sources with >= 2^16 k-mers can share >= 2^16 of them: those tiles must take
    the 32-bit counter kernel (the packed 16-bit tile would overflow) — or be cut into shares of
    fewer than 2^16 keys each, which meet in the tile's 32-bit buffer (KSP_DEBUG_NO16CUT=1: the
    host does not cut for that, the whole tile counts in 32 bits)."""
    if not cut16:
        monkeypatch.setenv("KSP_DEBUG_NO16CUT", "1")
    rng = np.random.default_rng(11)
    big = np.unique(rng.integers(1, 1 << 60, size=90000, dtype=np.uint64))
    runs = []
    for s in range(300):
        if s in (3, 7, 140, 290):          # two in block 0, one in block 1, one in block 2
            runs.append(big[: 70000 + 1000 * (s % 5)])
        else:
            own = rng.integers(1, 1 << 60, size=200, dtype=np.uint64)
            runs.append(np.concatenate([own, big[rng.integers(0, 90000, size=50)]]))
    sk = synth.from_runs(runs)
    edges, st = _check(sk, oracle_lib)
    assert int(edges["shared"].max()) >= 70000


def test_weighted_sums_beyond_16_bits(oracle_lib):
    """Colour weights whose per-source sums exceed 2^16 (32-bit counter kernel in weighted mode)."""
    n = 260
    keys, wts, offs = [], [], [0]
    for s in range(n):
        k = np.arange(1, 41, dtype=np.uint64) if s % 2 == 0 else np.arange(20, 61, dtype=np.uint64)
        keys.append(k)
        wts.append((k * 100).astype(np.uint32))      # sums ~ 80k-160k per source
        offs.append(offs[-1] + k.size)
    keys = np.concatenate(keys); wts = np.concatenate(wts); offs = np.array(offs, dtype=np.uint64)
    edges, st = engine.pairwise_host(keys, offs, wts)
    want = {}
    for a in range(n):
        for b in range(a + 1, n):
            ka = set(range(1, 41)) if a % 2 == 0 else set(range(20, 61))
            kb = set(range(1, 41)) if b % 2 == 0 else set(range(20, 61))
            want[(a, b)] = sum(100 * x for x in ka & kb)
    got = {(int(e["source_1"]), int(e["source_2"])): int(e["shared"]) for e in edges}
    assert got == want and max(want.values()) > 65535


def test_keys_sharing_their_top_bits(oracle_lib):
    """Distinct keys that agree in their top 32 significant bits: the prefix sort's fix-up path,
    and (thousands of them) the fall-back to a full-width sort."""
    rng = np.random.default_rng(12)
    base = np.uint64(0xABCDEF0100000000)
    # small mixed runs: pairs/triples of keys differing only in low bits, spread over sources
    lows = rng.integers(0, 1 << 20, size=(400, 3), dtype=np.uint64)
    his = (rng.integers(1, 1 << 30, size=400, dtype=np.uint64) << np.uint64(32))
    pool = (his[:, None] | lows).ravel()
    runs = [np.unique(pool[rng.integers(0, pool.size, size=150)]) for _ in range(200)]
    _check(synth.from_runs(runs), oracle_lib)
    # one giant run: 5000 distinct keys with identical top 32 bits -> overflow -> full sort
    giant = base | rng.integers(0, 1 << 31, size=5000, dtype=np.uint64)
    runs = [np.unique(giant[rng.integers(0, giant.size, size=300)]) for _ in range(150)]
    _check(synth.from_runs(runs), oracle_lib)


def test_bucket_grouping_and_its_fallback(oracle_lib, mode):
    """Uniform hashes are grouped in LDS hash buckets (two partition passes); a bucket above the table's entry
    capacity (one key held by thousands of sources) is streamed by the big-bucket kernel; thousands of
    distinct keys under one prefix send the build back to the sort path.  Same edges every way."""
    rng = np.random.default_rng(31)
    sk = synth.generate("C2", n_sources=400, mean_size=600, cluster_cap=20, seed=77)
    _, st = _check(sk, oracle_lib)
    if mode != "sort_grouping":
        assert 0 < st["sort_bits"] <= 16, st["sort_bits"]        # partitioned, not sorted
    else:
        assert st["sort_bits"] >= 32
    # full 64-bit hashes (the partition skips bit 63) with the largest key — the hash table's "empty" pattern —
    # and its neighbours held by many sources
    top = np.uint64(0xFFFFFFFFFFFFFFFF)
    runs = []
    for s_ in range(120):
        wide = rng.integers(0, 1 << 64, size=60, dtype=np.uint64)
        extra = [top] if s_ % 2 == 0 else [top - np.uint64(1)]
        if s_ % 3 == 0:
            extra.append(np.uint64(0))
        if s_ % 5 == 0:
            extra.append(np.uint64(1) << np.uint64(63))
        runs.append(np.unique(np.concatenate([wide, np.array(extra, dtype=np.uint64)])))
    _, st = _check(synth.from_runs(runs), oracle_lib)
    if mode != "sort_grouping":
        assert 0 < st["sort_bits"] <= 16, st["sort_bits"]
    # 6000 distinct keys under one 30-bit prefix, among uniform ones
    crowd = (np.uint64(0x2345678) << np.uint64(30)) | rng.integers(0, 1 << 30, size=6000, dtype=np.uint64)
    runs = []
    for s in range(300):
        wide = rng.integers(0, 1 << 58, size=200, dtype=np.uint64)
        runs.append(np.unique(np.concatenate([wide, crowd[rng.integers(0, crowd.size, size=120)]])))
    _, st = _check(synth.from_runs(runs), oracle_lib)
    assert st["sort_bits"] >= 32
    # one key held by every one of 3500 sources: its bucket is grouped by the streaming big-bucket kernel
    runs = [np.unique(np.concatenate([[np.uint64(123456789)], rng.integers(0, 1 << 50, size=3, dtype=np.uint64)]))
            for _ in range(3500)]
    _, st = _check(synth.from_runs(runs), oracle_lib)
    if mode != "sort_grouping":
        assert 0 < st["sort_bits"] <= 16, st["sort_bits"]


def test_keys_with_thousands_of_holders(oracle_lib):
    """Conserved keys of a large same-species collection: more holders than the key-by-key list build stages
    in LDS (2 048) — one workgroup per such key ORs the holders into per-block masks."""
    rng = np.random.default_rng(43)
    n = 2600
    everywhere = rng.integers(0, 1 << 58, size=3, dtype=np.uint64)
    most = rng.integers(0, 1 << 58, size=4, dtype=np.uint64)
    runs = []
    for s_ in range(n):
        own = rng.integers(0, 1 << 58, size=6, dtype=np.uint64)
        keep = most[rng.random(most.size) < 0.85]
        runs.append(np.unique(np.concatenate([own, everywhere, keep])))
    _check(synth.from_runs(runs), oracle_lib)


def test_weighted_keys_with_thousands_of_holders():
    """Colour weights through the big-bucket / huge-key kernels and the wave-per-key walk: two keys held by all
    2 300 sources (weights 7 and 1 000), one held by the first 100 (weight 50), private keys besides."""
    rng = np.random.default_rng(44)
    n = 2300
    a_key, b_key, c_key = np.uint64(3) << np.uint64(50), np.uint64(5) << np.uint64(49), np.uint64(9) << np.uint64(48)
    keys, wts, offs = [], [], [0]
    for s_ in range(n):
        own = np.unique(rng.integers(0, 1 << 45, size=5, dtype=np.uint64))
        k = np.concatenate([own, [c_key] if s_ < 100 else [], [b_key, a_key]]).astype(np.uint64)
        w = np.concatenate([np.full(own.size, 3), [50] if s_ < 100 else [], [1000, 7]]).astype(np.uint32)
        order = np.argsort(k)
        keys.append(k[order]); wts.append(w[order]); offs.append(offs[-1] + k.size)
    keys = np.concatenate(keys); wts = np.concatenate(wts); offs = np.array(offs, dtype=np.uint64)
    edges, _ = engine.pairwise_host(keys, offs, wts)
    assert len(edges) == n * (n - 1) // 2
    s1, s2, sh = edges["source_1"].astype(np.int64), edges["source_2"].astype(np.int64), edges["shared"].astype(np.int64)
    assert (s1 < s2).all()
    want = 1007 + 50 * ((s1 < 100) & (s2 < 100))
    assert (sh == want).all()


def test_active_tiles_are_exactly_the_block_pairs_that_share_a_key(oracle_lib, monkeypatch):
    """In the caller's order (KSP_REORDER=0) the blocks are source // 128, so the work list can be checked
    against the sketches: a tile is active iff its two blocks share a key (diagonal: a key with two holders
    in the block).  (The bitmap's first tiles once lost their flags to a trailing wave of k_pack_flags.)"""
    monkeypatch.setenv("KSP_REORDER", "0")
    rng = np.random.default_rng(41)
    for n, share in ((300, 0.02), (1100, 0.004), (2300, 0.0015)):
        pool = rng.integers(0, 1 << 60, size=4000, dtype=np.uint64)
        runs = [np.unique(np.concatenate([rng.integers(0, 1 << 60, size=30, dtype=np.uint64),
                                          pool[rng.random(pool.size) < share]])) for _ in range(n)]
        sk = synth.from_runs(runs)
        _, st = _check(sk, oracle_lib)
        if os.environ.get("KSP_NO_SCHED"):
            continue
        src = np.repeat(np.arange(n), np.diff(sk.offsets).astype(np.int64))
        order = np.argsort(sk.keys, kind="stable")
        ks, bs = sk.keys[order], (src[order] // 128)
        want = set()
        bounds = np.flatnonzero(np.diff(ks)) + 1
        for grp in np.split(bs, bounds):
            if grp.size < 2:
                continue
            u, c = np.unique(grp, return_counts=True)
            for x in range(u.size):
                if c[x] >= 2:
                    want.add((int(u[x]), int(u[x])))
                for y in range(x + 1, u.size):
                    want.add((int(u[x]), int(u[y])))
        nb = (n + 127) // 128
        if st["n_active_tiles"] != nb * (nb + 1) // 2:      # (mostly-dense inputs walk all tiles)
            assert st["n_active_tiles"] == len(want), (n, st["n_active_tiles"], len(want))


def _clustered(rng, sizes, n_core=40, n_own=6):
    """Sources in shuffled order; cluster c's members share most of its core keys."""
    member = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(member)
    cores = [rng.integers(0, 1 << 58, size=n_core, dtype=np.uint64) for _ in sizes]
    runs = []
    for c in member:
        own = rng.integers(0, 1 << 58, size=n_own, dtype=np.uint64)
        keep = cores[c][rng.random(n_core) < 0.9] if sizes[c] > 1 else cores[c][:0]
        runs.append(np.unique(np.concatenate([own, keep])))
    return synth.from_runs(runs), member


def test_block_boundaries_respect_the_clusters(oracle_lib):
    """k_pack_blocks: a cluster of <= 128 related sources is not cut by a block boundary while the spare slots last
    (the build holds half as many blocks again as the sources need); bigger clusters and an exhausted budget fall back to
    plain cuts.  Holes never show: the pair count of all tiles is N (N - 1) / 2 and the edges equal the oracle's."""
    if os.environ.get("KSP_REORDER") == "0":
        pytest.skip("boundaries follow the label order")
    rng = np.random.default_rng(53)
    aligned = os.environ.get("KSP_ALIGN") != "0"
    # (a) clusters that fit: 100 + 27 never share a block with a cut in between; (b) 65-source clusters: one per block
    # costs 63 holes, the budget (half the blocks) runs out; (c) clusters above 128 sources are cut wherever they fall
    for sizes, fits in (([100] * 12 + [27] * 9 + [1] * 40 + [128, 2, 3], True), ([65] * 40, False), ([300, 129, 90, 200, 50, 17] * 2, None)):
        sk, member = _clustered(rng, sizes)
        n = sk.n_sources
        dk = engine.DeviceBuffer.from_numpy(sk.keys)
        e = engine.Engine(0)
        e.build_blocks(dk.ptr.value, sk.offsets)
        st = e.stats()
        nb0 = (n + 127) // 128
        assert st["n_blocks"] == (nb0 + nb0 // 2 + 1 if aligned else nb0)
        T = e.num_tiles
        assert T == st["n_blocks"] * (st["n_blocks"] + 1) // 2
        assert e.tile_pairs(0, T) == n * (n - 1) // 2
        slot = e.source_order(n)
        assert len(np.unique(slot)) == n and slot.max() < st["n_blocks"] * 128
        blocks_of = [np.unique(slot[member == c] // 128) for c in range(len(sizes))]
        whole = sum(1 for c, b in enumerate(blocks_of) if sizes[c] <= 128 and len(b) == 1)
        small = sum(1 for c in sizes if c <= 128)
        if aligned and fits:
            assert whole == small, (whole, small)
        if aligned and fits is False:
            assert small // 2 <= whole < small      # (the budget: 1.5 x 21 + 1 = 32 blocks for 40 clusters of 65)
        cap = e.edge_bound(0, T) + 1
        de = engine.DeviceBuffer(cap * 16)
        cnt = e.join(0, T, de.ptr.value, cap)
        got = np.sort(de.to_numpy(engine.EDGE_DTYPE, cnt), order=["source_1", "source_2"])
        ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
        assert len(got) == len(ref) and (got == ref).all()
        de.free(); dk.free(); e.close()


def test_rank_skew_takes_the_oversized_cell_path(oracle_lib):
    """One block's keys crowd a narrow rank band while the others spread out: cells of that band hold
    far more than one window of keys (multi-chunk x multi-window fallback of the cell join)."""
    rng = np.random.default_rng(13)
    runs = []
    for s in range(128):                       # block 0: dense band
        runs.append(np.unique(rng.integers(0, 400_000, size=3000, dtype=np.uint64)))
    for s in range(260):                       # blocks 1..3: wide spread + a few band keys
        wide = rng.integers(1 << 20, 1 << 58, size=1500, dtype=np.uint64)
        band = rng.integers(0, 400_000, size=300, dtype=np.uint64)
        runs.append(np.unique(np.concatenate([wide, band])))
    sk = synth.from_runs(runs)
    _check(sk, oracle_lib)


def test_label_structures(oracle_lib):
    """Inputs that stress the source ordering: a chain (i shares one key with i + 1 only), one key held
    by every source, sources without any shared key, and two interleaved families."""
    n = 400
    runs = []
    for i in range(n):
        own = [10_000_000 + 17 * i, 20_000_000 + 31 * i]          # unique to i
        chain = [1_000 + i, 1_000 + i + 1]                          # shared with i - 1 and i + 1
        fam = [5_000_000 + (i % 2) * 1000 + k for k in range(6)]    # two families, interleaved ids
        hub = [42] if i % 3 else []                                 # one key in two thirds of the sources
        runs.append(sorted(set(own + chain + fam + hub)))
    runs += [[90_000_000 + i] for i in range(40)]                   # loners
    _check(synth.from_runs(runs), oracle_lib)


def test_postings_input_equals_sketch_input(oracle_lib):
    """The inverted-index entry (what the drop-in path feeds) against the sketch entry and the oracle:
    unweighted and weighted, keys in scrambled order, holders of a key in scrambled order."""
    sk = synth.generate("C2", n_sources=500, mean_size=400, cluster_cap=60, seed=909)
    n = sk.n_sources
    src = np.repeat(np.arange(n, dtype=np.uint32), np.diff(sk.offsets).astype(np.int64))
    order = np.argsort(sk.keys, kind="stable")
    k, s = sk.keys[order], src[order]
    uniq, start, cnt = np.unique(k, return_index=True, return_counts=True)
    rng = np.random.default_rng(3)
    groups = [rng.permutation(s[a:a + c]) for a, c in zip(start, cnt) if c >= 2]
    wts = rng.integers(0, 50, size=len(groups), dtype=np.uint32)           # some zero weights
    perm = rng.permutation(len(groups))
    groups = [groups[i] for i in perm]
    wts = wts[perm]
    key_off = np.zeros(len(groups) + 1, dtype=np.uint64)
    key_off[1:] = np.cumsum([g.size for g in groups])
    sources = np.concatenate(groups).astype(np.uint32)
    edges, st = engine.pairwise_postings_host(key_off, sources, None, n)
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert len(edges) == len(ref) and (edges == ref).all()
    # weighted: sum of the weights of the shared keys
    edges_w, st = engine.pairwise_postings_host(key_off, sources, wts, n)
    want = {}
    for g, w in zip(groups, wts):
        gs = np.sort(g)
        for x in range(gs.size):
            for y in range(x + 1, gs.size):
                want[(int(gs[x]), int(gs[y]))] = want.get((int(gs[x]), int(gs[y])), 0) + int(w)
    want = {p: v for p, v in want.items() if v}
    got = {(int(e["source_1"]), int(e["source_2"])): int(e["shared"]) for e in edges_w}
    assert got == want
    # no key at all
    e0, _ = engine.pairwise_postings_host(np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint32), None, n)
    assert len(e0) == 0


def test_postings_rejects_bad_source_index():
    dsrc = engine.DeviceBuffer.from_numpy(np.array([0, 1, 2, 7], dtype=np.uint32))   # 7 >= n_sources
    e = engine.Engine(0)
    with pytest.raises(engine.KspError) as ei:
        e.build_postings(np.array([0, 2, 4], dtype=np.uint64), dsrc.ptr.value, 0, 5)
    assert ei.value.code == engine.KSP_E_ARG
    with pytest.raises(engine.KspError):
        e.build_postings(np.array([0, 1, 4], dtype=np.uint64), dsrc.ptr.value, 0, 8)   # a key with one holder


def test_join_in_two_halves_with_the_next_build_queued_behind_it(oracle_lib):
    """ksp_engine_join_launch / _wait (what bench.py pipelines): the join of set A is launched, the build of set B is
    queued on the same stream behind it, and only then is A's count collected — A's edges are A's (the build of B
    overwrites the engine's lists only after the join has run), B's join after that is B's."""
    if os.environ.get("KSP_NO_SCHED"):
        pytest.skip("one ordering is enough")
    a = synth.generate("C2", n_sources=700, mean_size=500, cluster_cap=50, seed=611)
    b = synth.generate("C2", n_sources=450, mean_size=800, cluster_cap=30, seed=612)
    ref_a, ref_b = oracle_lib.brute_pairs(a.keys, a.offsets), oracle_lib.brute_pairs(b.keys, b.offsets)
    da, db = engine.DeviceBuffer.from_numpy(a.keys), engine.DeviceBuffer.from_numpy(b.keys)
    e = engine.Engine(0)
    cap = 1 << 20
    out_a, out_b = engine.DeviceBuffer(cap * 16), engine.DeviceBuffer(cap * 16)
    for _ in range(3):
        e.build_blocks(da.ptr.value, a.offsets)
        e.join_launch(0, e.num_tiles, out_a.ptr.value, cap)
        e.build_blocks(db.ptr.value, b.offsets)          # queued behind the launched join
        cnt_a = e.join_wait()
        e.join_launch(0, e.num_tiles, out_b.ptr.value, cap)
        cnt_b = e.join_wait()
        assert e.join_wait() == 0                        # nothing pending: no count
        got_a = np.sort(out_a.to_numpy(engine.EDGE_DTYPE, cnt_a), order=["source_1", "source_2"])
        got_b = np.sort(out_b.to_numpy(engine.EDGE_DTYPE, cnt_b), order=["source_1", "source_2"])
        assert len(got_a) == len(ref_a) and (got_a == ref_a).all()
        assert len(got_b) == len(ref_b) and (got_b == ref_b).all()
    for x in (out_a, out_b, da, db):
        x.free()
    e.close()
