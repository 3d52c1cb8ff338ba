"""Full-size checks on BASELINE.json's bench workload (C2: 10k sources, ~5k hashes each),
where the brute-force oracle is too slow for every pair: size-independent properties plus
sampled pair checks against direct set intersection."""
import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    sk = synth.generate("C2")
    dk = engine.DeviceBuffer.from_numpy(sk.keys)
    e = engine.Engine(0)
    e.build_blocks(dk.ptr.value, sk.offsets)
    cap = 1 << 24
    de = engine.DeviceBuffer(cap * 16)
    return sk, dk, e, de, cap


def _join(e, de, cap, t0, t1):
    cnt = e.join(t0, t1, de.ptr.value, cap)
    return np.sort(de.to_numpy(engine.EDGE_DTYPE, cnt), order=["source_1", "source_2"])


def test_c2_properties_and_sampled_pairs(c2):
    sk, dk, e, de, cap = c2
    T = e.num_tiles
    full = _join(e, de, cap, 0, T)
    n = sk.n_sources
    assert e.tile_pairs(0, T) == n * (n - 1) // 2
    # ordering / range / uniqueness
    assert (full["source_1"] < full["source_2"]).all() and full["source_2"].max() < n
    key = full["source_1"].astype(np.int64) * n + full["source_2"]
    assert (np.diff(key) > 0).all()
    # shared <= min(n_a, n_b), never zero
    sizes = sk.sizes
    assert (full["shared"] > 0).all()
    assert (full["shared"] <= np.minimum(sizes[full["source_1"]], sizes[full["source_2"]])).all()
    # idempotence: a second join gives the same multiset
    again = _join(e, de, cap, 0, T)
    assert (again == full).all()
    # partition invariance: joins over a split of the tile range unite to the same result
    cut = [0, T // 3, T // 2 + 7, T]
    parts = np.concatenate([_join(e, de, cap, cut[i], cut[i + 1]) for i in range(3)])
    parts = np.sort(parts, order=["source_1", "source_2"])
    assert (parts == full).all()
    # sampled reported pairs and sampled random pairs against direct intersection
    rng = np.random.default_rng(1)
    for i in rng.choice(len(full), size=400, replace=False):
        a, b, s = int(full["source_1"][i]), int(full["source_2"][i]), int(full["shared"][i])
        assert np.intersect1d(sk.run(a), sk.run(b), assume_unique=True).size == s
    present = set(key.tolist())
    for _ in range(400):
        a, b = sorted(rng.choice(n, size=2, replace=False).tolist())
        s = np.intersect1d(sk.run(a), sk.run(b), assume_unique=True).size
        assert (s > 0) == ((a * n + b) in present)
    # every same-cluster pair shares something in this generator; check the checksum of
    # per-source totals against an independent inverted-index count: sum_k C(m_k, 2)
    uniq, counts = np.unique(sk.keys, return_counts=True)
    assert int(full["shared"].sum()) == int((counts.astype(np.int64) * (counts - 1) // 2).sum())


def test_overflow_is_reported_not_silent(c2):
    sk, dk, e, de, cap = c2
    with pytest.raises(engine.KspError) as ei:
        e.join(0, e.num_tiles, de.ptr.value, 1000)
    assert ei.value.code == engine.KSP_E_OVERFLOW


def test_work_list_and_balanced_cuts(c2):
    """Stage 1 knows which block pairs share a key: most tiles of C2 are skipped, the edge bound is
    tighter than the pair count, and the per-GPU tile ranges of equal work unite to the full result."""
    sk, dk, e, de, cap = c2
    T = e.num_tiles
    full = _join(e, de, cap, 0, T)
    st = e.stats()
    assert st["n_tiles"] == T and 0 < st["n_active_tiles"] < T // 4
    assert st["last_active_tiles"] == st["n_active_tiles"]
    assert len(full) <= e.edge_bound(0, T) <= e.tile_pairs(0, T)
    for world in (2, 3, 8):
        cuts = e.balanced_cuts(world)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == T
        assert all(cuts[i] <= cuts[i + 1] for i in range(world))
        parts, active = [], 0
        for r in range(world):
            parts.append(_join(e, de, cap, cuts[r], cuts[r + 1]))
            assert len(parts[-1]) <= e.edge_bound(cuts[r], cuts[r + 1])
            active += e.stats()["last_active_tiles"]
        assert active == st["n_active_tiles"]
        parts = np.sort(np.concatenate(parts), order=["source_1", "source_2"])
        assert (parts == full).all()


@pytest.mark.parametrize("cfg,n,env", [("C4", 6000, {}), ("C4", 6000, {"KSP_NO_SCHED": "1"}),
                                       ("C5", 70000, {}), ("C3", 5000, {"KSP_REORDER": "0"})])
def test_other_shapes_checksum_and_samples(cfg, n, env, monkeypatch):
    """Shapes the small oracle cases do not reach (lognormal sizes with 32-bit-counter tiles, many tiny
    sketches, tiles whose search rounds find nothing): sum of all counts == sum_k C(holders_k, 2) from an
    independent host-side inverted index, plus sampled pairs by direct intersection."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sk = synth.generate(cfg, n_sources=n)
    dk = engine.DeviceBuffer.from_numpy(sk.keys)
    e = engine.Engine(0)
    e.build_blocks(dk.ptr.value, sk.offsets)
    T = e.num_tiles
    cap = int(e.edge_bound(0, T)) + 1
    de = engine.DeviceBuffer(cap * 16)
    ev = _join(e, de, cap, 0, T)
    _, counts = np.unique(sk.keys, return_counts=True)
    assert int(ev["shared"].sum()) == int((counts.astype(np.int64) * (counts - 1) // 2).sum())
    assert (ev["source_1"] < ev["source_2"]).all() and int(ev["source_2"].max()) < sk.n_sources
    key = ev["source_1"].astype(np.int64) * sk.n_sources + ev["source_2"]
    assert (np.diff(key) > 0).all()
    rng = np.random.default_rng(5)
    for i in rng.choice(len(ev), size=min(300, len(ev)), replace=False):
        a, b, s = int(ev["source_1"][i]), int(ev["source_2"][i]), int(ev["shared"][i])
        assert np.intersect1d(sk.run(a), sk.run(b), assume_unique=True).size == s


def test_join_to_host_in_pieces_equals_one_join(monkeypatch):
    """ksp_engine_join_to_host (the range cut into pieces by the edge bound, piece k copied to host memory under the
    join of piece k + 1) delivers exactly the edges of one ksp_engine_join over the same range — with pieces of one
    tile, of a few tiles, and one piece for everything; a host buffer that is too small reports the full count."""
    sk = synth.generate("C2", n_sources=3000, seed=77)
    keys_d = engine.DeviceBuffer.from_numpy(sk.keys)
    eng = engine.Engine(0)
    eng.build_blocks(keys_d.ptr.value, sk.offsets)
    T = eng.num_tiles
    cap = int(eng.edge_bound(0, T)) + 1
    ed = engine.DeviceBuffer(cap * 16)
    n = eng.join(0, T, ed.ptr.value, cap)
    want = np.sort(ed.to_numpy(engine.EDGE_DTYPE, n), order=["source_1", "source_2"])
    host = np.zeros(n + 8, dtype=engine.EDGE_DTYPE)   # (pageable memory works too: pinned memory only makes the copy faster)
    for piece in ("16384", "100000", "100000000"):
        monkeypatch.setenv("KSP_DEBUG_PIECE", piece)
        host[:] = 0
        m = eng.join_to_host(0, T, host.ctypes.data, host.size)
        assert m == n
        got = np.sort(host[:m], order=["source_1", "source_2"])
        assert (got == want).all(), piece
    # sub-range, and a buffer that is too small
    t_mid = T // 3
    n1 = eng.join(0, t_mid, ed.ptr.value, cap)
    assert eng.join_to_host(0, t_mid, host.ctypes.data, host.size) == n1
    with pytest.raises(engine.KspError) as ei:
        eng.join_to_host(0, T, host.ctypes.data, max(1, n // 2))
    assert ei.value.code == engine.KSP_E_OVERFLOW and ei.value.count == n
    eng.close()


def test_pipelined_join_reports_overflow_at_the_wait_and_step_launch_collects_it():
    """ksp_engine_join_launch into a buffer that is too small, the next build queued behind it, ksp_engine_join_wait: the
    overflow is only known at the wait (include/kspider_amd.h: size the buffer from ksp_engine_edge_bound first) and carries
    the full count.  ksp_engine_step_launch does build + range + bound + launch in one call, refuses to launch into a
    buffer below the bound, and hands back the count of the join that was pending."""
    sk = synth.generate("C2", n_sources=2000, seed=78)
    keys_d = engine.DeviceBuffer.from_numpy(sk.keys)
    eng = engine.Engine(0)
    eng.build_blocks(keys_d.ptr.value, sk.offsets)
    T = eng.num_tiles
    cap = int(eng.edge_bound(0, T)) + 1
    ed = engine.DeviceBuffer(cap * 16)
    n = eng.join(0, T, ed.ptr.value, cap)
    assert n > 1000
    eng.join_launch(0, T, ed.ptr.value, n // 3)          # too small: the surplus is dropped, the count keeps running
    eng.build_blocks(keys_d.ptr.value, sk.offsets)       # (the next step's stage 1 runs behind the join)
    with pytest.raises(engine.KspError) as ei:
        eng.join_wait()
    assert ei.value.code == engine.KSP_E_OVERFLOW and str(n) in str(ei.value)
    # step_launch: no launch below the bound; then a launch, and the next call collects its count
    t0, t1, bound, launched, prev = eng.step_launch(keys_d.ptr.value, sk.offsets, 0, 1, ed.ptr.value, 10)
    assert (t0, t1) == (0, T) and bound + 1 > 10 and not launched and prev is None
    t0, t1, bound, launched, prev = eng.step_launch(keys_d.ptr.value, sk.offsets, 0, 1, ed.ptr.value, cap)
    assert launched and prev is None and bound < cap
    t0, t1, bound, launched, prev = eng.step_launch(keys_d.ptr.value, sk.offsets, 0, 1, ed.ptr.value, cap)
    assert launched and prev == n
    assert eng.join_wait() == n
    # two ranks' ranges partition the tiles
    a = eng.step_launch(keys_d.ptr.value, sk.offsets, 0, 2, ed.ptr.value, cap)
    n0 = eng.join_wait()
    b = eng.step_launch(keys_d.ptr.value, sk.offsets, 1, 2, ed.ptr.value, cap)
    n1 = eng.join_wait()
    assert a[0] == 0 and a[1] == b[0] and b[1] == T and n0 + n1 == n
    eng.close()


@pytest.mark.parametrize("late", [False, True])
def test_step_launch_with_the_early_work_list_gives_the_reference_edges(oracle_lib, monkeypatch, late):
    """ksp_engine_step_launch copies the work list's inputs out in FRONT of the last two kernels of the build and cuts the
    join's shares while those run (KSP_DEBUG_LATE_SCHED=1: at the end of the build, as every other entry point).  Both give
    the restated reference's edge set, step after step on one engine; the build time is there when phases are being timed."""
    if late:
        monkeypatch.setenv("KSP_DEBUG_LATE_SCHED", "1")
    sk = synth.generate("C2", n_sources=1200, mean_size=900, cluster_cap=40, seed=91)
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    keys_d = engine.DeviceBuffer.from_numpy(sk.keys)
    eng = engine.Engine(0)
    eng.build_blocks(keys_d.ptr.value, sk.offsets)
    cap = 2 * int(eng.edge_bound(0, eng.num_tiles)) + 1   # (the bound moves a little from build to build: the blocks do)
    bufs = [engine.DeviceBuffer(cap * 16), engine.DeviceBuffer(cap * 16)]
    counts = []
    for step in range(4):   # (the join of step k is collected by the call of step k + 1)
        t0, t1, bound, launched, prev = eng.step_launch(keys_d.ptr.value, sk.offsets, 0, 1, bufs[step & 1].ptr.value, cap)
        assert launched and (t0, t1) == (0, eng.num_tiles)
        st = eng.stats()
        assert st["n_active_tiles"] > 0, st
        # (a step launched this way carries no timing events — each is a bubble in the stream — unless phases are being timed)
        assert (st["ms_build"] > 0) == (step == 2), st
        eng.set_profiling(step == 1)   # (the next step is timed)
        if prev is not None:
            counts.append(prev)
            got = np.sort(bufs[(step - 1) & 1].to_numpy(engine.EDGE_DTYPE, prev), order=["source_1", "source_2"])
            assert len(got) == len(ref) and (got == ref).all()
    n = eng.join_wait()
    got = np.sort(bufs[3 & 1].to_numpy(engine.EDGE_DTYPE, n), order=["source_1", "source_2"])
    assert len(got) == len(ref) and (got == ref).all()
    assert counts == [len(ref)] * 3
    eng.close()
