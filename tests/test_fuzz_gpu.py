"""Seeded random inputs against the oracle, in every engine mode: small key spaces force heavy overlap
(large postings, dense tiles), large ones sparse tiles; sizes are ragged, some sources are empty; weighted
cases exercise the LDS-counter path, the postings entry gets the same data as an inverted index."""
import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu

MODES = [{}, {"KSP_REORDER": "0"}, {"KSP_NO_SCHED": "1"}, {"KSP_COLLECT": "1"}, {"KSP_COLLECT": "0"},
         {"KSP_JOIN": "window"}, {"KSP_TAG32": "1"}, {"KSP_HASH_GROUP": "0"}, {"KSP_KEY_GROUPS": "0"},
         {"KSP_KEY_GROUPS": "0", "KSP_REORDER": "0"}, {"KSP_PART_MIN": "1"}, {"KSP_PARTITION": "rocprim"}, {"KSP_JOIN": "matches"}, {"KSP_JOIN": "matches", "KSP_COLLECT": "0"},
         {"KSP_JOIN": "matches", "KSP_COLLECT": "1", "KSP_REORDER": "0"}, {"KSP_ALIGN": "0"}, {"KSP_ALIGN": "0", "KSP_KEY_GROUPS": "0"}, {"KSP_PART_MIN": "1", "KSP_SEG": "1"}, {"KSP_PART_MIN": "1", "KSP_SEG": "0"}, {"KSP_MS": "0"}, {"KSP_MS": "0", "KSP_REORDER": "0"},
         # the bucket-resident middle of stage 1 (fused_kernels.hip.h): tiny inputs too (KSP_PART_MIN=1), paged and segment partition,
         # 32-bit tags, plain block cuts, one bucket per chunk of k_fkeys and many, sorted level 1 of the paged partition
         {"KSP_FUSED": "1", "KSP_PART_MIN": "1"}, {"KSP_FUSED": "1", "KSP_PART_MIN": "1", "KSP_SEG": "1"},
         {"KSP_FUSED": "1", "KSP_PART_MIN": "1", "KSP_SEG": "0", "KSP_TAG32": "1"}, {"KSP_FUSED": "1", "KSP_PART_MIN": "1", "KSP_ALIGN": "0"},
         {"KSP_FUSED": "1", "KSP_PART_MIN": "1", "KSP_DEBUG_FK_GB": "1"}, {"KSP_FUSED": "1", "KSP_PART_MIN": "1", "KSP_DEBUG_FK_GB": "32", "KSP_COLLECT": "0"},
         {"KSP_PART_MIN": "1", "KSP_SEG": "0", "KSP_DEBUG_PART_SORTED": "1"}, {"KSP_DEBUG_LABEL_SPREAD": "0"}]


def _random_sketches(rng):
    n = int(rng.choice([1, 2, 3, 17, 64, 127, 128, 129, 200, 257, 390]))
    universe = int(rng.choice([8, 60, 500, 5000, 1 << 20, 1 << 40]))
    mean = int(rng.choice([1, 5, 40, 300]))
    fam = max(1, int(rng.choice([1, 3, 20, n])))
    runs = []
    base = [np.unique(rng.integers(0, universe, size=max(1, mean), dtype=np.uint64)) for _ in range(fam)]
    for s in range(n):
        if rng.random() < 0.05:
            runs.append(np.zeros(0, dtype=np.uint64))
            continue
        own = rng.integers(0, universe, size=int(rng.integers(0, 2 * mean + 1)), dtype=np.uint64)
        shared = base[s % fam][rng.random(base[s % fam].size) < rng.random()]
        runs.append(np.unique(np.concatenate([own, shared])))
    return synth.from_runs(runs)


@pytest.mark.parametrize("seed", range(12))
def test_random_sketches_all_modes(oracle_lib, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(6):
        sk = _random_sketches(rng)
        ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
        for env in MODES:
            for k in ("KSP_REORDER", "KSP_NO_SCHED", "KSP_COLLECT", "KSP_JOIN", "KSP_TAG32", "KSP_HASH_GROUP", "KSP_KEY_GROUPS",
                      "KSP_PART_MIN", "KSP_PARTITION", "KSP_ALIGN", "KSP_SEG", "KSP_MS", "KSP_FUSED", "KSP_DEBUG_FK_GB",
                      "KSP_DEBUG_PART_SORTED", "KSP_DEBUG_LABEL_SPREAD"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            edges, _ = engine.pairwise_host(sk.keys, sk.offsets)
            assert len(edges) == len(ref) and (edges == ref).all(), (seed, env, sk.n_sources)


@pytest.mark.parametrize("seed", range(6))
def test_random_weighted_and_postings(oracle_lib, seed, monkeypatch):
    rng = np.random.default_rng(2000 + seed)
    for _ in range(4):
        sk = _random_sketches(rng)
        n = sk.n_sources
        if sk.keys.size == 0:
            continue
        src = np.repeat(np.arange(n, dtype=np.uint32), np.diff(sk.offsets).astype(np.int64))
        uniq, inv = np.unique(sk.keys, return_inverse=True)
        wkey = rng.integers(0, 1000, size=uniq.size, dtype=np.uint32)
        # expected: sum of the weights of the shared keys (pairs whose sum is 0 are not reported)
        want = {}
        order = np.argsort(inv, kind="stable")
        ks, ss = inv[order], src[order]
        bounds = np.flatnonzero(np.diff(ks)) + 1
        groups = np.split(ss, bounds)
        gkeys = ks[np.concatenate([[0], bounds])] if ks.size else []
        for g, kidx in zip(groups, gkeys):
            w = int(wkey[kidx])
            if g.size < 2 or w == 0:
                continue
            g = np.sort(g)
            for x in range(g.size):
                for y in range(x + 1, g.size):
                    want[(int(g[x]), int(g[y]))] = want.get((int(g[x]), int(g[y])), 0) + w
        for env in ({}, {"KSP_REORDER": "0"}, {"KSP_NO_SCHED": "1"}, {"KSP_KEY_GROUPS": "0"}):
            for k in ("KSP_REORDER", "KSP_NO_SCHED", "KSP_KEY_GROUPS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            edges, _ = engine.pairwise_host(sk.keys, sk.offsets, wkey[inv])
            got = {(int(e["source_1"]), int(e["source_2"])): int(e["shared"]) for e in edges}
            assert got == want, (seed, env)
            # the same data as an inverted index (keys with >= 2 holders, scrambled)
            big = [(g, int(wkey[kidx])) for g, kidx in zip(groups, gkeys) if g.size >= 2]
            if big:
                perm = rng.permutation(len(big))
                key_off = np.zeros(len(big) + 1, dtype=np.uint64)
                key_off[1:] = np.cumsum([big[i][0].size for i in perm])
                sources = np.concatenate([rng.permutation(big[i][0]) for i in perm]).astype(np.uint32)
                wts = np.array([big[i][1] for i in perm], dtype=np.uint32)
                e2, _ = engine.pairwise_postings_host(key_off, sources, wts, n)
                got2 = {(int(e["source_1"]), int(e["source_2"])): int(e["shared"]) for e in e2}
                assert got2 == want, (seed, env, "postings")
