"""Deterministic synthetic sketch sets shaped like BASELINE.json's five configs.

Follows SURVEY.md §8(d): every random draw comes from a stateless splitmix64
stream seeded with ``20241008 + config#``; sources belong to clusters whose
sizes follow a capped Zipf(1.2); a sketch mixes hashes sampled from its
cluster's core pool with hashes unique to the source; 1 % of the sources also
borrow from a second cluster.  Sketches are stored the way the engine wants
them in HBM: one concatenated array of sorted, unique uint64 hashes plus a
CSR offsets array (one run per source).

Only the *shape* of the reference's inputs is modelled (what
`src/sourmash_indexing.cpp:187` / `src/index.cpp:227` call a source's k-mer
set); k-mer extraction itself lives in kProcessor, which is out of scope.
"""
from __future__ import annotations

import dataclasses
import math
import os

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """Vectorised splitmix64 finaliser (uint64 in, uint64 out)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def _stream(seed: int, stream: int, idx):
    """uint64 pseudo-random values for (seed, stream, idx[...])."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)))
        return splitmix64(base + np.asarray(idx, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D))


def _uniform01(seed: int, stream: int, idx):
    return (_stream(seed, stream, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


@dataclasses.dataclass
class SketchSet:
    """CSR of sorted-unique uint64 hashes, one run per source (dense index 0..N-1)."""

    keys: np.ndarray      # uint64 [sum n_s]
    offsets: np.ndarray   # uint64 [N+1]
    cluster: np.ndarray   # int32 [N]  (generator metadata, not used by the engine)
    name: str = ""

    @property
    def n_sources(self) -> int:
        return int(self.offsets.shape[0] - 1)

    @property
    def sizes(self) -> np.ndarray:
        return np.diff(self.offsets).astype(np.int64)

    def run(self, s: int) -> np.ndarray:
        return self.keys[int(self.offsets[s]):int(self.offsets[s + 1])]

    def subset(self, n: int) -> "SketchSet":
        """First ``n`` sources (used for the bounded CPU-baseline sample)."""
        n = min(n, self.n_sources)
        end = int(self.offsets[n])
        return SketchSet(self.keys[:end].copy(), self.offsets[: n + 1].copy(),
                         self.cluster[:n].copy(), f"{self.name}[:{n}]")

    def algorithmic_bytes(self, weighted: bool = False) -> int:
        """A = 8*(N-1)*sum(n) + 4*N(N-1)/2 (12 B/elem when weighted)."""
        n = self.n_sources
        per = 12 if weighted else 8
        return per * (n - 1) * int(self.offsets[-1]) + 4 * (n * (n - 1) // 2)


# name -> (config#, N, size law, H_max, max cluster size rule)
CONFIGS = {
    "C1": dict(idx=1, n=1_000, size=("const", 980), hmax=4 ** 21, cap_frac=0.01),
    "C2": dict(idx=2, n=10_000, size=("normal", 5_000, 1_500, 500), hmax=(1 << 64) // 1000, cap_frac=0.01),
    "C3": dict(idx=3, n=100_000, size=("normal", 5_000, 1_000, 1_000), hmax=(1 << 64) // 1000, cap_frac=0.01),
    "C4": dict(idx=4, n=50_000, size=("lognormal", math.log(2_000), 1.5, 16, 2_000_000), hmax=(1 << 64) - 1, cap_frac=0.01),
    "C5": dict(idx=5, n=1_000_000, size=("uniform", 16, 256), hmax=(1 << 64) - 1, cap_abs=64),
}


def _sizes(law, seed, n):
    u1 = _uniform01(seed, 11, np.arange(n))
    u2 = _uniform01(seed, 12, np.arange(n))
    if law[0] == "const":
        return np.full(n, law[1], dtype=np.int64)
    if law[0] == "uniform":
        lo, hi = law[1], law[2]
        return (lo + np.floor(u1 * (hi - lo + 1))).astype(np.int64).clip(lo, hi)
    z = np.sqrt(-2.0 * np.log(np.maximum(u1, 1e-300))) * np.cos(2.0 * math.pi * u2)
    if law[0] == "normal":
        _, mu, sd, lo = law
        return np.maximum(lo, np.rint(mu + sd * z)).astype(np.int64)
    if law[0] == "lognormal":
        _, mu, sd, lo, hi = law
        return np.clip(np.rint(np.exp(mu + sd * z)), lo, hi).astype(np.int64)
    raise ValueError(law)


def generate(config: str = "C2", n_sources: int | None = None, seed: int | None = None,
             shuffle: bool = True, mean_size: int | None = None,
             cluster_cap: int | None = None, workers: int | None = None) -> SketchSet:
    """Build the sketch set for one BASELINE config.

    ``n_sources`` overrides N (tests shrink the configs; the multi-GPU bench grows
    C2 by sqrt(#GPUs) so that per-GPU pair count stays fixed).  ``mean_size``
    rescales the size law (tests only).
    """
    cfg = CONFIGS[config]
    n = int(n_sources if n_sources is not None else cfg["n"])
    seed = int(seed if seed is not None else 20241008 + cfg["idx"])
    hmax = np.uint64(cfg["hmax"])
    law = cfg["size"]
    if mean_size is not None:
        if law[0] == "const":
            law = ("const", mean_size)
        elif law[0] == "normal":
            f = mean_size / law[1]
            law = ("normal", mean_size, max(1, int(law[2] * f)), max(1, int(law[3] * f)))
        elif law[0] == "lognormal":
            law = ("lognormal", math.log(mean_size), law[2], max(1, law[3] * mean_size // 2000), max(2, law[4] * mean_size // 2000))
        elif law[0] == "uniform":
            law = ("uniform", max(1, mean_size // 8), 2 * mean_size)
    sizes = _sizes(law, seed, n)

    # --- clusters: Zipf(1.2) sizes (inverse CDF of the Pareto envelope), capped.
    cap = int(cfg["cap_abs"]) if "cap_abs" in cfg else max(2, int(n * cfg["cap_frac"]))
    if cluster_cap is not None:
        cap = int(cluster_cap)
    csz = []
    tot, c = 0, 0
    while tot < n:
        u = float(_uniform01(seed, 21, np.array([c]))[0])
        k = int(min(cap, max(1.0, math.floor(max(u, 1e-12) ** (-1.0 / 0.2)))))
        k = min(k, n - tot)
        csz.append(k)
        tot += k
        c += 1
    csz = np.asarray(csz, dtype=np.int64)
    cluster_of_slot = np.repeat(np.arange(len(csz), dtype=np.int32), csz)
    if shuffle:
        perm = np.argsort(_stream(seed, 22, np.arange(n)), kind="stable")
        cluster = np.empty(n, dtype=np.int32)
        cluster[perm] = cluster_of_slot
    else:
        cluster = cluster_of_slot
    nclusters = len(csz)

    d = 0.05 + 0.55 * _uniform01(seed, 31, np.arange(n))           # unique fraction
    borrow = _uniform01(seed, 32, np.arange(n)) < 0.01               # cross-links
    second = (_stream(seed, 33, np.arange(n)) % np.uint64(max(1, nclusters))).astype(np.int32)

    order = np.argsort(cluster, kind="stable")
    bounds = np.searchsorted(cluster[order], np.arange(nclusters + 1))
    mean_n = float(sizes.mean())
    ctx = (seed, hmax, sizes, d, borrow, second, order, bounds, mean_n)
    # clusters are independent (every draw is a stateless function of (seed, stream, index)): large sets are
    # generated by a few forked workers, each taking a contiguous range of clusters of about equal cost; the
    # result does not depend on the number of workers
    if workers is None:
        workers = int(os.environ.get("KSP_SYNTH_WORKERS", "0")) or min(12, len(os.sched_getaffinity(0)))
    if n < 20_000 or nclusters < 4 * workers:
        workers = 1
    if workers > 1:
        cmax = np.maximum.reduceat(sizes[order], bounds[:-1][np.diff(bounds) > 0]) if n else np.zeros(0)
        nonempty = np.flatnonzero(np.diff(bounds) > 0)
        cost = np.zeros(nclusters)
        cost[nonempty] = np.diff(bounds)[nonempty] * np.maximum(mean_n, cmax) + 2000.0 * np.diff(bounds)[nonempty]
        acc = np.cumsum(cost)
        parts_n = 4 * workers
        cuts = [0] + [int(np.searchsorted(acc, acc[-1] * k / parts_n)) for k in range(1, parts_n)] + [nclusters]
        tasks = [(cuts[k], cuts[k + 1]) for k in range(parts_n) if cuts[k + 1] > cuts[k]]
        global _GEN_CTX
        _GEN_CTX = ctx
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool_:
            results = pool_.map(_gen_clusters, tasks, chunksize=1)
        _GEN_CTX = None
    else:
        results = [_gen_range(ctx, 0, nclusters)]
    lens = np.zeros(n, dtype=np.int64)
    for ids, ls, _ in results:
        lens[ids] = ls
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens, dtype=np.uint64)
    keys = np.empty(int(offsets[-1]), dtype=np.uint64)
    for ids, ls, ks in results:
        pos = 0
        for s, l in zip(ids.tolist(), ls.tolist()):
            keys[int(offsets[s]):int(offsets[s]) + l] = ks[pos:pos + l]
            pos += l
    return SketchSet(keys, offsets, cluster, name=f"{config}(N={n})")


_GEN_CTX = None


def _gen_clusters(task):
    return _gen_range(_GEN_CTX, task[0], task[1])


def _gen_range(ctx, c_lo: int, c_hi: int):
    """Sketches of clusters [c_lo, c_hi): (source ids, run lengths, concatenated sorted-unique runs)."""
    seed, hmax, sizes, d, borrow, second, order, bounds, mean_n = ctx
    ids, lens, runs = [], [], []
    for c in range(c_lo, c_hi):
        members = order[bounds[c]:bounds[c + 1]]
        if members.size == 0:
            continue
        pool_n = int(max(8, 1.25 * max(mean_n, float(sizes[members].max()))))
        j = np.arange(pool_n, dtype=np.uint64)
        pool = _stream(seed, 1000 + 2 * c, j) % hmax
        for s in members:
            s = int(s)
            ns = int(sizes[s])
            p = min(1.0, (1.0 - d[s]) * ns / pool_n)
            pick = _uniform01(seed ^ (s * 0x9E3779B1 & 0x7FFFFFFF), 41, j) < p
            parts = [pool[pick]]
            n_u = max(0, ns - int(pick.sum()))
            if n_u:
                parts.append(_stream(seed ^ (s * 0x85EBCA6B & 0x7FFFFFFF), 42, np.arange(n_u)) % hmax)
            if borrow[s] and second[s] != c:
                c2 = int(second[s])
                pool2_n = pool_n
                j2 = np.arange(pool2_n, dtype=np.uint64)
                pool2 = _stream(seed, 1000 + 2 * c2, j2) % hmax
                pick2 = _uniform01(seed ^ (s * 0xC2B2AE35 & 0x7FFFFFFF), 43, j2) < min(1.0, 0.1 * ns / pool2_n)
                parts.append(pool2[pick2])
            r = np.unique(np.concatenate(parts))
            ids.append(s)
            lens.append(r.size)
            runs.append(r)
    return (np.asarray(ids, dtype=np.int64), np.asarray(lens, dtype=np.int64),
            np.concatenate(runs).astype(np.uint64) if runs else np.zeros(0, dtype=np.uint64))


def from_runs(runs, name: str = "custom") -> SketchSet:
    """SketchSet from explicit per-source iterables of hashes (sorted + de-duplicated here)."""
    rs = [np.unique(np.asarray(list(r), dtype=np.uint64)) for r in runs]
    offsets = np.zeros(len(rs) + 1, dtype=np.uint64)
    if rs:
        offsets[1:] = np.cumsum([r.size for r in rs], dtype=np.uint64)
    keys = np.concatenate(rs).astype(np.uint64) if rs else np.zeros(0, dtype=np.uint64)
    return SketchSet(keys, offsets, np.zeros(len(rs), dtype=np.int32), name)
