"""Parity of the HIP engine (through the C ABI) against the CPU oracle.

Bit-exact bar: integer pair counts, identical edge sets.
"""
import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu


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
