"""Key-range sharded stage 1 (the multi-GPU build) on one GPU: build every slice in turn, "exchange"
them through plain device buffers, assemble, join — the result must equal the single-build result and
the brute-force oracle."""
import numpy as np
import pytest

from kspider_amd import engine, synth

pytestmark = pytest.mark.gpu


def _sliced_edges(sk, nparts, weights=None):
    dk = engine.DeviceBuffer.from_numpy(sk.keys)
    dw = engine.DeviceBuffer.from_numpy(weights) if weights is not None else None
    e = engine.Engine(0)
    nb = None
    # what the ranks' MIN all-reduce does: element-wise minimum of the slices' source labels
    lab = engine.DeviceBuffer(sk.n_sources * 4)
    labels = np.full(sk.n_sources, 0xFFFFFFFF, dtype=np.uint32)
    for p in range(nparts):
        e.build_slice(dk.ptr.value, sk.offsets, p, nparts, d_weights_ptr=dw.ptr.value if dw else 0)
        e.slice_labels(lab.ptr.value)
        nb = e.stats()["n_blocks"]   # (blocks of the build: may hold spare ones for cluster-aligned boundaries)
        labels = np.minimum(labels, lab.to_numpy(np.uint32, sk.n_sources))
    lab = engine.DeviceBuffer.from_numpy(labels)
    sizes, parts = [], []
    for p in range(nparts):
        e.build_slice(dk.ptr.value, sk.offsets, p, nparts, d_weights_ptr=dw.ptr.value if dw else 0)
        e.slice_finish(lab.ptr.value)
        sz = e.slice_sizes()
        L, nbig = int(sz[0]), int(sz[2])
        bufs = dict(brk=engine.DeviceBuffer(max(4, L * 4)), info=engine.DeviceBuffer(max(4, L * 4)),
                    bw=engine.DeviceBuffer(max(4, L * 4)), raw=engine.DeviceBuffer((nb + 1) * 4),
                    pos=engine.DeviceBuffer((nb + 1) * 4), big=engine.DeviceBuffer(max(16, nbig * 16)))
        e.slice_export(bufs["brk"].ptr.value, bufs["info"].ptr.value, bufs["bw"].ptr.value, bufs["raw"].ptr.value,
                       bufs["pos"].ptr.value, bufs["big"].ptr.value)
        host = {k: b.to_numpy(np.uint8, b.nbytes) for k, b in bufs.items()}
        sizes.append(sz)
        parts.append(host)
    sizes = np.concatenate(sizes)
    lstride = max(1, int(sizes[0::4].max()))
    bigstride = max(1, int(sizes[2::4].max()))

    def stack(key, row_bytes):
        out = np.zeros((nparts, row_bytes), dtype=np.uint8)
        for p, h in enumerate(parts):
            n = min(row_bytes, h[key].size)
            out[p, :n] = h[key][:n]
        return engine.DeviceBuffer.from_numpy(out)

    brk_all, info_all, bw_all = stack("brk", lstride * 4), stack("info", lstride * 4), stack("bw", lstride * 4)
    raw_all, pos_all = stack("raw", (nb + 1) * 4), stack("pos", (nb + 1) * 4)
    big_all = stack("big", bigstride * 16)
    e.assemble(sizes, brk_all.ptr.value, info_all.ptr.value, bw_all.ptr.value, lstride, raw_all.ptr.value,
               pos_all.ptr.value, big_all.ptr.value, bigstride)
    cap = max(16, e.tile_pairs(0, e.num_tiles))
    de = engine.DeviceBuffer(cap * 16)
    cnt = e.join(0, e.num_tiles, de.ptr.value, cap)
    return np.sort(de.to_numpy(engine.EDGE_DTYPE, cnt), order=["source_1", "source_2"]), sizes


@pytest.mark.parametrize("nparts", [1, 2, 3, 8])
def test_sliced_build_equals_oracle(oracle_lib, nparts):
    sk = synth.generate("C2", n_sources=420, mean_size=500, cluster_cap=40, seed=321)
    got, sizes = _sliced_edges(sk, nparts)
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert len(got) == len(ref) and (got == ref).all()
    if nparts > 1:
        assert (sizes[3::4] > 0).all()           # every key range holds shared keys


def test_sliced_build_big_postings_weights_and_empty_slices(oracle_lib):
    # contiguous clusters -> postings with > 4 holders (masks); full 64-bit keys; 5 parts
    sk = synth.generate("C4", n_sources=300, mean_size=250, cluster_cap=60, seed=322, shuffle=False)
    got, _ = _sliced_edges(sk, 5)
    ref = oracle_lib.brute_pairs(sk.keys, sk.offsets)
    assert (got == ref).all()
    # weighted mode + keys crowded into the lowest range (most slices are empty)
    runs = [np.arange(1, 60, dtype=np.uint64) * np.uint64(3 + (s % 4)) for s in range(150)]
    runs.append(np.array([1 << 40], dtype=np.uint64))      # stretches the key range: parts 1.. are (almost) empty
    sk2 = synth.from_runs(runs)
    w = (sk2.keys % np.uint64(7) + np.uint64(1)).astype(np.uint32)
    got2, sizes2 = _sliced_edges(sk2, 4, weights=w)
    single, _ = engine.pairwise_host(sk2.keys, sk2.offsets, w)
    assert (got2 == single).all() and len(got2) > 1000
    assert (sizes2[3::4] == 0).any()
