"""Diagnostic: rebuild + join the C4-shape input many times (the key ranks, hence the block layout, differ from
build to build) and report any run whose edges differ from the first one."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sk = synth.generate(cfg, n_sources=n)
_, counts = np.unique(sk.keys, return_counts=True)
want = int((counts.astype(np.int64) * (counts - 1) // 2).sum())
dk = engine.DeviceBuffer.from_numpy(sk.keys)
junk = [engine.DeviceBuffer.from_numpy(np.full(1 << 24, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)) for _ in range(8)]
del junk   # freed device memory now holds a non-zero pattern
ref = None
bad = 0
for it in range(iters):
    e = engine.Engine(0)
    e.build_blocks(dk.ptr.value, sk.offsets)
    T = e.num_tiles
    cap = int(e.edge_bound(0, T)) + 1
    de = engine.DeviceBuffer(cap * 16)
    cnt = e.join(0, T, de.ptr.value, cap)
    ev = np.sort(de.to_numpy(engine.EDGE_DTYPE, cnt), order=["source_1", "source_2"])
    got = int(ev["shared"].sum())
    st = e.stats()
    if got != want:
        bad += 1
        cnt2 = e.join(0, T, de.ptr.value, cap)
        ev2 = de.to_numpy(engine.EDGE_DTYPE, cnt2)
        print("  second join of the same build: edges", cnt2, "sum", int(ev2["shared"].sum()), flush=True)
        print(it, "MISMATCH edges", cnt, "sum", got, "want", want, {k: st[k] for k in ("n_active_tiles", "n_tiles", "sort_bits")}, flush=True)
        if ref is not None:
            kr = ref["source_1"].astype(np.int64) * n + ref["source_2"]
            kg = ev["source_1"].astype(np.int64) * n + ev["source_2"]
            miss = ref[~np.isin(kr, kg)]
            print("  missing", len(miss), "first", miss[:12].tolist(), flush=True)
            s = np.unique(np.concatenate([miss["source_1"], miss["source_2"]]))
            print("  sources involved", len(s), s[:40].tolist(), flush=True)
            both = ev[np.isin(kg, kr)]
            rr = ref[np.isin(kr, kg)]
            print("  common pairs with different counts", int((both["shared"] != rr["shared"]).sum()), flush=True)
    elif ref is None:
        ref = ev
    del e, de
print("iterations", iters, "mismatches", bad, flush=True)
