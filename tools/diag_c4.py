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
    if got != want and bad >= 3:
        bad += 1
        continue
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
            import ctypes
            ni = np.zeros(n, dtype=np.uint32)
            engine.lib().ksp_engine_source_order(ctypes.c_void_p(e._h.value if hasattr(e._h, "value") else e._h), ni.ctypes.data_as(ctypes.c_void_p))
            bi, bj = ni[miss["source_1"]] // 128, ni[miss["source_2"]] // 128
            tl = np.stack([np.minimum(bi, bj), np.maximum(bi, bj)], axis=1)
            ut, uc = np.unique(tl, axis=0, return_counts=True)
            print("  tiles of the missing pairs (I, J, pairs):", [(int(a), int(b), int(c)) for (a, b), c in zip(ut, uc)][:20], flush=True)
            sizes = np.diff(sk.offsets)
            for (a, b) in ut[:4]:
                for blk in {int(a), int(b)}:
                    members = np.flatnonzero(ni // 128 == blk)
                    print("   block", blk, "sources", len(members), "max size", int(sizes[members].max()), "sum size", int(sizes[members].sum()), flush=True)
                # pairs of that tile present in the reference but not missing?
                rbi, rbj = ni[ref["source_1"]] // 128, ni[ref["source_2"]] // 128
                in_tile = (np.minimum(rbi, rbj) == a) & (np.maximum(rbi, rbj) == b)
                print("   tile", int(a), int(b), "pairs in the full result", int(in_tile.sum()), flush=True)
                nbk = (n + 127) // 128
                t = int(a) * nbk - int(a) * (int(a) - 1) // 2 + (int(b) - int(a))
                c1 = e.join(t, t + 1, de.ptr.value, cap)
                print("   tile id", t, "edge_bound", int(e.edge_bound(t, t + 1)), "join of that tile alone ->", c1, "active in that call", e.stats()["last_active_tiles"], flush=True)
    elif ref is None:
        ref = ev
    del e, de
print("iterations", iters, "mismatches", bad, flush=True)
