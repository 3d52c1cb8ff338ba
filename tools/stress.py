"""Repeat build + join many times on the bench workload and check that every run gives the same edges
(catches races that a single parity run can miss).   python tools/stress.py [n_sources] [iterations]"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sk = synth.generate("C2", n_sources=N)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
e.build_blocks(dk.ptr.value, sk.offsets)
cap = int(e.edge_bound(0, e.num_tiles)) + 1
de = engine.DeviceBuffer(cap * 16)
ref = None
for it in range(IT):
    e.build_blocks(dk.ptr.value, sk.offsets)
    cuts = e.balanced_cuts(1 + it % 4)
    parts = []
    for r in range(len(cuts) - 1):
        cnt = e.join(cuts[r], cuts[r + 1], de.ptr.value, cap)
        parts.append(de.to_numpy(engine.EDGE_DTYPE, cnt))
    ev = np.sort(np.concatenate(parts), order=["source_1", "source_2"])
    h = hashlib.sha1(ev.tobytes()).hexdigest()
    if ref is None:
        ref = h
        print("edges", len(ev), "sha1", h, flush=True)
    elif h != ref:
        print("MISMATCH at iteration", it, len(ev), h)
        sys.exit(1)
print("stable over", IT, "iterations")
