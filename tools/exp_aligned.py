#!/usr/bin/env python3
"""Experiment: what would block boundaries that respect the clusters be worth?  C2's sources ordered by generator
cluster, (a) packed contiguously — what the label reordering achieves today — and (b) with empty dummy sources
inserted so that no cluster of <= 128 sources straddles a 128-source block.  Both run with KSP_REORDER=0."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KSP_REORDER"] = "0"
from kspider_amd import engine, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sk = synth.generate(cfg)
n = sk.n_sources
order = np.argsort(sk.cluster, kind="stable")
csz = np.bincount(sk.cluster)


def layout(aligned):
    slots = []   # source id or -1 (dummy)
    pos = 0
    for c, k in enumerate(csz):
        if k == 0:
            continue
        if aligned and k <= 128 and (pos % 128) + k > 128:
            pad = 128 - pos % 128
            slots += [-1] * pad
            pos += pad
        slots += [0] * int(k)
        pos += int(k)
    slots = np.asarray(slots)
    slots[slots == 0] = order
    return slots


for aligned in (False, True):
    sl = layout(aligned)
    sizes = np.where(sl >= 0, sk.sizes[np.maximum(sl, 0)], 0)
    off = np.zeros(len(sl) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    keys = np.concatenate([sk.run(s) for s in sl if s >= 0])
    dk = engine.DeviceBuffer.from_numpy(keys)
    e = engine.Engine(0)
    e.build_blocks(dk.ptr.value, off)
    cap = int(min(e.edge_bound(0, e.num_tiles), 1 << 26)) + 1
    de = engine.DeviceBuffer(cap * 16)
    tb = tj = 0.0
    for _ in range(5):
        e.build_blocks(dk.ptr.value, off)
        cnt = e.join(0, e.num_tiles, de.ptr.value, cap)
        st = e.stats()
        tb += st["ms_build"] / 5
        tj += st["ms_join"] / 5
    print(json.dumps({"aligned": aligned, "slots": len(sl), "blocks": (len(sl) + 127) // 128, "build_ms": round(tb, 3), "join_ms": round(tj, 3),
                      "edges": int(cnt), "active": st["n_active_tiles"], "wgs": st["n_join_workgroups"], "words": st["n_block_keys"]}), flush=True)
    de.free(); dk.free(); e.close()
