#!/usr/bin/env python3
"""Phase times of stage 1 + join on one config (default C2), averaged over a few builds:
    python tools/time_build.py [C2] [reps]         (environment knobs apply: KSP_PARTITION, KSP_DEBUG_BUCKET_MEAN, ...)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sk = synth.generate(cfg)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
e.build_blocks(dk.ptr.value, sk.offsets)
cap = int(min(e.edge_bound(0, e.num_tiles), 1 << 26)) + 1
de = engine.DeviceBuffer(cap * 16)
for _ in range(2):
    e.build_blocks(dk.ptr.value, sk.offsets)
    e.join(0, e.num_tiles, de.ptr.value, cap)
e.set_profiling(True)
acc, order, tb, tj = {}, [], 0.0, 0.0
for _ in range(reps):
    e.build_blocks(dk.ptr.value, sk.offsets)
    cnt = e.join(0, e.num_tiles, de.ptr.value, cap)
    st = e.stats()
    tb += st["ms_build"]; tj += st["ms_join"]
    for name, ms in e.phase_times():
        if name not in acc:
            order.append(name)
        acc[name] = acc.get(name, 0.0) + ms
out = {"config": cfg, "env": {k: v for k, v in os.environ.items() if k.startswith("KSP_")}, "build_ms": tb / reps,
       "join_ms": tj / reps, "edges": cnt, "partition_kind": st["partition_kind"], "partition_fallback": st["partition_fallback"], "matches": st["n_match_records"], "wgs": st["n_join_workgroups"], "active": st["n_active_tiles"], "kept": st["n_kept_entries"], "keys": st["n_kept_keys"], "words": st["n_block_keys"],
       "phases": {k: round(acc[k] / reps, 4) for k in order}}
print(json.dumps(out))
