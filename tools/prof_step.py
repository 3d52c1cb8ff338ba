#!/usr/bin/env python3
"""One configuration, a few steps (build + join into a device buffer), for rocprofv3 runs:
    python tools/prof_step.py C3 [steps]
Prints a heartbeat every 20 s (a PMC pass over a large configuration is silent for minutes otherwise) and one JSON
line with the event times of the last step."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stop = threading.Event()


def beat():
    t0 = time.time()
    while not stop.wait(20):
        print(f"[prof_step {cfg}] {time.time() - t0:.0f} s", flush=True)


threading.Thread(target=beat, daemon=True).start()
sk = synth.generate(cfg)
print(f"[prof_step {cfg}] {sk.n_sources} sources, {int(sk.offsets[-1])} hashes", flush=True)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
de = None
out = {}
for s in range(steps):
    e.build_blocks(dk.ptr.value, sk.offsets)
    T = e.num_tiles
    if de is None:
        cap = int(min(e.edge_bound(0, T), 1 << 27)) + 1
        de = engine.DeviceBuffer(cap * 16)
    n = e.join(0, T, de.ptr.value, cap)
    st = e.stats()
    out = {"config": cfg, "step": s, "build_ms": st["ms_build"], "join_ms": st["ms_join"], "edges": int(n),
           "partition_kind": st["partition_kind"], "stage1_kind": st["stage1_kind"], "kept_entries": int(st["n_kept_entries"]),
           "kept_keys": int(st["n_kept_keys"]), "list_words": int(st["n_block_keys"]), "active_tiles": int(st["n_active_tiles"])}
    print(json.dumps(out), flush=True)
stop.set()
