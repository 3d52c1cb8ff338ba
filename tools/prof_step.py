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
# (the generator runs OUTSIDE the profiler: `prof_step.py C4 dump FILE` writes the set, `... load FILE` reads it back —
#  under rocprofv3 the numpy-heavy generation of C4 took minutes instead of seconds)
mode = sys.argv[3] if len(sys.argv) > 3 else ""
path = sys.argv[4] if len(sys.argv) > 4 else ""
if mode == "load":
    z = np.load(path)
    sk = synth.SketchSet(z["keys"], z["offsets"], z["cluster"], cfg)
else:
    kw = {}
    if os.environ.get("PROF_SOURCES"):   # (a variant of the configuration: PROF_SOURCES=50000 PROF_MEAN=10000)
        kw["n_sources"] = int(os.environ["PROF_SOURCES"])
    if os.environ.get("PROF_MEAN"):
        kw["mean_size"] = int(os.environ["PROF_MEAN"])
    sk = synth.generate(cfg, **kw)
    if mode == "dump":
        np.savez(path, keys=sk.keys, offsets=sk.offsets, cluster=sk.cluster)
        print(f"[prof_step {cfg}] dumped to {path}", flush=True)
        sys.exit(0)
print(f"[prof_step {cfg}] {sk.n_sources} sources, {int(sk.offsets[-1])} hashes", flush=True)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
de = None
out = {}
builds, joins = [], []
for s in range(steps):
    if os.environ.get("PROF_PHASES") and s == steps - 1:   # (the last step with an event per phase: where the build's time goes)
        e.set_profiling(True)
    e.build_blocks(dk.ptr.value, sk.offsets)
    if os.environ.get("PROF_PHASES") and s == steps - 1:
        print(f"[prof_step {cfg}] phases: " + ", ".join(f"{n} {ms:.3f}" for n, ms in e.phase_times()), flush=True)
    T = e.num_tiles
    if de is None:
        cap = int(min(e.edge_bound(0, T), 1 << 27)) + 1
        de = engine.DeviceBuffer(cap * 16)
    n = e.join(0, T, de.ptr.value, cap)
    st = e.stats()
    builds.append(st["ms_build"]); joins.append(st["ms_join"])
    out = {"config": cfg, "step": s, "build_ms": st["ms_build"], "join_ms": st["ms_join"],
           "build_ms_min": min(builds[1:] or builds), "join_ms_min": min(joins[1:] or joins), "edges": int(n),
           "partition_kind": st["partition_kind"], "stage1_kind": st["stage1_kind"], "kept_entries": int(st["n_kept_entries"]),
           "kept_keys": int(st["n_kept_keys"]), "list_words": int(st["n_block_keys"]), "active_tiles": int(st["n_active_tiles"])}
    print(json.dumps(out), flush=True)
stop.set()
