#!/usr/bin/env python3
"""Join time of one config under a few environment variations (each in a fresh engine)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth
cfg = sys.argv[1]
sk = synth.generate(cfg)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
variants = [{}, {"KSP_JOIN": "search"}, {"KSP_JOIN": "matches"}, {"KSP_JOIN": "matches", "KSP_COLLECT": "0"}, {"KSP_JOIN": "matches", "KSP_COLLECT": "1"}]
for env in variants:
    for k in ("KSP_COLLECT", "KSP_NO_SCHED", "KSP_DEBUG_SHARES", "KSP_JOIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = engine.Engine(0)
    e.build_blocks(dk.ptr.value, sk.offsets)
    cap = int(min(e.edge_bound(0, e.num_tiles), 1 << 26)) + 1
    de = engine.DeviceBuffer(cap * 16)
    ms = []
    for _ in range(3):
        cnt = e.join(0, e.num_tiles, de.ptr.value, cap)
        ms.append(e.stats()["ms_join"])
    st = e.stats()
    print(json.dumps({"config": cfg, "env": env, "join_ms": [round(x, 3) for x in ms], "edges": cnt, "active": st["n_active_tiles"],
                      "tiles": st["n_tiles"], "block_keys": st["n_block_keys"], "build_ms": round(st["ms_build"], 2)}), flush=True)
    de.free(); e.close()
