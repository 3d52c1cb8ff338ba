#!/usr/bin/env python3
"""One build + two joins of a config (for rocprofv3 kernel traces): python tools/run_once.py C4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth
sk = synth.generate(sys.argv[1])
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
e.build_blocks(dk.ptr.value, sk.offsets)
cap = int(min(e.edge_bound(0, e.num_tiles), 1 << 26)) + 1
de = engine.DeviceBuffer(cap * 16)
for _ in range(2):
    cnt = e.join(0, e.num_tiles, de.ptr.value, cap)
print(cnt, e.stats()["ms_join"])
