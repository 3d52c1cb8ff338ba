#!/usr/bin/env python3
"""Per-stage time of k_key_groups / k_bucket_group / k_seg_scatter summed over their workgroups (timing build:
`make -C kspider_amd/csrc fktime`):    python tools/st_times.py [C2]
Clock: s_memtime (shader clock), thread 0 of every workgroup; prints every stage's share of its kernel."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KSPIDER_AMD_LIB", os.path.join(ROOT, "kspider_amd", "lib", "libkspider_amd_fktime.so"))
from kspider_amd import engine, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sk = synth.generate(cfg)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
L = engine.lib()
buf = (ctypes.c_ulonglong * 64)()
for _ in range(3):
    e.build_blocks(dk.ptr.value, sk.offsets)
probe = 2 if os.environ.get("ST_PROBE") else 1
L.ksp_debug_sttime(buf, probe)
reps = 5
for _ in range(reps):
    e.build_blocks(dk.ptr.value, sk.offsets)
L.ksp_debug_sttime(buf, 1)
t = np.array(list(buf), dtype=np.float64) / reps
print(f"{cfg}: build {e.stats()['ms_build']:.3f} ms")
KERNELS = [("k_key_groups", 16, ["bounds: crank / first loads", "staging: tags -> new indices -> LDS + barrier", "thread-per-key walks",
                                 "barrier", "wave-per-key keys"]),
           ("k_bucket_group (per bucket pass)", 32, ["table init + barrier", "inserts + next keys issued + barrier", "slot scan + barrier",
                                                      "offsets written + barrier", "placement + records stored + end barrier",
                                                      "(probe) wait for the next keys before the placement", "(probe) placement + record stores issued",
                                                      "(probe) wait for the record stores"]),
           ("k_seg_scatter", 48, ["segment table + scan + barrier", "tile size + barrier", "segment search + key loads + bucket counts + barrier: the barrier",
                                  "bucket starts / reservations + barrier", "LDS scatter + barrier", "  (of stage 3) segment of every entry", "  (of stage 3) key loads issued and waited for",
                                  "  (of stage 3) bucket + LDS counts (the rest: barrier)", "runs written out"])]
for name, base, stages in KERNELS:
    n = len(stages)
    x = t[base:base + n]
    tot = x.sum()
    if tot == 0:
        continue
    print(f"{name}: thread-0 cycles per build {tot:.3e}")
    for s, v in zip(stages, x):
        if s:
            print(f"  {s:58s} {100 * v / tot:5.1f} %   {v:.3e}")
