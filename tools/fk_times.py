#!/usr/bin/env python3
"""Per-stage time of k_fkeys summed over its workgroups (timing build: `make -C kspider_amd/csrc fktime`):
    python tools/fk_times.py [C2]
Clock: s_memtime (shader clock); prints every stage's share and its time per workgroup-batch."""
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
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    e.build_blocks(dk.ptr.value, sk.offsets)
L.ksp_debug_fktime(buf, 1)
reps = 5
for _ in range(reps):
    e.build_blocks(dk.ptr.value, sk.offsets)
L.ksp_debug_fktime(buf, 1)
t = np.array(list(buf), dtype=np.float64)[:9] / reps
names = ["batch tables + barrier", "A: tags + key starts loaded", "gather new indices + stores + barrier", "B: head prefix (wave 0)", "zero masks + barrier",
         "C: mask ORs + barrier", "D: key per thread + barrier", "several-block keys by wave + barrier", "tail: hist / work out"]
tot = t.sum()
st = e.stats()
print(f"{cfg}: stage1_kind {st['stage1_kind']}, build {st['ms_build']:.3f} ms; k_fkeys thread-0 cycles per build {tot:.3e}")
for n, x in zip(names, t):
    print(f"  {n:42s} {100 * x / tot:5.1f} %   {x:.3e}")
