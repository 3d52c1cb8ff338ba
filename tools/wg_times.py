#!/usr/bin/env python3
"""Per-workgroup timeline of k_join on one config (timing build: `make -C kspider_amd/csrc wgtime`):
    python tools/wg_times.py [C2]
Prints the span of the launch, how long the workgroups of each kind (diagonal / off-diagonal, split or not) run,
when they start, and the longest ones.  Clock: s_memrealtime, 100 MHz (10 ns ticks)."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KSPIDER_AMD_LIB", os.path.join(ROOT, "kspider_amd", "lib", "libkspider_amd_wgtime.so"))
out = os.path.join(tempfile.gettempdir(), "ksp_wgt.bin")
os.environ["KSP_WGTIME_FILE"] = out
from kspider_amd import engine, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sk = synth.generate(cfg)
dk = engine.DeviceBuffer.from_numpy(sk.keys)
e = engine.Engine(0)
e.build_blocks(dk.ptr.value, sk.offsets)
cap = int(min(e.edge_bound(0, e.num_tiles), 1 << 26)) + 1
de = engine.DeviceBuffer(cap * 16)
for _ in range(3):
    e.join(0, e.num_tiles, de.ptr.value, cap)
st = e.stats()
r = np.fromfile(out, dtype=np.uint64).reshape(-1, 16)
r = r[r[:, 1] != 0]
t0 = r[:, 0].astype(np.int64)
t1 = r[:, 1].astype(np.int64)
I = (r[:, 2] >> np.uint64(32)).astype(np.int64)
J = (r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
sp = (r[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
base = t0.min()
us = lambda x: np.asarray(x, dtype=np.float64) / 100.0
dur = us(t1 - t0)
print(json.dumps({"config": cfg, "join_ms": st["ms_join"], "workgroups": int(len(r)), "span_us": float(us(t1.max() - base)),
                  "last_start_us": float(us(t0.max() - base)), "sum_wg_us": float(dur.sum())}))
for name, sel in (("diag sp=1", (I == J) & (sp == 1)), ("diag split", (I == J) & (sp > 1)), ("off sp=1", (I != J) & (sp == 1)),
                  ("off split", (I != J) & (sp > 1))):
    if sel.any():
        d = dur[sel]
        print(f"{name:11s} n={sel.sum():6d}  dur us: min {d.min():8.1f} med {np.median(d):8.1f} p90 {np.percentile(d, 90):8.1f} max {d.max():8.1f}"
              f"  sum {d.sum():10.0f}   start us: med {np.median(us(t0[sel] - base)):7.1f} max {us(t0[sel] - base).max():7.1f}"
              f"   end us: max {us(t1[sel] - base).max():7.1f}")
ph = us(r[:, 4:15].astype(np.int64))
names = {"diag": ["setup", "load+transpose", "barrier", "popcount", "barrier", "partial atomics", "edges (unsplit)", "", "", "fence+done", "last share: edges"],
         "off": ["setup", "search step", "barrier", "masks+transpose", "barrier", "popcount", "barrier", "loop exit", "edges / partial atomics", "fence+done", "last share: edges"]}
for kind, sel in (("diag", I == J), ("off", I != J)):
    if sel.any():
        print(f"{kind}: mean us per workgroup by phase (thread 0): " + ", ".join(f"{n} {ph[sel, k].mean():.1f}" for k, n in enumerate(names[kind]) if n))
order = np.argsort(-dur)[:12]
for i in order:
    print(f"  wg {i:6d} tile ({I[i]},{J[i]}) sp {sp[i]:3d}: start {us(t0[i] - base):7.1f} dur {dur[i]:7.1f} us  phases " + " ".join(f"{x:.0f}" for x in ph[i]))

# ---- what the host's cost model sees against what the shares take (WG_FIT=1): per tile, time x shares vs list lengths ----
if os.environ.get("WG_FIT"):
    import ctypes
    L = engine.lib()
    nbk = int(e.stats()["n_blocks"])
    arr = np.zeros(nbk + 1, dtype=np.uint32)
    L.ksp_engine_block_key_counts.restype = ctypes.c_int
    L.ksp_engine_block_key_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.ksp_engine_block_key_counts(e._h, arr.ctypes.data_as(ctypes.c_void_p))
    off = arr.astype(np.int64)
    w = np.diff(off)
    rows = {}
    for i in range(len(r)):
        rows.setdefault((int(I[i]), int(J[i])), []).append(dur[i])
    xs_d, ys_d, xs_o, ys_o = [], [], [], []
    for (a, b), ds in rows.items():
        tot, mx = float(np.sum(ds)), float(np.max(ds))
        if a == b:
            xs_d.append(w[a]); ys_d.append(tot)
        else:
            xs_o.append(w[a] + w[b]); ys_o.append(tot)
    # matches per off-diagonal tile (keys both blocks hold), from the sketches and the engine's source order — what the
    # host's cost model does not see
    order = e.source_order(sk.n_sources).astype(np.int64)
    blk_of = order // 128
    src = np.repeat(np.arange(sk.n_sources, dtype=np.int64), np.diff(sk.offsets.astype(np.int64)))
    comb = np.unique((sk.keys.astype(np.uint64) << np.uint64(7)) | blk_of[src].astype(np.uint64))   # (key, block) pairs; C2: 100 blocks, keys < 2^55
    kk, bb = comb >> np.uint64(7), (comb & np.uint64(127)).astype(np.int64)
    match = {}
    nbk2 = int(bb.max()) + 1
    cnt2 = np.zeros(nbk2 * nbk2, dtype=np.int64)
    for d in range(1, 64):
        same = kk[d:] == kk[:-d]
        if not same.any():
            break
        cnt2 += np.bincount(bb[:-d][same] * nbk2 + bb[d:][same], minlength=nbk2 * nbk2)
    xs2, ys2 = [], []
    for (a, b), ds in rows.items():
        if a != b:
            xs2.append((w[a] + w[b], cnt2[a * nbk2 + b])); ys2.append(float(np.sum(ds)))
    X = np.asarray(xs2, dtype=np.float64); Y = np.asarray(ys2)
    A2 = np.vstack([X[:, 0], X[:, 1], np.ones(len(X))]).T
    co = np.linalg.lstsq(A2, Y, rcond=None)[0]
    res2 = Y - A2 @ co
    print(f"off: us = {co[0] * 1000:.3f} per 1000 words + {co[1] * 1000:.3f} per 1000 matches + {co[2]:.1f}; residual sd {res2.std():.1f} us (max {res2.max():.1f}, min {res2.min():.1f}); matches per tile min {X[:, 1].min():.0f} median {np.median(X[:, 1]):.0f} max {X[:, 1].max():.0f}")
    for name, xs, ys in (("diag: sum of share us vs words", xs_d, ys_d), ("off: sum of share us vs words(I) + words(J)", xs_o, ys_o)):
        xs, ys = np.asarray(xs, dtype=np.float64), np.asarray(ys, dtype=np.float64)
        A = np.vstack([xs, np.ones_like(xs)]).T
        k, c0 = np.linalg.lstsq(A, ys, rcond=None)[0]
        res = ys - (k * xs + c0)
        print(f"{name}: n={len(xs)} fit us = {k * 1000:.3f} per 1000 words + {c0:.1f}; residual sd {res.std():.1f} us, max {res.max():.1f}, min {res.min():.1f}; words min {xs.min():.0f} max {xs.max():.0f}")
    # the worst under-estimated tiles
    pred = {}
    for (a, b), ds in rows.items():
        pred[(a, b)] = (float(np.sum(ds)), len(ds), int(w[a]), int(w[b]))
    worst = sorted(pred.items(), key=lambda kv: -kv[1][0] / kv[1][1])[:15]
    for (a, b), (tot, nsh, wa, wb) in worst:
        print(f"  tile ({a},{b}) shares {nsh} sum {tot:.0f} us = {tot / nsh:.0f} per share; words {wa} / {wb}")
