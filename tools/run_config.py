#!/usr/bin/env python3
"""Run one BASELINE.json configuration (C1..C5 synthetic shape) through the engine on one GPU and check
size-independent properties: sum of all shared counts == sum_k C(holders_k, 2) (independent inverted-index
count on the host), ordering, bounds, and sampled pairs against direct set intersection.

    python tools/run_config.py C3 [n_sources]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from kspider_amd import engine, synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n_over = int(sys.argv[2]) if len(sys.argv) > 2 else None
    t = time.time()
    sk = synth.generate(cfg, n_sources=n_over)
    t_gen = time.time() - t
    n = sk.n_sources
    print(f"{cfg}: {n} sources, {int(sk.offsets[-1])} hashes, sizes min/mean/max {sk.sizes.min()}/{sk.sizes.mean():.0f}/"
          f"{sk.sizes.max()} (generated in {t_gen:.1f} s)", flush=True)
    dev = torch.device("cuda", 0)
    keys_d = torch.from_numpy(sk.keys.view(np.int64)).to(dev)
    eng = engine.Engine(0)
    for _ in range(2):
        t = time.time()
        eng.build_blocks(keys_d.data_ptr(), sk.offsets)
        torch.cuda.synchronize()
        t_build = time.time() - t
    st = eng.stats()
    T = eng.num_tiles
    cap = 1 << 26
    edges_d = torch.empty((cap, 16), dtype=torch.uint8, device=dev)
    # tile ranges as large as the edge buffer allows: halve the range on overflow
    total_shared, n_edges, ms_join, chunks, launches = 0, 0, 0.0, [], 0
    t = time.time()
    t0, step = 0, T
    while t0 < T:
        t1 = min(T, t0 + step)
        launches += 1
        try:
            cnt = eng.join(t0, t1, edges_d.data_ptr(), cap)
        except engine.KspError as ex:
            if ex.code != engine.KSP_E_OVERFLOW or t1 - t0 <= 1:
                raise
            ms_join += eng.stats()["ms_join"]
            step = (t1 - t0) // 2
            continue
        ms_join += eng.stats()["ms_join"]
        if cnt:
            ev = edges_d[:cnt].cpu().numpy().view(engine.EDGE_DTYPE).reshape(-1)
            assert (ev["source_1"] < ev["source_2"]).all() and int(ev["source_2"].max()) < n
            assert (ev["shared"] > 0).all()
            assert (ev["shared"] <= np.minimum(sk.sizes[ev["source_1"]], sk.sizes[ev["source_2"]])).all()
            total_shared += int(ev["shared"].sum())
            n_edges += cnt
            if len(chunks) < 4:
                chunks.append(ev[:: max(1, cnt // 200)][:200].copy())
        t0 = t1
    t_join_wall = time.time() - t
    pairs = n * (n - 1) // 2
    # independent checksum: every key held by m sources contributes C(m, 2)
    t = time.time()
    _, counts = np.unique(sk.keys, return_counts=True)
    want = int((counts.astype(np.int64) * (counts - 1) // 2).sum())
    t_chk = time.time() - t
    ok = want == total_shared
    for ev in chunks:
        for a, b, s in zip(ev["source_1"][:50], ev["source_2"][:50], ev["shared"][:50]):
            assert np.intersect1d(sk.run(int(a)), sk.run(int(b)), assume_unique=True).size == int(s)
    out = dict(config=cfg, n_sources=n, hashes=int(sk.offsets[-1]), pairs=pairs, nonzero_pairs=n_edges,
               block_keys=int(st["n_block_keys"]), tiles=int(T), launches=launches, build_ms=1e3 * t_build, join_kernel_ms=ms_join,
               join_wall_ms=1e3 * t_join_wall, pairs_per_s_kernels=pairs / ((1e3 * t_build + ms_join) / 1e3),
               checksum_ok=bool(ok), checksum_secs=t_chk)
    print(json.dumps(out), flush=True)
    if not ok:
        raise SystemExit(f"checksum mismatch: {total_shared} != {want}")


if __name__ == "__main__":
    main()
