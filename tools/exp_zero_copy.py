#!/usr/bin/env python3
"""Experiment: the join writing its edges straight into pinned host memory (zero-copy stores over PCIe, overlapped
with the join) against edges in HBM + a D2H copy afterwards.  One timed step per variant and config."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth  # noqa: E402

dev = torch.device("cuda", 0)
for cfg in sys.argv[1:] or ["C2", "C5", "C3"]:
    sk = synth.generate(cfg)
    keys_d = torch.from_numpy(sk.keys.view(np.int64)).to(dev)
    eng = engine.Engine(0)
    stream = torch.cuda.current_stream(dev)
    eng.build_blocks(keys_d.data_ptr(), sk.offsets, stream=stream.cuda_stream)
    T = eng.num_tiles
    need = int(min(eng.edge_bound(0, T), 1 << 27)) + 1
    need += need // 8
    edges_d = torch.empty((need, 16), dtype=torch.uint8, device=dev)
    edges_h = torch.empty((need, 16), dtype=torch.uint8).pin_memory()
    for variant in ("copy", "zero-copy", "copy", "zero-copy"):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        eng.build_blocks(keys_d.data_ptr(), sk.offsets, stream=stream.cuda_stream)
        T = eng.num_tiles
        if variant == "copy":
            cnt = eng.join(0, T, edges_d.data_ptr(), need, stream=stream.cuda_stream)
            edges_h[:cnt].copy_(edges_d[:cnt], non_blocking=True)
        else:
            cnt = eng.join(0, T, edges_h.data_ptr(), need, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t
        st = eng.stats()
        ev = edges_h[:cnt].numpy().view(engine.EDGE_DTYPE).reshape(-1)
        print(json.dumps({"config": cfg, "variant": variant, "step_ms": round(1e3 * wall, 3), "build_ms": round(st["ms_build"], 3),
                          "join_ms": round(st["ms_join"], 3), "edges": int(cnt), "sum_shared": int(ev["shared"].sum())}), flush=True)
    eng.close()
    del keys_d, edges_d, edges_h
    torch.cuda.empty_cache()
