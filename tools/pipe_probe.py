#!/usr/bin/env python3
"""Step time of C2 under a few host-side orderings (same box): join synchronous or launched + collected one step
later, edges copied to pinned memory on a copy stream or not at all."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kspider_amd import engine, synth  # noqa: E402

sk = synth.generate("C2")
dev = torch.device("cuda", 0)
keys_d = torch.from_numpy(sk.keys.view(np.int64)).to(dev)
stream = torch.cuda.current_stream(dev)
copy_stream = torch.cuda.Stream(device=dev)
eng = engine.Engine(0)
cap = 1 << 21
edges_d = [torch.empty((cap, 16), dtype=torch.uint8, device=dev) for _ in range(2)]
edges_h = [torch.empty((cap, 16), dtype=torch.uint8).pin_memory() for _ in range(2)]


def run(mode, copy, steps=30):
    copied = [None, None]
    pend = None
    tb = 0.0

    def hand(job):
        buf, cnt, ready = job
        if not copy:
            return
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            edges_h[buf][:cnt].copy_(edges_d[buf][:cnt], non_blocking=True)
            copied[buf] = torch.cuda.Event()
            copied[buf].record(copy_stream)

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        buf = k & 1
        t = time.perf_counter()
        eng.build_blocks(keys_d.data_ptr(), sk.offsets, stream=stream.cuda_stream)
        tb += time.perf_counter() - t
        T = eng.num_tiles
        if copied[buf] is not None:
            copied[buf].synchronize()
        if mode == "sync":
            cnt = eng.join(0, T, edges_d[buf].data_ptr(), cap, stream=stream.cuda_stream)
            ready = torch.cuda.Event()
            ready.record(stream)
            hand((buf, cnt, ready))
        else:
            prev = None
            if pend is not None:
                prev = (pend[0], eng.join_wait(), pend[1])
            eng.join_launch(0, T, edges_d[buf].data_ptr(), cap, stream=stream.cuda_stream)
            ready = torch.cuda.Event()
            ready.record(stream)
            pend = (buf, ready)
            if prev is not None:
                hand(prev)
    if mode != "sync" and pend is not None:
        hand((pend[0], eng.join_wait(), pend[1]))
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    return {"mode": mode, "copy": copy, "ms_per_step": round(1e3 * el / steps, 4), "build_call_ms": round(1e3 * tb / steps, 4)}


for _ in range(2):
    for mode in ("sync", "pipe"):
        for copy in (False, True):
            print(json.dumps(run(mode, copy)), flush=True)
