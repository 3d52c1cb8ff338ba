#!/usr/bin/env python3
"""bench.py — pairwise containment throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic sketches that already
sit in HBM as sorted uint64 runs:
    stage 1  build_blocks  (prune singletons, rank-encode, merge the runs of every 128-source block;
                            N > 1: per-rank hash-range slices + all-gather + assemble)
    stage 2  join          (LDS-tiled intersection of this rank's tile range -> edges in HBM)
    gather   RCCL point-to-point gather of the edge lists to rank 0 (N > 1 only)
    D2H      rank 0 copies the edges to pinned host memory (the hand-over to the TSV writer)

Workload: BASELINE.json configs[1] ("10k sourmash signatures, scaled=1000, k=31" ->
synthetic C2: 10 000 sketches, n ~ N(5000, 1500), hashes < 2^64/1000).  N > 1 is WEAK
scaling: the source count grows as 10 000 * sqrt(N) so that every GPU keeps the pair count
of the 1-GPU job; every rank holds the full sketch set; stage 1 is sharded by hash range
(rank r builds the block-list slices of its 1/N share of the keys; the slices are exchanged
with one RCCL all-gather), tiles are sharded, and the edges are gathered to rank 0.

value = whole-job source pairs per second = [S(S-1)/2] * K / t, t = max over ranks of the
wall time of K steps bracketed by barrier + torch.cuda.synchronize().
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(sizes: np.ndarray, n_sources: int, t0: int, t1: int) -> int:
    """SURVEY §8(d) per-pair figure B(a,b) = 8 (n_a + n_b) + 4, summed over the source pairs
    of tiles [t0, t1) of the row-major block-pair upper triangle (128-source blocks)."""
    tb = 128
    nb = (n_sources + tb - 1) // tb
    cnt = np.array([min(tb, n_sources - b * tb) for b in range(nb)], dtype=np.int64)
    tot = np.add.reduceat(sizes.astype(np.int64), np.arange(0, n_sources, tb))
    sq = None
    total = 0
    t = 0
    for i in range(nb):
        row = nb - i
        lo, hi = max(t0, t), min(t1, t + row)
        if lo < hi:
            js = np.arange(i + (lo - t), i + (hi - t))
            for j in js:
                if j == i:
                    pairs = cnt[i] * (cnt[i] - 1) // 2
                    total += 8 * (cnt[i] - 1) * tot[i] + 4 * pairs
                else:
                    total += 8 * (cnt[j] * tot[i] + cnt[i] * tot[j]) + 4 * cnt[i] * cnt[j]
        t += row
        if t >= t1:
            break
    return int(total)


def cpu_baseline(sk, sample_sources: int) -> dict:
    """Reference algorithm (oracle restatement of src/pairwise.cpp:194-237) on the host cores,
    on a bounded sample of the same workload.  Reported next to the GPU number, not a target."""
    import oracle
    sub = sk.subset(sample_sources)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    t0 = time.perf_counter()
    co, src, w = oracle.build_colors(sub.keys, sub.offsets)
    t_index = time.perf_counter() - t0
    secs, n_edges, n_updates, _ = oracle.accumulate_mem(co, src, w, cores, want_edges=False)
    n = sub.n_sources
    return {
        "value": (n * (n - 1) // 2) / secs, "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": f"first {n} of the workload's sources ({int(sub.offsets[-1])} hashes, {len(w)} colours, "
                  f"{n_updates} map updates, {n_edges} non-zero pairs); accumulate region only "
                  f"(src/pairwise.cpp:200-239 equivalent) {secs:.2f} s; colour index build {t_index:.1f} s not counted",
        "secs": secs,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--n-sources", type=int, default=0, help="override the source count (debug)")
    ap.add_argument("--cpu-sample", type=int, default=10000, help="sources in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from kspider_amd import dist as kdist
    from kspider_amd import engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # test hooks (tests/test_bench_gpu.py): several ranks may share one card, exchanging through gloo
    share_gpu = os.environ.get("KSP_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("KSP_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    base_n = synth.CONFIGS[args.config]["n"]
    n_sources = args.n_sources or int(round(base_n * math.sqrt(world)))
    sk = synth.generate(args.config, n_sources=n_sources)   # identical on every rank (deterministic)
    n = sk.n_sources
    total_pairs = n * (n - 1) // 2

    keys_d = torch.from_numpy(sk.keys.view(np.int64)).to(dev)
    key_bits = max(1, int(sk.keys.max()).bit_length()) if sk.keys.size else 1   # the sketcher knows its hash range
    stream = torch.cuda.current_stream(dev)
    eng = engine.Engine(local_rank)
    sharded = world > 1 and os.environ.get("KSP_BENCH_REPLICATED_BUILD") != "1"
    # one untimed build fixes this rank's tile range (equal estimated work per GPU; the data and
    # therefore the cuts are the same in every step) and sizes the edge buffers
    if sharded:
        kdist.build_blocks_sharded(eng, keys_d.data_ptr(), sk.offsets, world, rank, dev, stream=stream.cuda_stream)
    else:
        eng.build_blocks(keys_d.data_ptr(), sk.offsets, key_bits=key_bits, stream=stream.cuda_stream)
    cuts = eng.balanced_cuts(world)
    t0, t1 = cuts[rank], cuts[rank + 1]
    cap = int(min(eng.edge_bound(t0, t1), 1 << 27)) + 1
    # double-buffered results: the D2H copy of step k runs on its own stream under stage 1 of step k + 1
    edges_dd = [torch.empty((cap, 16), dtype=torch.uint8, device=dev) for _ in range(2)]
    host_cap = int(min(eng.edge_bound(0, eng.num_tiles), total_pairs, 1 << 27)) + 1
    edges_hh = [torch.empty((host_cap, 16), dtype=torch.uint8).pin_memory() for _ in range(2)] if rank == 0 else None
    copy_stream = torch.cuda.Stream(device=dev)
    copied = [None, None]     # event: the D2H copy out of buffer i has finished
    in_flight = [None, None]  # keeps the gathered device tensor alive until its copy is done
    step_no = [0]

    stats = {"ms_join": 0.0, "ms_build": 0.0, "edges": 0, "stream_bytes": 0, "xchg_bytes": 0}

    def step(record: bool):
        if sharded:   # stage 1 sharded by hash range + all-gather of the block-list slices
            stats["xchg_bytes"] = kdist.build_blocks_sharded(eng, keys_d.data_ptr(), sk.offsets, world, rank, dev,
                                                             stream=stream.cuda_stream)
        else:
            eng.build_blocks(keys_d.data_ptr(), sk.offsets, key_bits=key_bits, stream=stream.cuda_stream)
        buf = step_no[0] & 1
        step_no[0] += 1
        if copied[buf] is not None:
            copied[buf].synchronize()       # the copy that last read this buffer pair (two steps ago)
        edges_d, edges_h = edges_dd[buf], (edges_hh[buf] if rank == 0 else None)
        cnt = eng.join(t0, t1, edges_d.data_ptr(), cap, stream=stream.cuda_stream)
        local = edges_d[:cnt]
        if world > 1 and backend != "nccl":
            local = local.cpu()     # gloo exchanges host tensors (test hook only)
        allv = kdist.gather_edges(local, dst=0) if world > 1 else local
        if rank == 0:
            ready = torch.cuda.Event()
            ready.record(stream)            # (the gather's kernels run on the compute stream)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready)
                edges_h[: allv.shape[0]].copy_(allv, non_blocking=True)
                copied[buf] = torch.cuda.Event()
                copied[buf].record(copy_stream)
            in_flight[buf] = allv
        if record:
            st = eng.stats()
            stats["ms_join"] += st["ms_join"]
            stats["ms_build"] += st["ms_build"]
            stats["ms_sort"] = stats.get("ms_sort", 0.0) + st["ms_sort"]
            stats["sort_entries"], stats["sort_bits"] = st["sort_entries"], st["sort_bits"]
            stats["stream_bytes"] = st["last_stream_bytes"]
            stats["edges"] = int(allv.shape[0]) if rank == 0 else cnt
            if rank == 0 and os.environ.get("KSP_BENCH_CHECKSUM") == "1":
                copy_stream.synchronize()
                ev = edges_h[: allv.shape[0]].numpy().view(engine.EDGE_DTYPE).reshape(-1)
                stats["checksum"] = int(ev["shared"].sum()) ^ (int(ev["source_1"].astype(np.int64).sum()) << 20) ^ int(
                    ev["source_2"].astype(np.int64).sum())
        return cnt

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    stats["checksum"] = 0

    for _ in range(args.warmup):
        step(False)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_join = stats["ms_join"] / max(1, args.steps)
    ms_build = stats["ms_build"] / max(1, args.steps)
    alg_bytes = algorithmic_bytes(sk.sizes, n, t0, t1)       # this rank's launch
    achieved = alg_bytes / (ms_join * 1e-3) / 1e9 if ms_join > 0 else 0.0
    stream_gbs = stats["stream_bytes"] / (ms_join * 1e-3) / 1e9 if ms_join > 0 else 0.0

    if rank == 0:
        traffic = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("k_join_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "sketch-pairs/sec (NxN containment)", "value": total_pairs * args.steps / elapsed,
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 keys -> u32 ranks, u32 counters", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {args.config} synthetic sourmash-like sketches "
                                   f"(scaled=1000 hash range), {n} sources, {int(sk.offsets[-1])} hashes; "
                                   f"weak scaling: sources = {base_n}*sqrt(n_gpus)",
                       "n_sources": n, "pairs": total_pairs, "nonzero_pairs": stats["edges"],
                       "checksum": stats["checksum"],
                       "tiles": int(eng.num_tiles), "active_tiles": int(eng.stats()["n_active_tiles"]),
                       "parallelism": (f"tile-range shard x{world}; stage 1 "
                                       + ("sharded by hash range + RCCL all-gather of the block-list slices "
                                          f"({stats['xchg_bytes'] / 1e6:.0f} MB received per rank)" if sharded
                                          else "replicated") + "; RCCL p2p gather of the edges to rank 0")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ksp::k_join", "ms_per_launch": ms_join,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes = SURVEY 8(d): sum over the launch's source pairs of "
                                 "8(n_a+n_b)+4; the kernel streams block-merged rank lists instead, "
                                 "see stream_model"},
            "stage1_sort": (lambda ms, n_e, bits, tb=(2 if n <= 65536 and not os.environ.get("KSP_TAG32") else 4): {
                "kernel": "rocprim radix_sort_onesweep (stage 1 partitions the entries by their top key bits before the LDS hash grouping: the largest kernel group of the step)",
                "entries": n_e, "key_bits": bits, "passes": (bits + 7) // 8, "ms": ms,
                "bytes": n_e * (8 + tb) * 2 * ((bits + 7) // 8),
                "GBps": (n_e * (8 + tb) * 2 * ((bits + 7) // 8)) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                "frac_of_hbm_peak": (n_e * (8 + tb) * 2 * ((bits + 7) // 8)) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                "note": f"8-byte key + {tb}-byte tag read and written once per 8-bit pass (histogram pass not counted)"})(
                    stats.get("ms_sort", 0.0) / max(1, args.steps), int(stats.get("sort_entries", 0)), int(stats.get("sort_bits", 0))),
            "stream_model": {"bytes_per_launch": stats["stream_bytes"], "GBps": stream_gbs,
                             "frac_of_peak": stream_gbs / HBM_PEAK_GBS,
                             "note": "4-byte ranks of both block lists per tile (what k_join must read)"},
            "stage_ms": {"build_blocks": ms_build, "join": ms_join,
                         "other (gather, D2H, sync)": 1e3 * elapsed / args.steps - ms_build - ms_join},
        }
        if world == 1 and args.cpu_sample > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(sk, args.cpu_sample)
            except Exception as ex:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {ex}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
