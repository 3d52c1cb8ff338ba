#!/usr/bin/env python3
"""bench.py — pairwise containment throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (what src/pairwise.cpp:194-237 does on the CPU) over one batch of
synthetic sketches that already sit in HBM as sorted uint64 runs:
    stage 1  build_blocks  (partition by key, group equal keys, prune singletons, rank-encode, order the sources,
                            one posting list per 128-source block; N > 1: per-rank hash-range slices + all-gather)
    stage 2  join          (LDS-tiled intersection of this rank's tile range -> edges in HBM)
    gather   RCCL point-to-point gather of the edge lists to rank 0 (N > 1 only)
    D2H      rank 0 copies the edges to pinned host memory (the hand-over to the TSV writer)
Nothing about the data is handed to the engine from the host side (no key range, no cached layout): tile
cuts and buffer sizes are taken from each step's own build.

Workload: BASELINE.json configs[1] ("10k sourmash signatures, scaled=1000, k=31" -> synthetic C2: 10 000
sketches, n ~ N(5000, 1500), hashes < 2^64/1000).  N > 1, default: WEAK scaling — the source count grows as
10 000 * sqrt(N) so that every GPU keeps the pair count of the 1-GPU job.  `--scaling strong --config C3` is
BASELINE.json configs[2] ("100k genomes, 1 vs 8 GPU row-block shard"): the same 100 000 sources on N GPUs.
Every rank holds the full sketch set; stage 1 is sharded by hash range (one RCCL all-gather of the block-list
slices), tiles are sharded by estimated work, the edges are gathered to rank 0.

value = whole-job source pairs per second = [S(S-1)/2] * K / t, t = max over ranks of the wall time of K
steps bracketed by barrier + torch.cuda.synchronize().

The JSON line also carries (N = 1):
  roofline       HBM roofline of the whole step: the bytes the step MUST move (SURVEY 8d "compulsory":
                 8 * sum(n) read once + 16 bytes per non-zero pair written) / step time / 8 TB/s — a fraction
                 <= 1 — plus the dominant kernel group and, per kernel group of stage 1 + the join, HIP-event
                 time, modelled bytes and (when profiles/traffic.json was measured on the same kernels) PMC bytes
  naive_pairwise the pairwise-merge-equivalent rate A/t (SURVEY 8d's per-pair bytes; exceeds the HBM peak by
                 design: the engine never reads a sketch once per source pair) — NOT a roofline
  cpu_baseline   the reference algorithm (oracle port) on the host cores, 1 thread and all cores
  other_configs  one timed step each of C3 / C4 / C5 at full size (build + join + D2H into pinned memory)
"""
from __future__ import annotations

import argparse
import glob
import hashlib
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


def kernel_source_hash() -> str:
    """sha256 over the HIP sources: profiles/traffic.json is only quoted when it was measured on these kernels."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "kspider_amd", "csrc", "*.hip*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def naive_pairwise_bytes(sizes: np.ndarray, n_sources: int) -> int:
    """SURVEY 8(d) per-pair figure B(a,b) = 8 (n_a + n_b) + 4 summed over all source pairs:
    A = 8 (N - 1) sum(n) + 4 N (N - 1) / 2."""
    return int(8 * (n_sources - 1) * int(sizes.sum()) + 4 * (n_sources * (n_sources - 1) // 2))


def cpu_baseline(sk, sample_sources: int, gpu_edges: np.ndarray | None = None) -> dict:
    """Reference algorithm (oracle restatement of src/pairwise.cpp:194-237: inverted-index walk, Combo pair lists,
    4 096 mutex-protected open-addressing submaps, static colour slices) on the host cores, on a bounded sample of
    the same workload, swept over user_threads.  A reported baseline, not a target.

    Also the CHECKER of the timed run: the edges of the 1-thread run are compared, row for row, with `gpu_edges`
    — the pinned host buffer the last TIMED step of the pipelined GPU path delivered (copied aside after the
    clock stopped) — the same comparison /root/reference/test/validate.py:100-108 makes per key."""
    import oracle
    sub = sk.subset(sample_sources)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    co, src, w = oracle.build_colors(sub.keys, sub.offsets)
    t_index = time.perf_counter() - t0
    n = sub.n_sources
    pairs = n * (n - 1) // 2
    runs = []
    verified = None
    n_edges = n_updates = 0
    budget_s = float(os.environ.get("KSP_BENCH_CPU_BUDGET", "75"))   # wall budget of the whole sweep
    t_sweep = time.perf_counter()
    for threads in sorted({1, min(8, avail), min(32, avail), min(64, avail), max(1, avail)}):
        if runs and time.perf_counter() - t_sweep + runs[0]["secs"] > budget_s:
            runs.append({"cores": threads, "skipped": "sweep budget"})
            continue
        want = threads == 1 and gpu_edges is not None
        secs, n_edges, n_updates, ed = oracle.accumulate_mem(co, src, w, threads, want_edges=want)
        runs.append({"cores": threads, "secs": secs, "value": pairs / secs})
        if want:
            g = gpu_edges[gpu_edges["source_2"] < n] if n < sk.n_sources else gpu_edges   # (edges among the sample's sources)
            g = g[np.lexsort((g["source_2"], g["source_1"]))]
            verified = bool(len(g) == len(ed) and (g["source_1"] + 1 == ed["source_1"]).all()
                            and (g["source_2"] + 1 == ed["source_2"]).all() and (g["shared"] == ed["shared"]).all())
    timed = [r for r in runs if "value" in r]
    best = max(timed, key=lambda r: r["value"])
    why = ""
    if best["cores"] != max(r["cores"] for r in timed):
        why = ("; more threads are not faster: every update takes one of 4 096 std::mutex locks (src/pairwise.cpp:22-27), "
               "a thread that finds its submap locked sleeps in the kernel (futex) and the pairs of a cluster keep "
               "hitting the same submaps")
    dropin = None
    if os.environ.get("KSP_BENCH_DROPIN", "1") == "1":
        try:
            dropin = dropin_end_to_end(oracle, sub, co, src, w, best["cores"])
        except Exception as ex:   # (never takes the bench line down)
            dropin = {"error": str(ex)}
    return {
        "dropin_s": dropin,
        "value": best["value"], "unit": "pairs/s", "cores": best["cores"], "kind": "port",
        "sample": f"first {n} of the workload's sources ({int(sub.offsets[-1])} hashes, {len(w)} colours, "
                  f"{n_updates} map updates, {n_edges} non-zero pairs); accumulate region only "
                  f"(src/pairwise.cpp:200-239 equivalent); colour index build {t_index:.1f} s not counted; "
                  f"host has {avail} usable cores; best of the thread sweep in `runs`{why}",
        "secs": best["secs"], "runs": runs, "edges_equal_gpu": verified,
    }


def dropin_end_to_end(oracle, sub, co, src, w, cpu_threads: int) -> dict:
    """The reference's actual API end to end on index files (part of the CPU-baseline leg: the oracle writes the three
    .bin files of the same sample and runs its restatement of kSpider::pairwise on them): `pairwise PREFIX T` of
    this library — index load, device round trip, TSV — next to the restated reference, TSVs compared byte for byte."""
    import shutil
    import subprocess
    import tempfile
    from kspider_amd import engine
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="ksp_dropin_", dir=base)
    try:
        prefix = os.path.join(d, "idx")
        oracle.write_index(prefix, co, src, w, None, np.diff(sub.offsets.astype(np.int64)).astype(np.uint32))
        exe = os.path.join(os.path.dirname(engine.LIB_PATH), "pairwise")
        threads = min(64, len(os.sched_getaffinity(0)))
        env = dict(os.environ, KSPIDER_VERBOSE="1")
        out = None
        for _ in range(2):   # (the first run pays the GPU start-up of a fresh process; the second is reported)
            t = time.perf_counter()
            out = subprocess.run([exe, prefix, str(threads)], capture_output=True, text=True, check=True, env=env)
            wall = time.perf_counter() - t
        got = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
        phases = {}
        for line in out.stdout.splitlines():
            for key, name in (("mapping colors to groups:", "load_s"), ("kmer counting:", "kmer_counts_s"),
                              ("pairwise hashmap construction:", "construction_s")):
                if line.startswith(key):
                    phases[name] = float(line.split(":")[1].split()[0])
            if "device round trip" in line:
                phases["device_round_trip_s"] = float(line.split("device round trip")[1].split()[0])
        # (the restated reference prints its phase line on stdout, as the reference does: kept off this program's stdout,
        #  which carries the one JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            t = time.perf_counter()
            secs, n_rows, _ = oracle.ref_pairwise(prefix, cpu_threads)
            cpu_wall = time.perf_counter() - t
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        want = open(prefix + "_kSpider_pairwise.tsv", "rb").read()
        return {"wall_s": wall, **phases, "tsv_bytes": len(got), "host_threads": threads,
                "index_bytes": sum(os.path.getsize(prefix + f) for f in ("_color_to_sources.bin", "_color_count.bin", "_groupID_to_kmerCount.bin")),
                "cpu_restatement_wall_s": cpu_wall, "cpu_restatement_accumulate_s": secs, "cpu_threads": cpu_threads,
                "tsv_identical": bool(got == want), "rows": int(n_rows),
                "note": "wall_s: the whole `pairwise PREFIX T` process of this library (start-up, index load, device, TSV); "
                        "construction_s: the region the reference times as 'pairwise hashmap construction'"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def edge_checksum(ev: np.ndarray) -> int:
    """Order-independent checksum of an edge set: sum of (source_1 * 2^40 + source_2 * 2^20 + shared) mod 2^64."""
    if len(ev) == 0:
        return 0
    x = (ev["source_1"].astype(np.uint64) << np.uint64(40)) + (ev["source_2"].astype(np.uint64) << np.uint64(20)) + ev["shared"]
    return int(x.sum(dtype=np.uint64))


def holder_pair_sum(torch, keys_d) -> int:
    """sum over distinct hashes of C(holders, 2), from a plain torch sort of the keys on the GPU — independent of the
    engine.  Every source pair shares exactly the keys both hold, so the sum of all `shared` counts must equal it
    (the identity tests/test_configs_gpu.py checks on the host)."""
    srt = torch.sort(keys_d)[0]
    head = torch.ones_like(srt, dtype=torch.bool)
    head[1:] = srt[1:] != srt[:-1]
    del srt
    pos = torch.nonzero(head).flatten()
    del head
    cnt = torch.diff(pos, append=torch.tensor([keys_d.numel()], device=keys_d.device, dtype=pos.dtype))
    return int((cnt * (cnt - 1) // 2).sum().item())


def run_two_in_flight(cfg, torch, dev, engine, eng0, keys_d, sk, edges_h0, cnt0, jobs: int = 12) -> dict:
    """`jobs` independent jobs (build + join in pieces + edges to pinned host memory) on two engines driven by two host
    threads: ms per job with two in flight.  Every job's edge count must equal the single job's."""
    import threading
    eng1 = engine.Engine(dev.index or 0)
    try:
        edges_h1 = torch.empty_like(edges_h0).pin_memory()
        engs, bufs = (eng0, eng1), (edges_h0, edges_h1)
        streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        # (the second engine's buffers grow to their final size outside the clock)
        eng1.build_blocks(keys_d.data_ptr(), sk.offsets, stream=streams[1].cuda_stream)
        eng1.join_to_host(0, eng1.num_tiles, edges_h1.data_ptr(), edges_h1.shape[0], stream=streams[1].cuda_stream)
        torch.cuda.synchronize(dev)
        start = threading.Barrier(3)
        errs, counts = [], [[], []]

        def worker(i: int):
            try:
                start.wait()
                for _ in range(jobs // 2):
                    engs[i].build_blocks(keys_d.data_ptr(), sk.offsets, stream=streams[i].cuda_stream)
                    counts[i].append(engs[i].join_to_host(0, engs[i].num_tiles, bufs[i].data_ptr(), bufs[i].shape[0],
                                                          stream=streams[i].cuda_stream))
            except Exception as ex:   # (reported by the caller: a thread must not die silently)
                errs.append(ex)

        th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t_ in th:
            t_.start()
        start.wait()
        t = time.perf_counter()
        for t_ in th:
            t_.join()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t
        if errs:
            raise errs[0]
        done = len(counts[0]) + len(counts[1])
        ok = all(c == cnt0 for c in counts[0] + counts[1])
        return {"two_in_flight": {"jobs": done, "ms_per_job": 1e3 * wall / max(done, 1), "counts_equal_single_job": bool(ok),
                                  "note": "two engines / host threads / streams: job k's result travels while job k + 1 is built"}}
    finally:
        eng1.close()


def run_other_config(cfg: str, torch, dev, engine, synth) -> dict:
    """One timed step of a full-size BASELINE config on this GPU: build + join + the edges in pinned host memory
    (a first, untimed step sizes the buffers).  The join runs in pieces, every piece copied over PCIe under the
    join of the next (ksp_engine_join_to_host).  No host-side checks inside the timed region: after the clock,
    sum(shared) is compared with sum over hashes of C(holders, 2) from a plain torch sort of the keys; the full
    checks live in tests/test_configs_gpu.py."""
    t = time.perf_counter()
    sk = synth.generate(cfg)
    t_gen = time.perf_counter() - t
    n = sk.n_sources
    keys_d = torch.from_numpy(sk.keys.view(np.int64)).to(dev)
    eng = engine.Engine(dev.index or 0)
    stream = torch.cuda.current_stream(dev)
    out = {"config": cfg, "n_sources": n, "hashes": int(sk.offsets[-1]), "pairs": n * (n - 1) // 2,
           "generate_s": round(t_gen, 1)}
    edges_h = None
    cnt = 0
    for timed in (False, False, True):   # (sizing step, one warm-up step with the final buffers, the timed step)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        eng.build_blocks(keys_d.data_ptr(), sk.offsets, stream=stream.cuda_stream)
        T = eng.num_tiles
        if edges_h is None:   # (untimed step: a join with no room reports the count the pinned buffer is sized from)
            try:
                cnt = eng.join_to_host(0, T, 0, 0, stream=stream.cuda_stream)
            except engine.KspError as ex:
                if ex.code != engine.KSP_E_OVERFLOW:
                    raise
                cnt = ex.count
            edges_h = torch.empty((cnt + cnt // 8 + 1, 16), dtype=torch.uint8).pin_memory()
            continue
        cnt = eng.join_to_host(0, T, edges_h.data_ptr(), edges_h.shape[0], stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t
        if not timed:
            continue
        st = eng.stats()
        out.update({"step_ms": 1e3 * wall, "build_ms": st["ms_build"], "join_ms": st["ms_join"],
                    "d2h_and_host_ms": 1e3 * wall - st["ms_build"] - st["ms_join"], "nonzero_pairs": int(cnt),
                    "pairs_per_s": out["pairs"] / wall, "active_tiles": int(st["last_active_tiles"]),
                    "tiles": int(T), "partition_kind": int(st["partition_kind"]), "stage1_kind": int(st["stage1_kind"]),
                    "compulsory_GBps": (8 * out["hashes"] + 16 * cnt) / wall / 1e9,
                    "result": "edges in pinned host memory; join in pieces, each copied under the join of the next"})
    # Throughput with two jobs in flight (after the one-job clock above): two engines, two host threads, two streams — while
    # the result of job k travels over PCIe (hundreds of MB: longer than its join), job k + 1 is built.  One job alone cannot
    # hide that copy behind anything but its own join; a service that runs job after job can.
    try:
        out.update(run_two_in_flight(cfg, torch, dev, engine, eng, keys_d, sk, edges_h, cnt))
    except Exception as ex:
        out["two_in_flight_note"] = f"not measured: {ex}"
    # after the clock: sum of all shared counts == sum over hashes of C(holders, 2) (torch sort on the GPU)
    try:
        ev = edges_h[:cnt].numpy().view(engine.EDGE_DTYPE).reshape(-1)
        got = int(ev["shared"].sum(dtype=np.uint64))
        want = holder_pair_sum(torch, keys_d)
        out["sum_shared"] = got
        out["sum_shared_equals_holder_pairs"] = bool(got == want)
        out["checksum"] = edge_checksum(ev)
    except Exception as ex:
        out["sum_shared_equals_holder_pairs"] = None
        out["check_note"] = f"identity check failed to run: {ex}"
    eng.close()
    del keys_d, edges_h
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: sources = N(config) * sqrt(gpus); strong: the config's own size on every GPU count")
    ap.add_argument("--n-sources", type=int, default=0, help="override the source count (debug)")
    ap.add_argument("--cpu-sample", type=int, default=10000, help="sources in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--other-configs", default="C3,C4,C5",
                    help="full-size configs timed once each after the main measurement (N = 1 only; '' = skip)")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra steps with per-phase events (N = 1; 0 = skip)")
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
    if args.n_sources:
        n_sources = args.n_sources
    elif args.scaling == "weak":
        n_sources = int(round(base_n * math.sqrt(world)))
    else:
        n_sources = base_n
    sk = synth.generate(args.config, n_sources=n_sources)   # identical on every rank (deterministic)
    n = sk.n_sources
    total_pairs = n * (n - 1) // 2
    n_hashes = int(sk.offsets[-1])

    keys_d = torch.from_numpy(sk.keys.view(np.int64)).to(dev)
    stream = torch.cuda.current_stream(dev)
    eng = engine.Engine(local_rank)
    sharded = world > 1
    # double-buffered results: the D2H copy of step k runs on its own stream under stage 1 of step k + 1
    edges_dd = [None, None]
    edges_hh = [None, None]
    copy_stream = torch.cuda.Stream(device=dev)
    copied = [None, None]     # event: the D2H copy out of buffer i has finished
    in_flight = [None, None]  # keeps the gathered device tensor alive until its copy is done
    step_no = [0]

    stats = {"ms_join": 0.0, "ms_build": 0.0, "ms_sort": 0.0, "edges": 0, "stream_bytes": 0, "xchg_bytes": 0,
             "regrown": 0}

    pending = [None]   # the step whose join has been launched but whose results have not been handed on yet

    def hand_over(job):
        """Rank 0 (after the gather) copies the edges of a finished join to pinned host memory on the copy stream."""
        buf, cnt, edges_d, record = job["buf"], job["cnt"], job["edges_d"], job["record"]
        local = edges_d[:cnt]
        if world > 1 and backend != "nccl":
            local = local.cpu()     # gloo exchanges host tensors (test hook only)
        allv = kdist.gather_edges(local, dst=0) if world > 1 else local
        if rank == 0:
            if edges_hh[buf] is None or edges_hh[buf].shape[0] < allv.shape[0]:
                edges_hh[buf] = torch.empty((allv.shape[0] + allv.shape[0] // 8 + 1, 16), dtype=torch.uint8).pin_memory()
                stats["regrown"] += 1
            edges_h = edges_hh[buf]
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(job["ready"])
                edges_h[: allv.shape[0]].copy_(allv, non_blocking=True)
                copied[buf] = torch.cuda.Event()
                copied[buf].record(copy_stream)
            in_flight[buf] = allv
        if record:
            stats["edges"] = int(allv.shape[0]) if rank == 0 else cnt
            stats["last_buf"] = buf

    def collect(record_join: bool, known_count=None):
        """Count of the launched join (it has finished whenever a later build on the same stream has returned);
        known_count: ksp_engine_step_launch has collected it already."""
        job = pending[0]
        if job is None:
            return None
        pending[0] = None
        job["cnt"] = eng.join_wait() if known_count is None else known_count
        if job["record"] and record_join:
            # (one field: this sits between a build and the launch of its join)
            stats["ms_join"] += eng.ms_join() if known_count is None else eng.prev_ms_join
        return job

    def step(record: bool):
        # Software pipeline over the steps: the join of step k is only LAUNCHED here; the next step's stage 1 is queued
        # behind it on the same stream, and the host collects the count and hands the edges on (gather, D2H on the copy
        # stream) while the device is already busy — the device does not wait for the host between a join and the next
        # build.  Every step still takes its tile range and buffer sizes from ITS OWN build (the engine's source order,
        # and with it the tile numbering, differs from build to build).
        buf = step_no[0] & 1
        step_no[0] += 1
        if copied[buf] is not None:
            copied[buf].synchronize()       # the copy that last read this buffer pair (two steps ago; long done)
        launched = False
        if sharded:   # sharded by hash range + all-gather of the block-list slices: every rank assembles the same lists
            stats["xchg_bytes"] = kdist.build_blocks_sharded(eng, keys_d.data_ptr(), sk.offsets, world, rank, dev,
                                                             stream=stream.cuda_stream)
            cuts = eng.balanced_cuts(world)
            t0, t1 = cuts[rank], cuts[rank + 1]
            need = int(min(eng.edge_bound(t0, t1), 1 << 27)) + 1
        else:
            # build + cuts + bound + join launch in one call of the library: nothing of Python between a build and its join
            have = edges_dd[buf].shape[0] if edges_dd[buf] is not None else 0
            t0, t1, bound, launched, prev_cnt = eng.step_launch(keys_d.data_ptr(), sk.offsets, rank, world,
                                                                edges_dd[buf].data_ptr() if have else 0, have, stream=stream.cuda_stream)
            need = int(min(bound, 1 << 27)) + 1
        prev = collect(True, prev_cnt if not sharded else None)   # (the previous join ran in front of this build: the wait returns at once)
        if not launched:
            if edges_dd[buf] is None or edges_dd[buf].shape[0] < need:
                edges_dd[buf] = torch.empty((need + need // 8, 16), dtype=torch.uint8, device=dev)
                stats["regrown"] += 1
            eng.join_launch(t0, t1, edges_dd[buf].data_ptr(), edges_dd[buf].shape[0], stream=stream.cuda_stream)
        edges_d = edges_dd[buf]
        ready = torch.cuda.Event()
        ready.record(stream)
        pending[0] = {"buf": buf, "edges_d": edges_d, "record": record, "ready": ready, "cnt": 0}
        # ---- from here on the device is busy with this step's join ----
        if record:
            st = eng.stats()
            stats["ms_build"] += st["ms_build"]
            stats["ms_sort"] += st["ms_sort"]
            stats["sort_entries"], stats["sort_bits"] = st["sort_entries"], st["sort_bits"]
            stats["partition_kind"] = st["partition_kind"]
            stats["active_tiles"] = int(st["n_active_tiles"])
        if prev is not None:
            hand_over(prev)

    def drain():
        """Results of the last launched join (end of a run of steps)."""
        job = collect(True)
        if job is not None:
            stats["stream_bytes"] = eng.stats()["last_stream_bytes"]
            hand_over(job)
            return job["cnt"]
        return 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 2)):   # (at least two: both buffer pairs get their size before the clock starts)
        step(False)
    drain()
    stats["regrown"] = 0
    # (no collector pause inside the timed region: one step in ~50 took 9 ms longer with the cyclic GC left on, on an
    #  otherwise idle box; every step allocates a few event objects)
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t_start = time.perf_counter()
    trace = os.environ.get("KSP_BENCH_TRACE") == "1"   # (diagnostic: host time of every step's calls, on stderr)
    for _ in range(args.steps):
        t_s = time.perf_counter()
        step(True)
        if trace and rank == 0:
            print(f"step {1e3 * (time.perf_counter() - t_s):.3f} ms", file=sys.stderr)
    drain()                 # (the last step's count, gather and D2H: inside the timed region)
    barrier()
    elapsed = time.perf_counter() - t_start
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # the result of the last TIMED step (pipelined path: join_launch -> next build -> join_wait -> D2H on the copy
    # stream), copied aside now that the clock has stopped: the profiling steps below reuse the pinned buffers
    final_edges = None
    if rank == 0 and "last_buf" in stats:
        copy_stream.synchronize()
        final_edges = edges_hh[stats["last_buf"]][: stats["edges"]].numpy().view(engine.EDGE_DTYPE).reshape(-1).copy()

    steps = max(1, args.steps)
    ms_step = 1e3 * elapsed / steps
    ms_join = stats["ms_join"] / steps
    ms_build = stats["ms_build"] / steps
    ms_part = stats["ms_sort"] / steps
    stage_from_phase_pass = False

    if rank == 0:
        E = int(stats["edges"])
        compulsory = 8 * n_hashes + 16 * E                 # every hash read once, every non-zero pair written once
        achieved = compulsory / (ms_step * 1e-3) / 1e9 / world
        # ---- per kernel group: HIP events at the phase starts of a few extra, untimed steps ----------------
        groups = []
        tag_b = 2 if n <= 65536 and not os.environ.get("KSP_TAG32") else 4
        if world == 1 and args.profile_steps > 0:
            eng.set_profiling(True)
            acc, joins = {}, 0.0
            order = []
            for _ in range(args.profile_steps):
                step(False)
                drain()
                torch.cuda.synchronize(dev)
                for name, ms in eng.phase_times():
                    if name not in acc:
                        order.append(name)
                    acc[name] = acc.get(name, 0.0) + ms
                joins += eng.stats()["ms_join"]
            eng.set_profiling(False)
            # (the timed steps carry no timing events — an event record is a ~6 us bubble in the stream, ksp_engine_step_launch
            #  records none unless phases are being timed: the split of a step into build / partition / join comes from this pass)
            if ms_part == 0.0 and "partition" in acc:
                ms_part = acc["partition"] / args.profile_steps
            if ms_build == 0.0 and acc:
                ms_build = sum(acc.values()) / args.profile_steps
                stage_from_phase_pass = True
            if ms_join == 0.0 and joins:
                ms_join = joins / args.profile_steps
            kept = None
            # bytes each group has to move per step; None where no simple model applies
            kind = stats.get("partition_kind")
            model = {
                # 3: segment partition (boundary pass reads the keys, the scatter reads them again and writes key + tag);
                # 2: paged levels (level 1 reads keys, writes key + tag + digit; level 2 reads digit, key + tag, writes key + tag);
                # else the library's radix passes
                "partition": n_hashes * (8 + 8 + (8 + tag_b)) if kind == 3
                else n_hashes * ((8 + 8 + tag_b + 1) + (1 + 8 + tag_b) + (8 + tag_b)) if kind == 2
                else n_hashes * ((8 + tag_b) * 2 * ((int(stats.get("sort_bits", 16)) + 7) // 8) + 8),
                "bucket grouping": n_hashes * (8 + 4 + 4 + tag_b),
                "tags + source sizes": n_hashes * tag_b,
            }
            traffic = {}
            prof = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(prof):
                try:
                    tj = json.load(open(prof))
                    if tj.get("kernel_source_hash") == kernel_source_hash():
                        traffic = tj.get("groups", {})
                except Exception:
                    traffic = {}
            for name in order + ["join"]:
                ms = (acc[name] if name != "join" else joins) / args.profile_steps
                mb = model.get(name) if name != "join" else stats["stream_bytes"] + 16 * E
                cb = traffic.get(name, {}).get("hbm_bytes_per_step")
                groups.append({"group": name, "ms": ms, "model_bytes": mb,
                               "model_GBps": (mb / (ms * 1e-3) / 1e9) if (mb and ms > 0) else None,
                               "counter_bytes": cb,
                               "frac_of_hbm": ((cb or mb) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ((cb or mb) and ms > 0) else None})
        dominant = max(groups, key=lambda g: g["ms"]) if groups else {"group": "stage 1", "ms": ms_build}
        step_traffic = None
        if groups and all(g["counter_bytes"] is not None for g in groups):
            step_traffic = int(sum(g["counter_bytes"] for g in groups))
        naive = naive_pairwise_bytes(sk.sizes, n)
        out = {
            "metric": "sketch-pairs/sec (NxN containment)", "value": total_pairs * args.steps / elapsed,
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u64 keys -> u32 ranks, u32 counters", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{synth.CONFIGS[args.config]['idx'] - 1}]: {args.config} synthetic "
                                    f"sketches, {n} sources, {n_hashes} hashes; "
                                    + (f"weak scaling: sources = {base_n}*sqrt(n_gpus)" if args.scaling == "weak"
                                       else "strong scaling: the same sources on every GPU count")),
                       "n_sources": n, "pairs": total_pairs, "nonzero_pairs": E,
                       "checksum": edge_checksum(final_edges) if final_edges is not None else None,
                       "verified_against_cpu_baseline": None,
                       "tiles": int(eng.num_tiles), "active_tiles": stats.get("active_tiles"),
                       "key_range": "found on the device (no key_bits hint)",
                       "buffers_regrown_in_timed_region": stats["regrown"],
                       "parallelism": (f"tile-range shard x{world} by estimated work, cuts from every step's own build; stage 1 "
                                       + ("sharded by hash range + RCCL all-gather of the block-list slices "
                                          f"({stats['xchg_bytes'] / 1e6:.0f} MB received per rank)" if sharded
                                          else "on the one GPU") + "; RCCL p2p gather of the edges to rank 0")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": step_traffic,
                         "kernel": f"whole step; dominant kernel group: {dominant['group']} ({dominant['ms']:.3f} ms)",
                         "bytes_per_step": compulsory,
                         "note": "compulsory traffic of one step (SURVEY 8d): 8 B x every hash read once + 16 B x every "
                                 "non-zero pair written once, / ms_per_step / n_gpus; `traffic` = PMC bytes per step summed "
                                 "over the kernel groups when profiles/traffic.json matches these kernels",
                         "groups": groups},
            "naive_pairwise": {"bytes_per_step": naive, "GBps": naive / (ms_step * 1e-3) / 1e9,
                               "x_hbm_peak": naive / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS / world,
                               "note": "SURVEY 8(d) per-pair bytes 8(n_a+n_b)+4 over all pairs / step time: what a "
                                       "pair-by-pair merge would have to stream; not a roofline of this engine"},
            "stage_ms": ({"build_blocks": ms_build, "of which partition": ms_part, "join": ms_join,
                          "measured_in": "the phase pass after the timed steps (one timing event per phase start: ~0.05 ms of "
                                         "bubbles per step that the timed steps, which carry no events, do not have)"}
                         if stage_from_phase_pass else
                         {"build_blocks": ms_build, "of which partition": ms_part, "join": ms_join,
                          "other (gather, D2H, sync, host)": ms_step - ms_build - ms_join}),
        }
        if world == 1 and args.cpu_sample > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(sk, args.cpu_sample, final_edges)
                out["config"]["verified_against_cpu_baseline"] = out["cpu_baseline"].pop("edges_equal_gpu")
                out["dropin_s"] = out["cpu_baseline"].pop("dropin_s")
            except Exception as ex:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {ex}"}
        if world == 1 and args.other_configs and not args.n_sources:
            del keys_d
            others = []
            for cfg in [c for c in args.other_configs.split(",") if c and c != args.config]:
                try:
                    others.append(run_other_config(cfg, torch, dev, engine, synth))
                except Exception as ex:
                    others.append({"config": cfg, "error": str(ex)})
            out["other_configs"] = others
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
