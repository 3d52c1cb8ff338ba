"""world_size-2 gloo test of the N > 1 path: contiguous tile-range shards + the final
variable-length gather to rank 0 (the exchange step that runs over RCCL on the GPUs)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kspider_amd import dist as kdist
from kspider_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sk = synth.generate("C2", n_sources=n, mean_size=60, cluster_cap=12, seed=77)
        ref = oracle.brute_pairs(sk.keys, sk.offsets)
        T = kdist.num_tiles_for(n)
        t0, t1 = kdist.tile_range(T, world, rank)
        # stand-in for the engine's join over [t0, t1): the reference rows whose tile is mine
        tiles = kdist.tile_of_pair(ref["source_1"].astype(np.int64), ref["source_2"].astype(np.int64), n)
        mine = ref[(tiles >= t0) & (tiles < t1)]
        local = torch.from_numpy(np.frombuffer(mine.tobytes(), dtype=np.uint8).copy()).reshape(-1, 16)
        out = kdist.gather_edges(local, dst=0)
        if rank == 0:
            got = np.frombuffer(out.numpy().tobytes(), dtype=ref.dtype)
            got = np.sort(got, order=["source_1", "source_2"])
            ret["ok"] = bool(len(got) == len(ref) and (got == ref).all())
            ret["n"] = int(len(ref))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_join_plus_gather_equals_single(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, 400, ret), nprocs=world, join=True)
    assert ret["ok"] and ret["n"] > 100


def test_tile_ranges_partition_the_triangle():
    for n in (1, 127, 128, 129, 1000, 10000):
        T = kdist.num_tiles_for(n)
        for world in (1, 2, 3, 8):
            rs = [kdist.tile_range(T, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == T
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_tile_of_pair_matches_row_major_order():
    n = 700
    nb = (n + 127) // 128
    t = 0
    for i in range(nb):
        for j in range(i, nb):
            a = np.array([i * 128]); b = np.array([min(n - 1, j * 128 + 5)])
            if a[0] < b[0]:
                assert kdist.tile_of_pair(a, b, n)[0] == t
            t += 1
