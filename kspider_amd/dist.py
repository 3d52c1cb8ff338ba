"""One-process-per-GPU sharding of the pair space and the final edge gather.

The N x N pair space is cut into 128 x 128 tiles (block pairs); the tiles of the upper
triangle, in row-major order, are split into `world_size` contiguous ranges of equal tile
count — a row-block-wise shard with equal triangular area.  Tiles are
independent, so the only exchange step is the final variable-length gather of the edge
lists to rank 0: a batch of point-to-point sends (ncclSend/ncclRecv inside one group call
with the "nccl" backend = RCCL over xGMI; the same code runs on "gloo" for CPU tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

EDGE_BYTES = 16  # struct ksp_edge


def tile_range(num_tiles: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous slice [t0, t1) of the row-major tile list owned by `rank`."""
    return (num_tiles * rank) // world_size, (num_tiles * (rank + 1)) // world_size


def num_tiles_for(n_sources: int, tb: int = 128) -> int:
    nb = (n_sources + tb - 1) // tb
    return nb * (nb + 1) // 2


def tile_of_pair(a: np.ndarray, b: np.ndarray, n_sources: int, tb: int = 128) -> np.ndarray:
    """Row-major tile index of source pairs (a < b); used by tests to shard reference results."""
    nb = (n_sources + tb - 1) // tb
    i = (a // tb).astype(np.int64)
    j = (b // tb).astype(np.int64)
    return i * nb - i * (i - 1) // 2 + (j - i)


def gather_edges(local: torch.Tensor, dst: int = 0, group=None) -> torch.Tensor | None:
    """Gather variable-length edge lists ([n, 16] uint8 tensors) to rank `dst`.

    Returns the concatenation (rank order) on `dst`, None elsewhere.  Works on CUDA tensors
    (backend nccl = RCCL) and CPU tensors (gloo).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert local.dtype == torch.uint8 and local.dim() == 2 and local.shape[1] == EDGE_BYTES
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, mine, group=group) if local.is_cuda else dist.all_gather(
        list(counts.split(1)), mine, group=group)
    counts_h = counts.cpu().tolist()
    if rank == dst:
        total = int(sum(counts_h))
        out = torch.empty((total, EDGE_BYTES), dtype=torch.uint8, device=local.device)
        ops, off = [], 0
        for r in range(world):
            n = int(counts_h[r])
            if r == dst:
                out[off:off + n].copy_(local)
            elif n:
                ops.append(dist.P2POp(dist.irecv, out[off:off + n], r, group))
            off += n
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    if local.shape[0]:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst, group)]):
            req.wait()
    return None
