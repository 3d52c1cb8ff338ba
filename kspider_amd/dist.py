"""One-process-per-GPU sharding of the pair space and the final edge gather.

The N x N pair space is cut into 128 x 128 tiles (block pairs); the tiles of the upper
triangle, in row-major order, are split into `world_size` contiguous ranges of equal
ESTIMATED WORK (ksp_engine_balanced_cuts: equal numbers of work-list shares — equal tile
counts would be badly unbalanced now that the work sits on the diagonal).  Tiles
are independent, so the only exchange step is the final variable-length gather of the edge
lists to rank 0: a batch of point-to-point sends (ncclSend/ncclRecv inside one group call
with the "nccl" backend = RCCL over xGMI; the same code runs on "gloo" for CPU tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

EDGE_BYTES = 16  # struct ksp_edge


def tile_range(num_tiles: int, world_size: int, rank: int) -> tuple[int, int]:
    """(tests only: equal tile COUNTS; the bench and the engine cut by estimated work, Engine.balanced_cuts)
    Contiguous slice [t0, t1) of the row-major tile list owned by `rank`."""
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


def build_blocks_sharded(eng, d_keys_ptr: int, h_offsets: np.ndarray, world_size: int, rank: int, device,
                         d_weights_ptr: int = 0, stream: int = 0, group=None):
    """Stage 1 sharded over the ranks (every rank holds the full sketch set).

    Rank r builds the block-list slices of its 1/world share of the hash range
    (`Engine.build_slice` / `slice_finish`, after a MIN all-reduce of the source labels that fixes the
    common source order), the slices are all-gathered (RCCL `all_gather_into_tensor`; host tensors
    with gloo), and every rank assembles the full lists (`Engine.assemble`).  With world_size 1 this
    is a plain `build_blocks`.  Returns the bytes this rank received in the exchange.
    """
    if world_size == 1:
        eng.build_blocks(d_keys_ptr, h_offsets, d_weights_ptr=d_weights_ptr, stream=stream)
        return 0
    eng.build_slice(d_keys_ptr, h_offsets, rank, world_size, d_weights_ptr=d_weights_ptr, stream=stream)
    on_gpu = dist.get_backend(group) == "nccl"
    comm_dev = device if on_gpu else torch.device("cpu")
    # common source order: element-wise MIN of the ranks' labels (n_sources x 4 bytes)
    labels = torch.empty(len(h_offsets) - 1, dtype=torch.int32, device=device)
    eng.slice_labels(labels.data_ptr(), stream=stream)
    if on_gpu:
        dist.all_reduce(labels, op=dist.ReduceOp.MIN, group=group)
    else:
        lab_h = labels.cpu()
        dist.all_reduce(lab_h, op=dist.ReduceOp.MIN, group=group)
        labels.copy_(lab_h)
    eng.slice_finish(labels.data_ptr(), stream=stream)
    mine = torch.from_numpy(eng.slice_sizes().astype(np.int64)).to(comm_dev)
    all_sz = torch.zeros(4 * world_size, dtype=torch.int64, device=comm_dev)
    if on_gpu:
        dist.all_gather_into_tensor(all_sz, mine, group=group)
    else:
        dist.all_gather(list(all_sz.split(4)), mine, group=group)
    sizes = all_sz.cpu().numpy().astype(np.uint64)
    nb = int(eng.stats()["n_blocks"])   # (may exceed ceil(sources / 128): spare blocks for cluster-aligned boundaries)
    lstride = max(4, int(sizes[0::4].max()))
    bigstride = max(1, int(sizes[2::4].max()))
    weighted = d_weights_ptr != 0

    def alloc(cols):   # (every word that assemble reads is written by the exchange: no zero fill per step)
        return torch.empty((world_size, cols), dtype=torch.int32, device=device)

    brk_all, info_all = alloc(lstride), alloc(lstride)
    bw_all = alloc(lstride) if weighted else None
    raw_all, pos_all, big_all = alloc(nb + 1), alloc(nb + 1), alloc(4 * bigstride)
    loc = [torch.empty(lstride, dtype=torch.int32, device=device), torch.empty(lstride, dtype=torch.int32, device=device),
           torch.empty(lstride, dtype=torch.int32, device=device) if weighted else None,
           torch.zeros(nb + 1, dtype=torch.int32, device=device), torch.zeros(nb + 1, dtype=torch.int32, device=device),
           torch.empty(4 * bigstride, dtype=torch.int32, device=device)]
    eng.slice_export(loc[0].data_ptr(), loc[1].data_ptr(), loc[2].data_ptr() if weighted else 0, loc[3].data_ptr(),
                     loc[4].data_ptr(), loc[5].data_ptr(), stream=stream)
    received = 0
    for out, src in ((brk_all, loc[0]), (info_all, loc[1]), (bw_all, loc[2]), (raw_all, loc[3]), (pos_all, loc[4]),
                     (big_all, loc[5])):
        if out is None:
            continue
        if on_gpu:
            dist.all_gather_into_tensor(out.view(-1), src, group=group)
        else:   # gloo: through host memory (test hook)
            parts = [torch.zeros(src.shape, dtype=src.dtype) for _ in range(world_size)]
            dist.all_gather(parts, src.cpu(), group=group)
            out.copy_(torch.stack(parts).to(device))
        received += out.numel() * 4 * (world_size - 1) // world_size
    eng.assemble(sizes, brk_all.data_ptr(), info_all.data_ptr(), bw_all.data_ptr() if weighted else 0, lstride,
                 raw_all.data_ptr(), pos_all.data_ptr(), big_all.data_ptr(), bigstride, stream=stream)
    return received


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
