"""Ray-tile sharding for multi-GPU renders.

Rays are the only sharded dimension; the scene is replicated on every GPU (as the reference replicates
it on every IPU, src/IpuScene.cpp:473-483, and deals ray batches round-robin over replicas, :676-684).
Here the shard unit is a tile of TILE_ROWS image rows, dealt round-robin to the ranks, which balances
the cheap (walls) and expensive (glass, monkey) parts of the frame. There is no exchange while a frame
renders; at frame end rank 0 collects the tiles with ONE gather (RCCL when the backend is nccl).
Because every pixel owns its RNG stream, the assembled image is bit-identical for any rank count.
"""
from __future__ import annotations

import numpy as np

TILE_ROWS = 8      # = the height of the 8x8 work tiles; 2880 rows / 8 = 360 bands = 45 per rank at 8 GPUs (16-row bands: 22.5)


def rank_pixels(width: int, height: int, rank: int, world: int, tile_rows: int = TILE_ROWS):
    """(rows, cols) of the pixels rank `rank` renders, in the order they appear in its ray stream."""
    rows = np.arange(height)
    mine = rows[(rows // tile_rows) % world == rank]
    rr, cc = np.meshgrid(mine, np.arange(width), indexing="ij")
    return rr.reshape(-1), cc.reshape(-1)


def padded_count(width: int, height: int, world: int, tile_rows: int = TILE_ROWS) -> int:
    """Largest per-rank pixel count: gather needs equal-sized contributions, short ranks pad."""
    return max(rank_pixels(width, height, r, world, tile_rows)[0].size for r in range(world))


_index_cache = {}


def _scatter_indices(width, height, world, tile_rows, device):
    """Flat pixel index of every gathered row, rank after rank (cached: it only depends on the geometry)."""
    import torch
    key = (width, height, world, tile_rows, str(device))
    if key not in _index_cache:
        n_pad = padded_count(width, height, world, tile_rows)
        idx = np.full((world, n_pad), -1, dtype=np.int64)
        for r in range(world):
            rows, cols = rank_pixels(width, height, r, world, tile_rows)
            idx[r, : rows.size] = rows * width + cols
        _index_cache[key] = torch.from_numpy(idx).to(device)
    return _index_cache[key]


def gather_frame(dist, rgb, width: int, height: int, tile_rows: int = TILE_ROWS):
    """One collective per frame. `rgb`: this rank's [n_r, 3] float32 torch tensor (any device the
    process group supports). Returns the [height, width, 3] frame on rank 0, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n_pad = padded_count(width, height, world, tile_rows)
    send = rgb
    if rgb.shape[0] != n_pad:
        send = torch.zeros(n_pad, 3, dtype=rgb.dtype, device=rgb.device)
        send[: rgb.shape[0]] = rgb
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, bufs, dst=0)
    if rank != 0:
        return None
    idx = _scatter_indices(width, height, world, tile_rows, rgb.device)
    frame = torch.zeros(height * width + 1, 3, dtype=rgb.dtype, device=rgb.device)    # last row swallows the padding
    frame[idx.reshape(-1)] = torch.stack(bufs).reshape(-1, 3)
    return frame[:-1].view(height, width, 3)
