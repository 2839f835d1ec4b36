"""Ray-band sharding for multi-GPU renders, Python side.

Rays are the only sharded dimension; the scene is replicated on every GPU (as the reference replicates
it on every IPU, src/IpuScene.cpp:473-483, and deals ray batches round-robin over replicas, :676-684).
The shard unit is a band of 8 image rows (one row of the kernel's 8x8 work tiles), band b to rank b % world, which
balances the cheap (walls) and expensive (glass, monkey) parts of the frame. WHO renders WHAT is defined once, in C
(ipu_ray_lib_amd/csrc/ray_shard.hpp): the single-process renderer `mi_group_render` (trace --gpus N) uses it
directly, and this module - used by the one-process-per-GPU ranks of bench.py - goes through the same functions as
exported by libmi_scene_host.so (mi_shard_*). There is no exchange while a frame renders; at frame end rank 0
collects the bands with ONE gather (RCCL when the backend is nccl). Because every pixel owns its RNG stream, the
assembled image is bit-identical for any rank count.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import host_lib, _check_host

TILE_ROWS = 8      # = the height of the 8x8 work tiles; 2880 rows / 8 = 360 bands = 45 per rank at 8 GPUs


def band_rays(n: int, width: int) -> int:
    return int(host_lib().mi_shard_band_rays(n, width))


def rank_count(n: int, band: int, world: int, rank: int) -> int:
    return int(host_lib().mi_shard_count(n, band, world, rank))


def rank_stream_index(n: int, band: int, world: int, rank: int) -> np.ndarray:
    """Stream positions of rank `rank`'s rays, in the order of its own ray stream."""
    out = np.empty(rank_count(n, band, world, rank), dtype=np.uint64)
    _check_host(host_lib().mi_shard_stream_index(n, band, world, rank, out.ctypes.data, out.size))
    return out


def frame_index(n: int, band: int, world: int) -> np.ndarray:
    """Position, in the rank-after-rank gathered buffer, of every ray of the frame's stream."""
    out = np.empty(n, dtype=np.uint64)
    _check_host(host_lib().mi_shard_frame_index(n, band, world, out.ctypes.data))
    return out


def rank_pixels(width: int, height: int, rank: int, world: int):
    """(rows, cols) of the pixels rank `rank` renders, in the order they appear in its ray stream."""
    n = width * height
    idx = rank_stream_index(n, band_rays(n, width), world, rank).astype(np.int64)
    return idx // width, idx % width


def padded_count(width: int, height: int, world: int) -> int:
    """Largest per-rank pixel count: gather needs equal-sized contributions, short ranks pad."""
    n = width * height
    band = band_rays(n, width)
    return max(rank_count(n, band, world, r) for r in range(world))


_index_cache = {}


def _gather_indices(width, height, world, device):
    """For every frame pixel its position in the padded, rank-after-rank gathered tensor (cached: geometry only)."""
    import torch
    key = (width, height, world, str(device))
    if key not in _index_cache:
        n = width * height
        band = band_rays(n, width)
        n_pad = padded_count(width, height, world)
        idx = np.empty(n, dtype=np.int64)
        for r in range(world):
            mine = rank_stream_index(n, band, world, r).astype(np.int64)
            idx[mine] = r * n_pad + np.arange(mine.size)
        _index_cache[key] = torch.from_numpy(idx).to(device)
    return _index_cache[key]


def gather_frame(dist, rgb, width: int, height: int):
    """One collective per frame. `rgb`: this rank's [n_r, 3] float32 torch tensor (any device the
    process group supports). Returns the [height, width, 3] frame on rank 0, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n_pad = padded_count(width, height, world)
    send = rgb
    if rgb.shape[0] != n_pad:
        send = torch.zeros(n_pad, 3, dtype=rgb.dtype, device=rgb.device)
        send[: rgb.shape[0]] = rgb
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, bufs, dst=0)
    if rank != 0:
        return None
    idx = _gather_indices(width, height, world, rgb.device)
    return torch.stack(bufs).reshape(-1, 3)[idx].view(height, width, 3)
