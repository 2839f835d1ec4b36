"""N>1 path on CPU: two gloo ranks render their round-robin row tiles (with the oracle standing in for
the device — this test is about partitioning, the single gather and re-assembly, not about the kernel),
rank 0 gathers, and the assembled frame must be bit-identical to a one-process render."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
W, H, SPP = 40, 52, 3        # 52 rows: 6 bands of 8 + one of 4 (ragged last band) -> ranks get different pixel counts


def _stream(irl, rows, cols):
    rays = np.zeros(rows.size, dtype=irl.TRACE_RESULT)
    rays["u"] = rows; rays["v"] = cols
    rays["h"]["primID"] = irl.INVALID_PRIM; rays["h"]["geomID"] = irl.INVALID_GEOM
    rays["h"]["normal"]["z"] = 1.0; rays["h"]["r"]["tMax"] = np.inf
    return rays


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import ipu_ray_lib_amd as irl
    import oracle_lib as ol
    from ipu_ray_lib_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = irl.HostScene.builtin("box-simple"); d = s.desc
    d.set_image(W, H); d.samples_per_pixel = SPP
    rows, cols = sharding.rank_pixels(W, H, rank, world)
    rays = _stream(irl, rows, cols)
    ol.path_trace_pixel_rng(d, rays, 2)
    rgb = torch.from_numpy(np.stack([rays["rgb"]["x"], rays["rgb"]["y"], rays["rgb"]["z"]], 1).copy())
    frame = sharding.gather_frame(dist, rgb, W, H)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_band_dealing_is_a_partition_and_matches_an_independent_restatement():
    """The C definition every multi-GPU path shares (csrc/ray_shard.hpp through mi_shard_*): band b of `band`
    consecutive rays belongs to replica b % R. Checked against a numpy restatement for row-structured and
    unstructured streams, ragged last bands, more replicas than bands, and the frame (de-interleave) index."""
    from ipu_ray_lib_amd import sharding
    for n, width in ((40 * 52, 40), (1440 * 24, 1440), (17 * 5, 17), (10001, 0), (4096 * 3 + 5, 7), (5, 5), (1, 1)):
        band = sharding.band_rays(n, width)
        assert band == (8 * width if width and n % width == 0 else 4096)
        for world in (1, 2, 3, 4, 8, 13):
            want_owner = (np.arange(n) // band) % world
            seen = np.zeros(n, int)
            gathered = []
            for r in range(world):
                idx = sharding.rank_stream_index(n, band, world, r).astype(np.int64)
                assert idx.size == sharding.rank_count(n, band, world, r) == int((want_owner == r).sum())
                assert np.array_equal(idx, np.nonzero(want_owner == r)[0])          # a rank's stream keeps stream order
                seen[idx] += 1
                gathered.append(idx)
            assert (seen == 1).all()
            fi = sharding.frame_index(n, band, world).astype(np.int64)
            assert np.array_equal(np.concatenate(gathered)[fi], np.arange(n))     # gathered[fi[i]] is stream ray i


def test_two_rank_tiles_and_single_gather(tmp_path):
    import ipu_ray_lib_amd as irl
    import oracle_lib as ol
    from ipu_ray_lib_amd import sharding
    # partition: every pixel exactly once, for several world sizes
    for world in (1, 2, 3, 4, 8):
        seen = np.zeros((H, W), int)
        for r in range(world):
            rows, cols = sharding.rank_pixels(W, H, r, world)
            seen[rows, cols] += 1
        assert (seen == 1).all()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    out = tmp_path / "frame.npy"
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    got = np.load(out)
    s = irl.HostScene.builtin("box-simple"); d = s.desc
    d.set_image(W, H); d.samples_per_pixel = SPP
    full = s.init_ray_stream(); ol.path_trace_pixel_rng(d, full, 2)
    want = np.stack([full["rgb"]["x"], full["rgb"]["y"], full["rgb"]["z"]], 1).reshape(H, W, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
