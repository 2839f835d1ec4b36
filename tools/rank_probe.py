#!/usr/bin/env python3
"""Times ONE rank's share of the N-GPU weak-scaling frame on a single GPU (no process group): shows how much of a
multi-GPU throughput difference is image content (aspect ratio) rather than scaling. Usage: rank_probe.py N [rank]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl
from ipu_ray_lib_amd import sharding
import bench

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 200
w, h = bench.image_shape(world, 1440)
scene = irl.HostScene.builtin("box"); d = scene.desc
d.set_image(w, h); d.samples_per_pixel = spp
dev = irl.IpuScene(d)
rows, cols = sharding.rank_pixels(w, h, rank, world)
rays = bench.make_stream(irl, scene, rows, cols)
n = rays.size
t = torch.from_numpy(rays.view(np.uint8).reshape(n, irl.TRACE_RESULT.itemsize).copy()).cuda()
st = torch.cuda.current_stream().cuda_stream
dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize(); dev.reset_counters()
t0 = time.perf_counter()
for _ in range(3):
    dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st)
torch.cuda.synchronize()
el = time.perf_counter() - t0
c = dev.counters()
print(f"world {world} rank {rank}: {w}x{h}, {n} pixels, casts/s {c['casts'] / el:.4g}, casts/path {c['casts'] / c['paths']:.3f}, ms/frame {el / 3 * 1e3:.1f}")
