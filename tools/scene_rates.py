#!/usr/bin/env python3
"""Casts/s of the path-trace kernel on every scene the repository ships (1024^2 x 64 spp, one GPU):
a guard against scheduling defaults that only suit the headline scene."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl

def rate(scene, name, size=1024, spp=64):
    d = scene.desc
    d.set_image(size, size); d.samples_per_pixel = spp; d.path_trace = 1
    dev = irl.IpuScene(d)
    rays = scene.init_ray_stream(); n = rays.size
    t = torch.from_numpy(rays.view(np.uint8).reshape(n, irl.TRACE_RESULT.itemsize).copy()).cuda()
    st = torch.cuda.current_stream().cuda_stream
    dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize(); dev.reset_counters()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    c = dev.counters(); dev.close()
    print(f"{name:12s} nodes {d.num_nodes:6d}  casts/s {c['casts'] / el:.3e}  paths/s {c['paths'] / el:.3e}  casts/path {c['casts'] / c['paths']:.2f}")

for name in ("box-simple", "box", "spheres", "monkey"):
    rate(irl.HostScene.builtin(name), name)
rate(irl.HostScene.import_file(ROOT / "assets" / "test_scene.dae", load_normals=True), "test_scene")
if len(sys.argv) > 1:
    print("-- steady state,", sys.argv[1], "spp")
    rate(irl.HostScene.builtin("box"), "box", 1440, int(sys.argv[1]))
    rate(irl.HostScene.import_file(ROOT / "assets" / "test_scene.dae", load_normals=True), "test_scene", 1440, int(sys.argv[1]))
    rate(irl.HostScene.builtin("monkey"), "monkey", 1440, int(sys.argv[1]))
