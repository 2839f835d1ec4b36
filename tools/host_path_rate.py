#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-stream entry point mi_render (upload + trace + download), per render mode:

    python tools/host_path_rate.py [edge=1440]

Prints wall time and rays/s for shadow trace and for path trace at 1, 16 and 256 spp, whole stream and in 8 batches.
Set MI_RAYLIB_PIN=0 to see the pageable-copy route."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ipu_ray_lib_amd as irl

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
s = irl.HostScene.builtin("box")
for mode, spp in ((irl.MODE_SHADOW_TRACE, 1), (irl.MODE_PATH_TRACE, 1), (irl.MODE_PATH_TRACE, 16), (irl.MODE_PATH_TRACE, 256)):
    d = s.desc
    d.set_image(edge, edge); d.samples_per_pixel = spp; d.path_trace = 1 if mode == irl.MODE_PATH_TRACE else 0
    dev = irl.IpuScene(d)
    for batches in (1, 8):
        rays = s.init_ray_stream()
        dev.setRayBatch(0 if batches == 1 else (rays.size + batches - 1) // batches)
        dev.run(rays, mode)                      # warm-up (allocations, code load)
        best = 1e9
        for _ in range(3):
            rays = s.init_ray_stream()
            t0 = time.perf_counter(); dev.run(rays, mode); best = min(best, time.perf_counter() - t0)
        print(f"mode {mode} spp {spp:4d} batches {batches}: {best * 1e3:8.2f} ms  {rays.size / best:.3e} pixels/s  "
              f"{2 * rays.nbytes / best / 1e9:.1f} GB/s over PCIe")
    dev.close()
