#!/usr/bin/env python3
"""CPU-baseline rows of BASELINE.md: the oracle (the CPU twin of renderCPU / traceShadowRay) timed on this box's host cores,
next to the GPU on the same inputs.  python tools/cpu_twin_rates.py  -> JSON lines."""
import json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ipu_ray_lib_amd as irl
import oracle_lib as ol

cores = min(len(os.sched_getaffinity(0)), 16)
s = irl.HostScene.builtin("box"); d = s.desc
# config 1: shadow trace 512^2
d.set_image(512, 512); d.path_trace = 0
r = s.init_ray_stream(); t0 = time.perf_counter(); st = ol.shadow_trace(d, r, cores); cpu = time.perf_counter() - t0
dev = irl.IpuScene(d); g = s.init_ray_stream(); dev.run(g, irl.MODE_SHADOW_TRACE)
g = s.init_ray_stream(); t0 = time.perf_counter(); dev.run(g, irl.MODE_SHADOW_TRACE); gpu = time.perf_counter() - t0
print(json.dumps({"config": 1, "workload": "box shadow-trace 512x512", "cores": cores, "cpu_ms": cpu * 1e3, "cpu_casts_per_s": st.casts / cpu, "cpu_rays_per_s": r.size / cpu,
                  "gpu_host_path_ms": gpu * 1e3, "gpu_casts_per_s_host_path": dev.counters()["casts"] / 2 / gpu, "identical": bool(g.tobytes() == r.tobytes())}))
dev.close()
# config 2 at reduced spp: per-pixel streams (scales with cores) and the reference-faithful shared generator (sequential)
d.set_image(1440, 1440); d.path_trace = 1; d.samples_per_pixel = 16
sub = s.init_ray_stream()[::16].copy(); t0 = time.perf_counter(); st = ol.path_trace_pixel_rng(d, sub, cores); cpu = time.perf_counter() - t0
print(json.dumps({"config": 2, "workload": "box path-trace 1440x1440, every 16th pixel x 16 spp, per-pixel RNG streams", "cores": cores, "cpu_s": cpu,
                  "cpu_casts_per_s": st.casts / cpu, "cpu_paths_per_s": st.paths / cpu, "extrapolated_ms_per_1000spp_frame": cpu * 16 * 1000 / 16 * 1e3}))
d.set_image(360, 360); d.samples_per_pixel = 4
one = s.init_ray_stream(); t0 = time.perf_counter(); st = ol.path_trace_shared_rng(d, one); cpu = time.perf_counter() - t0
print(json.dumps({"config": 2, "workload": "box path-trace 360x360 x 4 spp, ONE shared generator consumed sequentially (literal renderCPU)", "cores": 1, "cpu_s": cpu,
                  "cpu_casts_per_s": st.casts / cpu, "cpu_paths_per_s": st.paths / cpu}))
