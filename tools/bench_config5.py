#!/usr/bin/env python3
"""BASELINE config 5 on one GPU at reduced spp: scene 'monkey' (open environment) + NIF environment light
(synthetic weights, the reference's 6 x 320 shape), 1440 x 1440. Prints ms per sample and paths/s."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import torch
import ipu_ray_lib_amd as irl
from bench_nif import weights

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1440
s = irl.HostScene.builtin("monkey"); d = s.desc
d.set_image(size, size); d.samples_per_pixel = spp; d.path_trace = 1
dev = irl.IpuScene(d)
ks, bs, relu, dims = weights(np.random.default_rng(0))
dev.setNif(ks, bs, relu, 12, 3.4299468994140625, np.array([-2.3514461517333984, -2.2660605907440186, -1.9648972749710083], np.float32) - np.float32(1e-8), True)
rays = s.init_ray_stream(); n = rays.size
t = torch.from_numpy(rays.view(np.uint8).reshape(n, irl.TRACE_RESULT.itemsize).copy()).cuda()
st = torch.cuda.current_stream().cuda_stream
dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize(); dev.reset_counters()
t0 = time.perf_counter()
dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize()
el = time.perf_counter() - t0
c = dev.counters()
out = t.cpu().numpy().view(irl.TRACE_RESULT).reshape(-1)
esc = float(((out["h"]["flags"] & irl.FLAG_ESCAPED) != 0).mean())
print(json.dumps({"workload": f"monkey + NIF {size}x{size} x {spp} spp", "ms_per_sample": el / spp * 1e3, "paths_per_s": c["paths"] / el,
                  "casts_per_path": c["casts"] / max(c["paths"], 1), "escaped_fraction_last_sample": esc, "rgb_sum": float(out["rgb"]["x"].sum())}))
import os
if os.environ.get("MI_RAYLIB_FULL_STATS") == "1":
    p = dev.phase_stats(); cyc = p.pop("cycles")
    print("cycle shares:", {k: round(cyc[k] / cyc["total"], 3) for k in ("traverse", "shade", "gen")}, "other", round(1 - (cyc["traverse"] + cyc["shade"] + cyc["gen"]) / cyc["total"], 3))
    for k, v in p.items():
        print(k, "iters per 64 casts %.2f" % (v["iters"] * 64 / c["casts"]), "avg lanes %.1f" % (v["lanes"] / max(v["iters"], 1)))
    print("sum of wave loop cycles", cyc["total"], "per wave (8192 launched, x", spp + 1, "launches)", cyc["total"] / 8192 / (spp + 1), "memtime ticks (100 MHz => us x100)")
