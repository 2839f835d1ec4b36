#!/usr/bin/env python3
"""How much of a 1440^2 frame is drain tail? The same pixel stream is rendered as ONE launch of k concatenated copies
(same rays, same coherence, k x the work items per lane), and casts/s compared."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 250
s = irl.HostScene.builtin("box"); d = s.desc
d.set_image(1440, 1440); d.samples_per_pixel = spp; d.path_trace = 1
dev = irl.IpuScene(d)
base = s.init_ray_stream()
st = torch.cuda.current_stream().cuda_stream
for k in (1, 2, 4, 8):
    rays = np.concatenate([base] * k)
    n = rays.size
    t = torch.from_numpy(rays.view(np.uint8).reshape(n, irl.TRACE_RESULT.itemsize).copy()).cuda()
    dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize(); dev.reset_counters()
    t0 = time.perf_counter()
    dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    c = dev.counters()
    print(f"{k} copies: {n} rays, {c['casts'] / el:.4g} casts/s, {el * 1e3 / k:.1f} ms per 1440^2 frame")
