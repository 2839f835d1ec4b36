#!/usr/bin/env python3
"""How fast does a persistent launch hand its work units out over its life?

    python tools/launch_progress.py [scene=box] [spp=1000] [size=1440] [period_us=250]

One wave beside the launch (mi_debug_launch_progress) samples the launch's work counter every period; the counter counts work units
(pixel x segment) taken from the queue, 64 at a time. Prints the launch's duration (HIP events), the time the queue ran empty, and the
hand-out rate in twenty equal slices of the launch: the ramp at the start and what is left to do when the queue is empty (the drain)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "box"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 1440
    period_us = int(sys.argv[4]) if len(sys.argv) > 4 else 250
    s = irl.HostScene.builtin(name); d = s.desc
    d.set_image(size, size); d.samples_per_pixel = spp; d.path_trace = 1
    dev = irl.IpuScene(d)
    host = s.init_ray_stream(); n = host.size
    rays = torch.from_numpy(host.view(np.uint8).reshape(n, irl.TRACE_RESULT.itemsize).copy()).cuda()
    st = torch.cuda.current_stream().cuda_stream
    dev.run_device(rays.data_ptr(), n, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dev.run_device(rays.data_ptr(), n, irl.MODE_PATH_TRACE, st); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    seg = 1
    while seg < (spp + 15) // 16 and seg < 64: seg *= 2
    seg = max(seg, 4)
    items = n * ((spp + seg - 1) // seg)
    samples = int(ms * 1.5e3 / period_us) + 64
    buf = torch.zeros(2 * samples, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    dev.launch_progress(buf.data_ptr(), samples, period_us * 100, st)
    e0.record(); dev.run_device(rays.data_ptr(), n, irl.MODE_PATH_TRACE, st); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    a = buf.cpu().numpy().astype(np.uint64).reshape(-1, 2)
    t = (a[:, 0] - a[0, 0]).astype(np.float64) / 100.0 / 1e3          # ms
    c = a[:, 1].astype(np.float64)
    # the launch's own span: from the counter's reset (the sample before the first rise) to its last change
    rise = int(np.argmax(c[1:] > c[:-1])) if (c[1:] > c[:-1]).any() else 0
    # (a stale value of the previous launch stands in the counter until this launch's memset)
    drops = np.nonzero(c[1:] < c[:-1])[0]
    start = int(drops[-1]) + 1 if drops.size else rise
    t0 = t[start]
    last = int(np.nonzero(c[1:] != c[:-1])[0][-1]) + 1
    empty = int(np.argmax(c[start:] >= items)) + start if (c[start:] >= items).any() else last
    print(f"{name} {size}x{size} x {spp} spp: {items} work units of {seg} samples, launch {ms:.2f} ms (HIP events); queue empty {t[empty] - t0:.2f} ms after the first hand-out, "
          f"i.e. {ms - (t[empty] - t0):.2f} ms before the launch ended ({100 * (1 - (t[empty] - t0) / ms):.1f} % of it)")
    span = t[empty] - t0
    edges = np.linspace(0, span, 21)
    cc = np.interp(edges + t0, t, np.minimum(c, items))
    rates = np.diff(cc) / np.diff(edges)
    mean = items / span
    print("hand-out rate per twentieth of the time up to 'queue empty', relative to its mean:")
    print("  " + " ".join(f"{r / mean:5.2f}" for r in rates))
    dev.close()


if __name__ == "__main__":
    main()
