#!/usr/bin/env python3
"""Times path-trace kernel variants / tunings on one frame (device-resident stream, HIP events).

    python tools/k_sweep.py [--scene box] [--size 1440] [--spp 1000] [--reps 2] SPEC [SPEC ...]

SPEC = option=value[:option=value...] (mi_scene_set_option keys), e.g.
    kernel=1   kernel=3:pool_waves=8:pool_tune=4,48,12,8,4,4
Prints one line per spec: ms per frame, casts/s, and whether a 1-in-4099 pixel subsample equals the first spec's."""
import argparse, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="box"); ap.add_argument("--size", type=int, default=1440)
    ap.add_argument("--spp", type=int, default=1000); ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--file", default=None, help="import this scene file (with normals) instead of a built-in scene")
    ap.add_argument("specs", nargs="+")
    a = ap.parse_args()
    s = irl.HostScene.import_file(a.file, load_normals=True) if a.file else irl.HostScene.builtin(a.scene)
    d = s.desc
    d.set_image(a.size, a.size); d.samples_per_pixel = a.spp; d.path_trace = 1
    host = s.init_ray_stream()
    raw = torch.from_numpy(host.view(np.uint8).reshape(host.size, -1).copy())
    buf = raw.cuda()
    stream = torch.cuda.current_stream()
    first = None
    for spec in a.specs:
        # (options of the kernel families outside the shipped library select the variants build of the same sources)
        opts = dict(kv.split("=", 1) for kv in spec.split(":") if "=" in kv)
        dev = irl.IpuScene(d, variants=(opts.get("kernel", "1") not in ("0", "1") or opts.get("spec", "0") != "0" or opts.get("waves", "6") != "6" or opts.get("merge", "1") != "1"
                                        or any(k in opts for k in ("tune", "pool_waves", "pool_tune"))))
        for kv in spec.split(":"):
            k, v = kv.split("=", 1)
            dev.set_option(k, v)
        buf.copy_(raw); torch.cuda.synchronize()
        dev.run_device(buf.data_ptr(), host.size, irl.MODE_PATH_TRACE, stream.cuda_stream); torch.cuda.synchronize()
        sub = buf[::4099].cpu().numpy().tobytes()
        if first is None:
            first = sub
        dev.reset_counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.reps):
            dev.run_device(buf.data_ptr(), host.size, irl.MODE_PATH_TRACE, stream.cuda_stream)
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        c = dev.counters()
        print(f"{spec:60s} {ms:9.2f} ms  {c['casts'] / a.reps / ms / 1e6:8.3f}e9 casts/s  {'same' if sub == first else 'DIFFERENT'}", flush=True)
        dev.close()


if __name__ == "__main__":
    main()
