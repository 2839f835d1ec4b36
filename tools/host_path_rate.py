#!/usr/bin/env python3
"""PCIe-inclusive cost of the host-stream entry point mi_render (upload + trace + download) against its parts:

    python tools/host_path_rate.py [edge=1440]

For shadow trace and path trace at 1, 16 and 256 spp prints the wall time of mi_render on (a) ordinary pageable
memory, which the call page-locks for its duration, (b) memory the caller has page-locked already (torch pinned
tensor: used as it is), (c) option pin=0 (pageable copies), whole stream and in 8 pipelined batches; next to them the
kernel time on a device-resident stream and the bare transfer time (H2D + D2H of the stream, pinned), so that
overhead = mi_render - (transfer + kernel) can be read off."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
s = irl.HostScene.builtin("box")


def best_of(fn, reps=4):
    fn()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3


for mode, spp in ((irl.MODE_SHADOW_TRACE, 1), (irl.MODE_PATH_TRACE, 1), (irl.MODE_PATH_TRACE, 16), (irl.MODE_PATH_TRACE, 256)):
    d = s.desc
    d.set_image(edge, edge); d.samples_per_pixel = spp; d.path_trace = 1 if mode == irl.MODE_PATH_TRACE else 0
    dev = irl.IpuScene(d)
    proto = s.init_ray_stream()
    nbytes = proto.nbytes
    # parts: kernel on a resident stream, bare pinned transfers
    raw = torch.from_numpy(proto.view(np.uint8).reshape(proto.size, -1).copy())
    pinned = raw.pin_memory()
    dbuf = raw.cuda()
    st = torch.cuda.current_stream().cuda_stream
    kernel_ms = best_of(lambda: dev.run_device(dbuf.data_ptr(), proto.size, mode, st))
    xfer_ms = best_of(lambda: (dbuf.copy_(pinned, non_blocking=True), pinned.copy_(dbuf, non_blocking=True)))
    line = [f"mode {mode} spp {spp:4d}: kernel {kernel_ms:7.2f} ms  transfer {xfer_ms:6.2f} ms ({2 * nbytes / xfer_ms / 1e6:.0f} GB/s)"]
    for label, batches, pin_opt, prepinned in (("pageable", 1, 1, False), ("caller-pinned", 1, 1, True), ("pin=0", 1, 0, False), ("pageable/8", 8, 1, False), ("caller-pinned/8", 8, 1, True)):
        dev.set_option("pin", pin_opt)
        dev.setRayBatch(0 if batches == 1 else (proto.size + batches - 1) // batches)
        if prepinned:
            host = pinned.numpy().view(irl.TRACE_RESULT).reshape(-1)
        else:
            host = proto.copy()
        ms = best_of(lambda: dev.run(host, mode))
        line.append(f"{label} {ms:7.2f} (+{ms - kernel_ms - xfer_ms:5.2f})")
    print("  ".join(line), flush=True)
    dev.close()
