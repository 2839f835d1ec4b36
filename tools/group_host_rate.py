#!/usr/bin/env python3
"""Host-stream cost of the multi-replica path mi_group_render on ONE GPU (replicas share device 0; RCCL transport):
upload (one strided copy per replica), trace, one RCCL group call, download (one strided copy per replica) against
the single-scene mi_render of the same stream - what the dealing, the gather and the strided copies cost.

    python tools/group_host_rate.py [edge=2880] [spp=16] [replicas=8]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 2880
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
R = int(sys.argv[3]) if len(sys.argv) > 3 else 8
s = irl.HostScene.builtin("box"); d = s.desc
d.set_image(edge, edge); d.samples_per_pixel = spp; d.path_trace = 1
proto = s.init_ray_stream()


def best_of(fn, reps=3):
    fn()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3


pinned = torch.from_numpy(proto.view(np.uint8).reshape(proto.size, -1).copy()).pin_memory()
host = pinned.numpy().view(irl.TRACE_RESULT).reshape(-1)
one = irl.IpuScene(d)
ms_one = best_of(lambda: one.run(host, irl.MODE_PATH_TRACE))
one.close()
print(f"{edge}x{edge} x {spp} spp, {proto.nbytes / 1e6:.0f} MB stream (caller-pinned): single scene mi_render {ms_one:.2f} ms", flush=True)
for transport, name in ((irl.TRANSPORT_RCCL, "rccl"), (irl.TRANSPORT_COPY, "peer copies")):
    grp = irl.IpuGroup(d, [0] * R, transport)
    ms = best_of(lambda: grp.run(host, irl.MODE_PATH_TRACE))
    mv = grp.last_transfer()
    # the stages one by one
    t0 = time.perf_counter(); grp.upload(host); t_up = (time.perf_counter() - t0) * 1e3
    grp.trace(irl.MODE_PATH_TRACE)
    t0 = time.perf_counter(); grp.trace(irl.MODE_PATH_TRACE); t_tr = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); grp.download(host); t_dn = (time.perf_counter() - t0) * 1e3
    print(f"  {R} replicas on one GPU, {name}: mi_group_render {ms:.2f} ms ({mv}); stages: upload {t_up:.2f} ms ({proto.nbytes / t_up / 1e6:.0f} GB/s), "
          f"trace + gather {t_tr:.2f} ms, download {t_dn:.2f} ms ({proto.nbytes / t_dn / 1e6:.0f} GB/s)", flush=True)
    grp.close()
