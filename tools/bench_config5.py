#!/usr/bin/env python3
"""BASELINE config 5: scene 'monkey' (monkey bust in an open environment) + NIF environment light (synthetic weights of
the reference's 6 x 320 shape), 1440 x 1440 x 4000 spp, ray bands dealt to N GPUs.

    python tools/bench_config5.py [spp [size]] [--gpus N] [--devices 0,1,..] [--rehearsal] [--steps K] [--warmup W]

N = 1   one scene, device-resident stream (mi_render_device).
N > 1   the C++ host path mi_group_* in ONE process: a scene replica per device with the NIF model set on EVERY replica
        (the reference streams the weights to every replica, src/IpuScene.cpp:535), the frame dealt in 8-row bands, shares
        resident on their devices; a step = mi_group_trace = every replica runs its {trace slots; MLP; accumulate} loop on
        its share + ONE RCCL send/recv group call to device 0. `--devices` may name an ordinal more than once only with
        `--rehearsal` (several replicas on one GPU: the plumbing, not a scaling measurement).
Prints ONE JSON line: ms per frame, ms per sample, the MLP's share of the GPU time (HIP events round every MLP launch:
scene option "nif_timing"), paths/s, and for N > 1 `one_gpu_same_frame_ms` (the same frame on device 0 alone, same run),
the RCCL message count and the number of distinct devices in the communicator."""
import argparse, json, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import torch
import ipu_ray_lib_amd as irl
from bench_nif import weights

MEAN = np.array([-2.3514461517333984, -2.2660605907440186, -1.9648972749710083], np.float32) - np.float32(1e-8)
MAXV = 3.4299468994140625


def one_gpu(d, scene, ks, bs, relu, steps, warmup, device=0, opts=""):
    d1 = irl.SceneDesc.from_buffer_copy(d); d1.device = device
    torch.cuda.set_device(device)
    # (--opts key=value[:key=value]: scene options of the variants build, for same-box A/B of the trace kernel's forms)
    dev = irl.IpuScene(d1, variants=bool(opts) and any(k.split("=")[0] in ("merge", "waves", "kernel", "spec", "tune") for k in opts.split(":"))).set_option("nif_timing", 1)
    for kv in filter(None, opts.split(":")):
        dev.set_option(*kv.split("=", 1))
    dev.setNif(ks, bs, relu, 12, MAXV, MEAN, True)
    rays = scene.init_ray_stream(); n = rays.size
    t = torch.from_numpy(rays.view(np.uint8).reshape(n, irl.TRACE_RESULT.itemsize).copy()).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st)
    torch.cuda.synchronize(); dev.reset_counters(); dev.nif_timing()
    t0 = time.perf_counter()
    for _ in range(steps):
        dev.run_device(t.data_ptr(), n, irl.MODE_PATH_TRACE, st)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    c = dev.counters(); tm = dev.nif_timing()
    out = t.cpu().numpy().view(irl.TRACE_RESULT).reshape(-1)
    esc = float(((out["h"]["flags"] & irl.FLAG_ESCAPED) != 0).mean())
    res = {"ms_per_frame": el * 1e3, "mlp_ms_per_frame": tm["mlp_ms"] / steps, "mlp_launches_per_frame": tm["launches"] // steps,
           "paths": c["paths"] // steps, "casts": c["casts"] // steps, "escaped_fraction_last_sample": esc, "rgb_sum": float(out["rgb"]["x"].sum())}
    if os.environ.get("MI_RAYLIB_FULL_STATS") == "1":
        p = dev.phase_stats(); cyc = p.pop("cycles")
        print("cycle shares:", {k: round(cyc[k] / cyc["total"], 3) for k in ("traverse", "shade", "gen")}, file=sys.stderr)
        wc = c["casts"] / 64.0
        print("turns per 64 casts / occupancy:", {k: (round(p[k]["iters"] / wc, 2), round(p[k]["lanes"] / (64.0 * max(p[k]["iters"], 1)), 3)) for k in ("node", "leaf", "shade", "gen")}, file=sys.stderr)
    dev.close()
    return res


def main():
    # (stdout carries the one JSON line: RCCL prints its version banner on stdout when a communicator is created)
    sys.stdout.flush(); real_stdout = os.fdopen(os.dup(1), "w"); os.dup2(2, 1); sys.stdout = sys.stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("spp", nargs="?", type=int, default=4000)
    ap.add_argument("size", nargs="?", type=int, default=1440)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--devices", default="")
    ap.add_argument("--rehearsal", action="store_true", help="allow repeated ordinals in --devices (replicas sharing a GPU)")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--opts", default="", help="N = 1: scene options of the variants build, key=value[:key=value] (e.g. merge=0)")
    a = ap.parse_args()
    s = irl.HostScene.builtin("monkey"); d = s.desc
    d.set_image(a.size, a.size); d.samples_per_pixel = a.spp; d.path_trace = 1
    ks, bs, relu, dims = weights(np.random.default_rng(0))
    flops_per_ray = 2 * sum(k * c for k, c in dims)
    out = {"workload": f"monkey + NIF {a.size}x{a.size} x {a.spp} spp (BASELINE config 5{'' if (a.size, a.spp) == (1440, 4000) else ', reduced'})",
           "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup}
    if a.gpus == 1:
        r = one_gpu(d, s, ks, bs, relu, a.steps, a.warmup, opts=a.opts)
        out.update({"launch": "single", "opts": a.opts, "ms_per_frame": r["ms_per_frame"], "ms_per_sample": r["ms_per_frame"] / a.spp,
                    "mlp_share": r["mlp_ms_per_frame"] / r["ms_per_frame"], "mlp_ms_per_frame": r["mlp_ms_per_frame"],
                    "mlp_launches_per_frame": r["mlp_launches_per_frame"], "paths_per_s": r["paths"] / (r["ms_per_frame"] * 1e-3),
                    "casts_per_path": r["casts"] / max(r["paths"], 1), "escaped_fraction_last_sample": r["escaped_fraction_last_sample"],
                    "rgb_sum": r["rgb_sum"], "mlp_flops_per_ray": flops_per_ray})
        real_stdout.write(json.dumps(out) + "\n"); real_stdout.flush(); return
    devices = [int(x) for x in a.devices.split(",")] if a.devices else list(range(a.gpus))
    if len(devices) != a.gpus:
        raise SystemExit(f"bench_config5.py: --devices names {len(devices)} replicas, --gpus {a.gpus}")
    if len(set(devices)) != len(devices) and not a.rehearsal:
        raise SystemExit("bench_config5.py: --devices repeats an ordinal: replicas sharing a GPU measure the plumbing, not scaling; add --rehearsal if that is meant")
    try:
        grp = irl.IpuGroup(d, devices, irl.TRANSPORT_RCCL)
    except irl.RaylibError as e:
        raise SystemExit(f"bench_config5.py --gpus {a.gpus} ({torch.cuda.device_count()} GPU(s) visible): {e}")
    grp.set_option("nif_timing", 1)
    grp.setNif(ks, bs, relu, 12, MAXV, MEAN, True)
    host = s.init_ray_stream()
    grp.upload(host)
    for _ in range(a.warmup):
        grp.trace(irl.MODE_PATH_TRACE)
    grp.reset_counters()
    for sc in grp.scenes():
        sc.nif_timing()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        grp.trace(irl.MODE_PATH_TRACE)
        step_ms.append(grp.getTraceTimeSecs() * 1e3)
    el = (time.perf_counter() - t0) / a.steps
    c = grp.counters(); moved = grp.last_transfer()
    mlp = [sc.nif_timing()["mlp_ms"] / a.steps for sc in grp.scenes()]
    got = grp.download(host.copy())
    grp.close()
    anchor = one_gpu(d, s, ks, bs, relu, 1, 1, device=devices[0])
    out.update({"launch": "group", "devices": devices, "rccl_ranks": len(set(devices)), "rehearsal": len(set(devices)) != len(devices),
                "ms_per_frame": el * 1e3, "ms_per_sample": el * 1e3 / a.spp, "step_ms": step_ms,
                "mlp_ms_per_frame_per_replica": mlp, "mlp_share": max(mlp) / (el * 1e3) if len(set(devices)) == len(devices) else sum(mlp) / (el * 1e3),
                "paths_per_s": c["paths"] / a.steps / el, "casts_per_path": c["casts"] / max(c["paths"], 1),
                "gather": {"rccl_messages": moved["rccl_messages"], "peer_copies": moved["peer_copies"]},
                "one_gpu_same_frame_ms": anchor["ms_per_frame"], "one_gpu_mlp_share": anchor["mlp_ms_per_frame"] / anchor["ms_per_frame"],
                "rgb_sum": float(got["rgb"]["x"].sum()), "one_gpu_rgb_sum_after_two_frames": anchor["rgb_sum"], "mlp_flops_per_ray": flops_per_ray})
    real_stdout.write(json.dumps(out) + "\n"); real_stdout.flush()


if __name__ == "__main__":
    main()
