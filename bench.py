#!/usr/bin/env python3
"""bench.py — headline benchmark of the ray-parallel hot path on MI355X.

Metric (BASELINE.json): ray casts per second (CompactBvh::intersect + ::occluded calls/s, whole
job) and ms/frame for the built-in "box" scene, path-trace, 1440x1440 x 1000 spp, defaults of the
reference CLI (max path length 10, roulette start depth 3, AA sigma 0.25 px, seed 1442).

A "step" is one full frame: every pixel's 1000 samples, traced by ONE launch of the path-trace
kernel over a ray stream that is already resident in HBM. With --gpus N (one process per GPU,
launched by torch.distributed.run) the image grows to N x 1440^2 pixels, row-tiles of it are dealt
round-robin to the ranks (rays are the shard; the scene is replicated), and rank 0 collects the
rgb tiles with one RCCL gather at frame end — inside the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_SUMMARY = ROOT / "profiles" / "r01_pmc_hbm_summary.json"   # HBM bytes per launch from the committed rocprofv3 --pmc passes


def image_shape(n_gpus: int, base: int):
    """Weak scaling: the SAME square view at about N x base^2 pixels (edge = base*sqrt(N) rounded to the nearest
    multiple of 8N, so that the 8-row shard bands divide evenly among the ranks): 1440, 2032, 2880, 4096 for
    N = 1, 2, 4, 8. A square frame keeps the image content - and with it casts per path - the same for every N;
    a 2:1 frame of the Cornell box sees mostly empty space beside the box and measures a different workload
    (1.98 instead of 2.97 casts per path, tools/rank_probe.py)."""
    step = 8 * n_gpus
    edge = max(step, int(round(base * (n_gpus ** 0.5) / step)) * step)
    return edge, edge


def make_stream(irl, scene, rows, cols):
    """initPerspectiveRayStream for an arbitrary pixel set: only (u=row, v=col) and rgb=0 matter to
    the path-trace kernel (camera rays are regenerated per sample on the device)."""
    rays = np.zeros(rows.size, dtype=irl.TRACE_RESULT)
    rays["u"] = rows.astype(np.float32)
    rays["v"] = cols.astype(np.float32)
    rays["h"]["primID"] = irl.INVALID_PRIM
    rays["h"]["geomID"] = irl.INVALID_GEOM
    rays["h"]["normal"]["z"] = 1.0
    rays["h"]["r"]["tMax"] = np.inf
    return rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1440, help="image edge per GPU (default = BASELINE config)")
    ap.add_argument("--spp", type=int, default=1000, help="samples per pixel (default = BASELINE config)")
    ap.add_argument("--scene", default="box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import ipu_ray_lib_amd as irl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # Native libraries travel prebuilt; if anything is missing or stale only ONE process per node rebuilds it
    import __graft_entry__ as ge
    if local_rank == 0:
        ge.build_cpu()
        if not (ROOT / "ipu_ray_lib_amd" / "libmi_raylib.so").exists():
            ge.build_device()
    if dist is not None:
        dist.barrier()

    width, height = image_shape(world, args.size)
    scene = irl.HostScene.builtin(args.scene)
    d = scene.desc
    d.set_image(width, height)
    d.samples_per_pixel = args.spp
    d.device = local_rank
    dev = irl.IpuScene(d)

    from ipu_ray_lib_amd import sharding
    rows, cols = sharding.rank_pixels(width, height, rank, world)
    host_rays = make_stream(irl, scene, rows, cols)
    n = host_rays.size
    d_rays = torch.from_numpy(host_rays.view(np.uint8).reshape(n, irl.TRACE_RESULT.itemsize).copy()).cuda()
    stream = torch.cuda.current_stream()

    def gather():
        # the ONE collective of a frame: every rank's rgb tiles to rank 0 over RCCL/xGMI
        if dist is not None:
            rgb = d_rays.view(torch.float32).view(n, 21)[:, 0:3].contiguous()
            return sharding.gather_frame(dist, rgb, width, height)
        return None

    def frame():
        dev.run_device(d_rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream)
        gather()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        frame()
    barrier()
    dev.reset_counters()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record(stream)
        dev.run_device(d_rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream)
        ev[s][1].record(stream)
        gather()
    barrier()
    elapsed = time.perf_counter() - t0
    counters = dev.counters()
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    tot = torch.tensor([elapsed, float(counters["casts"]), float(counters["paths"])], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_casts, total_paths = float(tot[1]), float(tot[2])

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    casts_per_launch = counters["casts"] / max(args.steps, 1)
    paths_per_launch = counters["paths"] / max(args.steps, 1)

    # Algorithmic bytes per cast: 24 B per node visited + the primitive record per
    # leaf test + one 36 B material; per pixel 84 B in + 84 B out once per frame. V and T are measured
    # by the instrumented kernel variant on the same scene/seed at 4 spp (untimed).
    os.environ["MI_RAYLIB_FULL_STATS"] = "1"
    probe_desc = irl.SceneDesc.from_buffer_copy(d)
    probe_desc.samples_per_pixel = 4
    probe = irl.IpuScene(probe_desc)
    os.environ["MI_RAYLIB_FULL_STATS"] = "0"
    sub = slice(0, n, 7)
    probe_rays = host_rays[sub].copy()
    probe.run(probe_rays, irl.MODE_PATH_TRACE)
    pc = probe.counters()
    probe.close()
    nodes_per_cast = pc["nodes_visited"] / max(pc["casts"], 1)
    leaf_per_cast = pc["leaf_tests"] / max(pc["casts"], 1)
    bytes_per_cast = 24.0 * nodes_per_cast + 42.0 * leaf_per_cast + 36.0
    alg_bytes_launch = casts_per_launch * bytes_per_cast + paths_per_launch / args.spp * 168.0
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3 if kernel_ms else float("nan")
    achieved_gbs = alg_bytes_launch / avg_kernel_s / 1e9
    traffic = None
    if PMC_SUMMARY.exists():
        # measured by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on this same command
        # (profiles/), corrected as MI355X_MICROARCH.md prescribes; only valid for the profiled workload
        pm = json.loads(PMC_SUMMARY.read_text())
        if pm.get("workload") == [args.scene, width, height, args.spp, world]:
            traffic = pm.get("hbm_bytes_per_launch")

    out = {
        "metric": "rays/sec (ray casts/s: CompactBvh intersect+occluded calls, whole node), built-in scene 1440x1440 path-trace",
        "value": total_casts / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (built-in Cornell box + monkey bust scene, seeded per-pixel RNG streams)",
        "config": {"workload": f"built-in scene '{args.scene}', path-trace {width}x{height} x {args.spp} spp, max path length 10, "
                               f"roulette depth 3, AA 0.25, seed 1442, {n} pixels on rank 0",
                   "parallelism": f"ray tiles x{world}" + (" + 1 RCCL gather/frame" if world > 1 else "")},
        "paths_per_s": total_paths / elapsed,
        "ms_per_frame": elapsed / max(args.steps, 1) * 1e3,
        "casts_per_path": total_casts / max(total_paths, 1.0),
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "note": "algorithmic scene bytes are served by L1/L2 (the scene is < 1 MB), so this fraction can exceed 1; `traffic` is the HBM traffic of a launch",
                     "kernel": "path_trace_wavefront_kernel", "avg_launch_ms": avg_kernel_s * 1e3,
                     "bytes_per_cast": bytes_per_cast, "nodes_per_cast": nodes_per_cast, "leaf_tests_per_cast": leaf_per_cast},
    }

    if not args.no_cpu_baseline and world == 1:
        import oracle_lib
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, 16))            # the GPU box grants 16 host cores per GPU
        cpu_desc = irl.SceneDesc.from_buffer_copy(d)
        cpu_desc.samples_per_pixel = min(args.spp, 250)
        # calibrate on a coarse pixel grid, then size the sample for about 15 s of CPU work
        rr, cc = np.meshgrid(np.arange(0, height, 48), np.arange(0, width, 48), indexing="ij")
        cal = make_stream(irl, scene, rr.reshape(-1), cc.reshape(-1))
        tc = time.perf_counter()
        oracle_lib.path_trace_pixel_rng(cpu_desc, cal, cores)
        cal_s = max(time.perf_counter() - tc, 1e-3)
        want_px = cal.size * 15.0 / cal_s
        step_px = int(min(48, max(2, round((width * height / want_px) ** 0.5))))
        rr, cc = np.meshgrid(np.arange(0, height, step_px), np.arange(0, width, step_px), indexing="ij")
        cpu_rays = make_stream(irl, scene, rr.reshape(-1), cc.reshape(-1))
        tc = time.perf_counter()
        st = oracle_lib.path_trace_pixel_rng(cpu_desc, cpu_rays, cores)
        cpu_s = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": st.casts / cpu_s, "unit": "rays/s", "cores": cores, "kind": "port",
                               "sample": f"every {step_px}th pixel of the {width}x{height} frame ({cpu_rays.size} pixels) x "
                                         f"{cpu_desc.samples_per_pixel} spp, {st.casts} casts in {cpu_s:.1f} s, oracle/ray_oracle.c with OpenMP"}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
