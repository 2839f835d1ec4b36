#!/usr/bin/env python3
"""bench.py — headline benchmark of the ray-parallel hot path on MI355X.

Metric (BASELINE.json): ray casts per second (CompactBvh::intersect + ::occluded calls/s, whole
job) and ms/frame for the built-in "box" scene, path-trace, defaults of the reference CLI (max path
length 10, roulette start depth 3, AA sigma 0.25 px, seed 1442).

A "step" is one full frame: every pixel's 1000 samples, traced by ONE launch of the path-trace
kernel over a ray stream that is already resident in HBM.
  N = 1   BASELINE config 2: 1440x1440 x 1000 spp.
  N > 1   BASELINE config 4: 2880x2880 x 1000 spp, the frame's 8-row bands dealt round-robin to the GPUs (rays are the
          shard, the scene is replicated), no exchange while the frame renders, and ONE RCCL gather at frame end -
          inside the timed region. Two launch forms, the same dealing (csrc/ray_shard.hpp) behind both:
            "launch": "ranks"  under torch.distributed.run (WORLD_SIZE = N): one process per GPU, rank 0 gathers the
                               rgb tiles with one dist.gather;
            "launch": "group"  started as a plain `python bench.py --gpus N --launch group`: ONE process drives
                               the N devices through the C++ host path mi_group_* (what `trace --gpus N` runs): shares
                               resident on their devices, timed region = K x mi_group_trace (trace + one RCCL
                               send/recv group call of the full TraceResults to device 0).
          The frame is the same for N = 2, 4, 8 ("scaling": "strong"); --weak renders N x 1440^2 pixels instead.
          For N > 1 the line also carries "one_gpu_same_frame_ms": the same frame on one device alone, measured in
          the same run, as the anchor of the scaling curve; "rccl_ranks" (distinct devices in the communicator, with the
          backend's name) and per-step "gather_ms", so that a record says by itself how many devices it ran on. The group
          launch has not yet run on two physical devices: it is only taken when asked for by name, and `--devices` with a
          repeated ordinal (replicas sharing a GPU) needs `--rehearsal`.

After the timed loop (outside it) rank 0 at N = 1
  * copies ~2 000 pixels of the LAST timed frame back and compares all 84 bytes of each with the CPU oracle run over
    the same number of frames at the full sample count (`parity_checked_pixels`, `parity_mismatches`; a mismatch
    makes the exit code 1),
  * measures the numbers of the `roofline` record with the instrumented kernel build on the same frame (nodes and
    primitive tests per cast, lanes active per phase), the NIF MLP kernel (K3) on 1440^2 rays and the 16-spp
    preview frame,
  * times the CPU oracle on a bounded pixel sample (`cpu_baseline`).

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

# MI355X_MICROARCH.md: HBM3E 8 TB/s spec; L2 ~34.5 TB/s aggregate; ~2.5 PFLOP/s dense f16 MFMA. (No vector-L1 figure there: K1w's roof is measured, see `roofline`.)
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
MFMA_F16_PEAK_TFLOPS = 2500.0
PMC_SUMMARY = ROOT / "profiles" / "r05_pmc_summary.json"   # rocprofv3 --pmc passes of this same command (tools/prof_r05.sh + tools/collect_profiles.py)
GATHER_PROBE = ROOT / "profiles" / "r05_gather_probe.json"  # tools/gather_probe.py on the same scene (fallback when the probe cannot run in this process)


def image_shape(n_gpus: int, base: int, weak: bool):
    """N = 1: base x base (config 2). N > 1: config 4's 2 base x 2 base frame for every N (strong scaling), or with
    --weak the same square view at about N x base^2 pixels (edge = base*sqrt(N) rounded to a multiple of 8N: 1440,
    2032, 2880, 4096). Square frames keep the image content - and with it casts per path - the same for every N."""
    if n_gpus == 1:
        return base, base
    if not weak:
        return 2 * base, 2 * base
    step = 8 * n_gpus
    edge = max(step, int(round(base * (n_gpus ** 0.5) / step)) * step)
    return edge, edge


def make_stream(irl, rows, cols):
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


def to_device(torch, irl, host_rays):
    return torch.from_numpy(host_rays.view(np.uint8).reshape(host_rays.size, irl.TRACE_RESULT.itemsize).copy()).cuda()


def time_launches(torch, fn, reps, stream, ahead=0):
    """Average duration of `fn` (enqueues on `stream`) over `reps` calls, HIP events on that stream. `ahead` more calls run before
    the first event WITHOUT a wait in between (millisecond launches: a device that went idle behind a synchronize takes about a
    millisecond to get its clocks back, which a 20 x 1.5 ms region would book as 4 % of every launch - tools/bench_nif.py --reps
    5 / 20 / 100 / 400 without them: 1.76 / 1.61 / 1.55 / 1.54 ms on one box; launches in a render follow each other like these)."""
    fn(); torch.cuda.synchronize()
    for _ in range(ahead):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def nif_weights(rng, hidden=320, embed=12, layers=6):
    """Synthetic weights of the reference's shapes: 48->320->320->320->(320+48)->320->320->3 (nif_metadata.txt)."""
    F = 4 * embed
    dims = [(F, hidden)] + [((hidden + F) if l == layers // 2 else hidden, hidden) for l in range(1, layers)] + [(hidden, 3)]
    ks = [(rng.normal(size=d) * np.sqrt(2.0 / d[0])).astype(np.float32) for d in dims]
    bs = [(rng.normal(size=d[1]) * 0.05).astype(np.float32) for d in dims]
    return ks, bs, [1] * (len(dims) - 1) + [0], dims


MEAN_NIF = np.array([-2.3514461517333984, -2.2660605907440186, -1.9648972749710083], np.float32) - np.float32(1e-8)     # nif_metadata.txt: mean - eps
MAX_NIF = 3.4299468994140625


def config3_record(torch, irl, stream, spp=1000):
    """BASELINE config 3 (assets/test_scene.dae --load-normals, 1440 x 1440 x 4000 spp) at `spp` samples: one warm-up + one timed
    launch on a device-resident stream; from 1000 samples on a pixel's work units are 64-sample segments, so the
    frame time scales linearly in the sample count and the 4000-spp figure is quoted as an extrapolation, marked so."""
    s = irl.HostScene.import_file(ROOT / "assets" / "test_scene.dae", load_normals=True)
    d = s.desc
    d.set_image(1440, 1440); d.samples_per_pixel = spp; d.path_trace = 1
    dev = irl.IpuScene(d)
    rays = to_device(torch, irl, s.init_ray_stream()); n = 1440 * 1440
    ms = time_launches(torch, lambda: dev.run_device(rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream), 1, stream)
    c = dev.counters()            # two launches (warm-up + timed)
    dev.close()
    return {"workload": f"assets/test_scene.dae --load-normals, 1440x1440 x {spp} spp (BASELINE config 3 at {spp} of its 4000 samples)",
            "ms_per_frame": ms, "ms_per_sample": ms / spp, "extrapolated_ms_4000spp": ms / spp * 4000.0,
            "casts_per_s": c["casts"] / 2.0 / (ms * 1e-3), "paths_per_s": c["paths"] / 2.0 / (ms * 1e-3), "casts_per_path": c["casts"] / max(c["paths"], 1)}


def config5_record(torch, irl, stream, cores, spp=1024, check=True):
    """BASELINE config 5 (monkey bust + NIF environment, synthetic weights of the reference's 6 x 320 shape, 1440 x 1440 x 4000
    spp) at `spp` samples on one GPU (1 024: the 64-sample work atoms and the 512-sample launches of the 4000-spp frame, two of its
    eight launches): one warm-up frame, one timed frame on a fresh stream; the MLP's share from HIP events round
    every MLP launch (scene option nif_timing); rates as trace.cpp:328-333 defines them (paths/s = pixels x spp / s). Parity of
    the TIMED frame: every 4099th pixel against the oracle's NIF render - hit records bit for bit, rgb within the MLP's stated
    tolerance (tests/test_gpu_parity.py::test_config5_monkey_nif_1440_x_256spp_against_oracle)."""
    ks, bs, relu, dims = nif_weights(np.random.default_rng(0))
    s = irl.HostScene.builtin("monkey"); d = s.desc
    d.set_image(1440, 1440); d.samples_per_pixel = spp; d.path_trace = 1
    dev = irl.IpuScene(d).set_option("nif_timing", 1)
    dev.setNif(ks, bs, relu, 12, MAX_NIF, MEAN_NIF, True)
    host = s.init_ray_stream(); n = host.size
    warm = to_device(torch, irl, host)
    dev.run_device(warm.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream)
    torch.cuda.synchronize(); del warm
    dev.reset_counters(); dev.nif_timing()
    rays = to_device(torch, irl, host)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    dev.run_device(rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = dev.counters(); tm = dev.nif_timing()
    out = {"workload": f"built-in scene 'monkey' + NIF environment (synthetic 6 x 320 weights), 1440x1440 x {spp} spp (BASELINE config 5 at {spp} of its 4000 samples), one GPU",
           "ms_per_frame": ms, "ms_per_sample": ms / spp, "extrapolated_ms_4000spp": ms / spp * 4000.0,
           "mlp_share": tm["mlp_ms"] / ms, "mlp_ms_per_frame": tm["mlp_ms"], "mlp_launches": tm["launches"],
           "paths_per_s": c["paths"] / (ms * 1e-3), "casts_per_path": c["casts"] / max(c["paths"], 1),
           "mlp_shader_clock_ghz": dev.nif_clock_ghz(), "mlp_flops_per_ray": 2 * sum(k * q for k, q in dims)}
    if check:
        import oracle_lib as ol
        idx = np.arange(0, n, 4099)
        got = np.frombuffer(rays[torch.from_numpy(idx).cuda()].cpu().numpy().tobytes(), dtype=irl.TRACE_RESULT)
        want = host[idx].copy()
        nif, keep = ol.make_nif(ks, bs, relu, 12, MAX_NIF, MEAN_NIF, True, half_features=True, half_weights_acts=True)
        st = ol.Stats()
        tp = time.perf_counter()
        ol.lib().o_path_trace_nif_pixel_rng(C.byref(d), C.byref(nif), 0.0, want.ctypes.data, want.size, cores, C.byref(st))
        hg = np.ascontiguousarray(got["h"]).view(np.uint8).reshape(got.size, -1); hw = np.ascontiguousarray(want["h"]).view(np.uint8).reshape(want.size, -1)
        hit_bad = int((hg != hw).any(axis=1).sum())
        g = np.stack([got["rgb"][k] for k in "xyz"], 1).astype(np.float64); w = np.stack([want["rgb"][k] for k in "xyz"], 1).astype(np.float64)
        err = np.abs(g - w) / (np.abs(w) + 0.05 * spp)
        q995, worst = float(np.quantile(err, 0.995)), float(err.max())
        out["parity"] = {"checked_pixels": int(idx.size), "hit_record_mismatches": hit_bad, "rgb_rel_err_q995": q995, "rgb_rel_err_max": worst,
                         "rgb_tolerance": "q99.5 < 0.01 and max < 0.1 of |want| + 0.05 spp (the MLP's 2 % / 10 % per-sample tolerance averaged over a pixel's samples)",
                         "ok": bool(hit_bad == 0 and q995 < 0.01 and worst < 0.1),
                         "note": f"every 4099th pixel of the timed frame vs oracle/ray_oracle.c's NIF render ({time.perf_counter() - tp:.1f} s): 64-byte hit records bit for bit, rgb sums within the tolerance"}
    dev.close()
    return out


def one_gpu_same_frame_ms(torch, irl, desc, width, height, device):
    """The N > 1 frame on ONE GPU (device `device`), device-resident stream, one warm-up + two timed launches: the
    same-frame anchor of the scaling curve, measured in the same run."""
    d1 = irl.SceneDesc.from_buffer_copy(desc)
    d1.device = device
    torch.cuda.set_device(device)
    dev = irl.IpuScene(d1)
    rows, cols = np.divmod(np.arange(width * height, dtype=np.int64), width)
    rays = to_device(torch, irl, make_stream(irl, rows, cols))
    st = torch.cuda.current_stream()
    ms = time_launches(torch, lambda: dev.run_device(rays.data_ptr(), width * height, irl.MODE_PATH_TRACE, st.cuda_stream), 2, st)
    dev.close()
    return ms


def bench_group(args, torch, irl):
    """--gpus N > 1 in ONE process: the C++ host path (mi_group_*, csrc/group_render.hpp). One scene replica per device
    0..N-1, the frame dealt in 8-row bands, shares resident on their devices; a step = mi_group_trace = every replica
    traces its share + ONE RCCL group call gathers the shares on device 0."""
    n_gpus = args.gpus
    width, height = image_shape(n_gpus, args.size, args.weak)
    scene = irl.HostScene.builtin(args.scene)
    d = scene.desc
    d.set_image(width, height)
    d.samples_per_pixel = args.spp
    d.path_trace = 1
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(n_gpus))
    if len(devices) != n_gpus:
        raise SystemExit(f"bench.py: --devices names {len(devices)} replicas, --gpus {n_gpus}")
    if len(set(devices)) != len(devices) and not args.rehearsal:
        raise SystemExit("bench.py: --devices repeats an ordinal: replicas that share a GPU rehearse the plumbing, they do not measure scaling. "
                         "Add --rehearsal if that is what is meant (the line is then marked as one).")
    try:
        grp = irl.IpuGroup(d, devices, irl.TRANSPORT_RCCL)
    except irl.RaylibError as e:
        raise SystemExit(f"bench.py --gpus {n_gpus} (single-process group path, {torch.cuda.device_count()} GPU(s) visible): {e}")
    n = width * height
    host_rays = scene.init_ray_stream()
    grp.upload(host_rays)
    for _ in range(args.warmup):
        grp.trace(irl.MODE_PATH_TRACE)
    grp.reset_counters()
    for dev_i in sorted(set(devices)):
        torch.cuda.synchronize(dev_i)
    step_ms, gather_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        grp.trace(irl.MODE_PATH_TRACE)            # returns when every replica's stream and the gather have drained
        step_ms.append(grp.getTraceTimeSecs() * 1e3)
        gather_ms.append(grp.last_gather_ms())    # HIP events on the root's stream round the group call (already complete: no wait)
    for dev_i in sorted(set(devices)):
        torch.cuda.synchronize(dev_i)
    elapsed = time.perf_counter() - t0
    c = grp.counters()
    moved = grp.last_transfer()
    out = {
        "metric": "rays/sec (ray casts/s: CompactBvh intersect+occluded calls, whole node), built-in scene 1440x1440 path-trace",
        "value": c["casts"] / elapsed, "unit": "rays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (built-in Cornell box + monkey bust scene, seeded per-pixel RNG streams)",
        "config": {"workload": f"built-in scene '{args.scene}', path-trace {width}x{height} x {args.spp} spp, max path length 10, "
                               f"roulette depth 3, AA 0.25, seed 1442 (BASELINE config {'4' if not args.weak else '2, weak-scaled frame'})",
                   "parallelism": f"ray bands x{n_gpus}, one process (mi_group_*) on devices {devices}, 1 RCCL send/recv group call per frame"},
        "paths_per_s": c["paths"] / elapsed, "ms_per_frame": elapsed / max(args.steps, 1) * 1e3,
        "casts_per_path": c["casts"] / max(c["paths"], 1), "launch": "group",
        "rccl_ranks": {"distinct_devices": len(grp.devices()), "devices": grp.devices(), "backend": "rccl (ncclCommInitAll, one communicator per distinct device)",
                       "replicas": n_gpus, "rehearsal": len(grp.devices()) != n_gpus},
        "step_ms": step_ms, "gather_ms": gather_ms, "gather": {"rccl_messages": moved["rccl_messages"], "peer_copies": moved["peer_copies"],
                                        "bytes_to_root": int((n - grp.gathered_device()[1][1]) * irl.TRACE_RESULT.itemsize)},
    }
    rc = 0
    if not args.no_cpu_baseline:
        import oracle_lib
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        got = grp.download(host_rays.copy())
        frames = args.warmup + args.steps
        stride = max(1, n // 1009)
        want = scene.init_ray_stream()[::stride].copy()
        tp = time.perf_counter()
        for _ in range(frames):
            oracle_lib.path_trace_pixel_rng(d, want, cores)
        gb = got[::stride].copy().view(np.uint8).reshape(want.size, -1); wb = want.view(np.uint8).reshape(want.size, -1)
        bad = int((gb != wb).any(axis=1).sum())
        out["parity_checked_pixels"] = int(want.size); out["parity_mismatches"] = bad
        out["parity_note"] = (f"every {stride}th pixel of the gathered frame after {frames} accumulated frames, all 84 bytes, vs "
                              f"oracle/ray_oracle.c ({time.perf_counter() - tp:.1f} s)")
        rc = 1 if bad else 0
    grp.close()
    out["one_gpu_same_frame_ms"] = one_gpu_same_frame_ms(torch, irl, d, width, height, 0)
    out["roofline"] = {"kernel": "path_trace_wavefront_kernel", "note": "see the N = 1 line: the per-GPU kernel is the same launch on a share of the frame"}
    emit(out)
    if rc:
        sys.exit(rc)


def claim_stdout():
    """stdout carries exactly ONE JSON line. Native libraries write there too (RCCL prints its version banner on stdout when
    a communicator is created), so file descriptor 1 is pointed at stderr for the run and the line goes to the real stdout,
    kept aside here."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    return real


REAL_STDOUT = None


def emit(out):
    REAL_STDOUT.write(json.dumps(out) + "\n")
    REAL_STDOUT.flush()


def main():
    global REAL_STDOUT
    REAL_STDOUT = claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1440, help="image edge of the 1-GPU frame (default = BASELINE config 2)")
    ap.add_argument("--spp", type=int, default=1000, help="samples per pixel (default = BASELINE config)")
    ap.add_argument("--scene", default="box")
    ap.add_argument("--weak", action="store_true", help="N>1: N x size^2 pixels instead of config 4's fixed (2 size)^2 frame")
    ap.add_argument("--devices", default="", help="N>1, single-process group path: comma list of HIP ordinals, one per replica (default 0..N-1; "
                                                  "ordinals may repeat - '0,0' rehearses the two-GPU path on a one-GPU box)")
    ap.add_argument("--launch", default="auto", choices=["auto", "ranks", "group"],
                    help="N>1: 'ranks' = one process per GPU under torch.distributed.run (what the driver starts; 'auto' takes it when WORLD_SIZE is set); "
                         "'group' = ONE process driving the N devices through mi_group_* - must be asked for by name")
    ap.add_argument("--rehearsal", action="store_true", help="allow --devices to repeat an ordinal (several replicas on one GPU); the line is marked as a rehearsal")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend of the ranks launch (nccl = RCCL; gloo with the ranks "
                                                                              "sharing the visible GPUs rehearses the ranks path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (parity spot check and CPU baseline)")
    ap.add_argument("--no-extras", action="store_true", help="skip the instrumented probe, the NIF kernel and the 16-spp frame (profiling runs)")
    args = ap.parse_args()

    import torch
    import ipu_ray_lib_amd as irl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Native libraries travel prebuilt; the build steps are no-ops when a binary's recorded source hash matches the
    # sources beside it. Only ONE process per node rebuilds what is missing or stale, and it does so before anything
    # touches the GPU or joins the process group (the rendezvous only happens afterwards, so no rank sits in a
    # collective while hipcc runs).
    import __graft_entry__ as ge
    if local_rank == 0:
        ge.build_cpu()
        ge.build_device()
    else:
        ge.wait_built()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if world == 1 and args.gpus > 1 and args.launch == "group":
        return bench_group(args, torch, irl)
    if args.launch == "group":
        raise SystemExit("bench.py: --launch group is the single-process form: start it as a plain `python bench.py --gpus N --launch group`, N > 1")
    if world != args.gpus:
        # (a plain `python bench.py --gpus N` does NOT fall into the single-process group launch: that path has never run on two
        # physical devices, so it has to be asked for by name)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                         f"--master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...` (one rank per GPU), or ask for the single-process C++ host path "
                         f"with `--launch group`")
    # one rank per GPU; with --backend gloo (a rehearsal) the ranks may outnumber the visible GPUs and share them
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
        dist.barrier()
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"       # where the collectives' tensors live

    width, height = image_shape(world, args.size, args.weak)
    scene = irl.HostScene.builtin(args.scene)
    d = scene.desc
    d.set_image(width, height)
    d.samples_per_pixel = args.spp
    d.device = dev_index
    dev = irl.IpuScene(d)

    from ipu_ray_lib_amd import sharding
    rows, cols = sharding.rank_pixels(width, height, rank, world)
    host_rays = make_stream(irl, rows, cols)
    n = host_rays.size
    d_rays = to_device(torch, irl, host_rays)
    stream = torch.cuda.current_stream()

    def gather():
        # the ONE collective of a frame: every rank's rgb tiles to rank 0 over RCCL/xGMI
        if dist is not None:
            rgb = d_rays.view(torch.float32).view(n, 21)[:, 0:3].contiguous()
            return sharding.gather_frame(dist, rgb if coll_dev == "cuda" else rgb.cpu(), width, height)
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
    if args.warmup == 0:
        gather()          # (a collective's first call builds its channels: never inside the timed region, whatever --warmup says)
    barrier()
    dev.reset_counters()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record(stream)
        dev.run_device(d_rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream)
        ev[s][1].record(stream)
        gather()
        ev[s][2].record(stream)       # (the collective's work.wait() made `stream` wait for it: the event is behind the gather)
    barrier()
    elapsed = time.perf_counter() - t0
    counters = dev.counters()
    kernel_ms = [a.elapsed_time(b) for a, b, _ in ev]
    gather_ms = [b.elapsed_time(c) for _, b, c in ev]

    tot = torch.tensor([elapsed, float(counters["casts"]), float(counters["paths"])], dtype=torch.float64, device=coll_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_casts, total_paths = float(tot[1]), float(tot[2])

    # N > 1: one more gather behind the timed region (a collective: every rank takes part), so that rank 0 can check the
    # GATHERED frame - dealing, every rank's render and the collective - against the oracle
    final_frame = None
    if dist is not None and not args.no_cpu_baseline:
        final_frame = gather()
        if final_frame is not None:
            final_frame = final_frame.cpu().numpy()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    frames_rendered = args.warmup + args.steps
    casts_per_launch = counters["casts"] / max(args.steps, 1)
    paths_per_launch = counters["paths"] / max(args.steps, 1)
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) * 1e-3 if kernel_ms else float("nan")
    props = torch.cuda.get_device_properties(dev_index)
    cus = int(props.multi_processor_count)
    clock_ghz = float(getattr(props, "clock_rate", 2400000)) / 1e6

    out = {
        "metric": "rays/sec (ray casts/s: CompactBvh intersect+occluded calls, whole node), built-in scene 1440x1440 path-trace",
        "value": total_casts / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if (world == 1 or args.weak) else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (built-in Cornell box + monkey bust scene, seeded per-pixel RNG streams)",
        "config": {"workload": f"built-in scene '{args.scene}', path-trace {width}x{height} x {args.spp} spp, max path length 10, "
                               f"roulette depth 3, AA 0.25, seed 1442 (BASELINE config {'2' if world == 1 else '4' if not args.weak else '2, weak-scaled frame'}), "
                               f"{n} pixels on rank 0",
                   "parallelism": f"ray tiles x{world}" + ((" + 1 RCCL gather/frame" if args.backend == "nccl" else " + 1 gloo gather/frame (rehearsal)") if world > 1 else "")},
        "paths_per_s": total_paths / elapsed,
        "ms_per_frame": elapsed / max(args.steps, 1) * 1e3,
        "casts_per_path": total_casts / max(total_paths, 1.0),
        "launch": "ranks" if world > 1 else "single",
    }
    if world > 1:
        out["one_gpu_same_frame_ms"] = one_gpu_same_frame_ms(torch, irl, d, width, height, dev_index)
        # what the record ran on, so that it cannot pass for something else: ranks of the process group, its backend, and how
        # many DISTINCT devices they used (gloo rehearsals share the visible GPUs)
        out["rccl_ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "distinct_devices": min(world, torch.cuda.device_count()) if args.backend == "gloo" else world,
                             "rehearsal": args.backend != "nccl"}
        out["gather_ms"] = gather_ms          # rank 0, per step: rgb tiles of every rank to rank 0 (pack + dist.gather), HIP events on the launch stream

    # ---------------- roofline of the dominant kernel (the path-trace launch) ----------------
    # Algorithmic bytes per cast: 24 B per node visited + the 42-B primitive record per leaf test + one
    # 36-B material; per pixel 84 B in + 84 B out once per frame. Nodes and primitive tests per cast are counted by the
    # instrumented kernel build on the same frame at 64 spp (untimed), which also reports the lanes active per phase.
    # The roof. The 0.5 MB scene is cache resident (HBM sees 0.1 % of its peak: `hbm` below), so the memory-side roof of
    # this kernel is the rate at which a CU can serve its walk's dependent 32-byte node gathers. MI355X_MICROARCH.md has
    # no figure for that (LDS, L2, HBM and MFMA rates only), so it is MEASURED: csrc/probe/gather_probe.hip performs only
    # the gathers - two global_load_dwordx4 per lane at an index that depends on the node fetched before, a tree-shaped
    # walk over THIS scene's node array, K1w's launch shape (256-thread workgroups, 6 per CU), as many of a wave's 64
    # lanes active as K1w's box-test turns have - and its rate (TA busy 0.99: profiles/r05_gather_probe_pmc.txt) is the
    # peak; `frac` = K1w's gathers per second / the probe's. In this run when the probe library is there, else from
    # profiles/r05_gather_probe.json.
    roof = {"kernel": "path_trace_wavefront_kernel", "avg_launch_ms": avg_kernel_s * 1e3}
    if not args.no_extras:
        probe_desc = irl.SceneDesc.from_buffer_copy(d)
        probe_desc.samples_per_pixel = min(args.spp, 64)
        probe = irl.IpuScene(probe_desc).set_option("full_stats", 1)
        probe_rays = to_device(torch, irl, host_rays)
        probe.run_device(probe_rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream)
        torch.cuda.synchronize()
        pc, ph = probe.counters(), probe.phase_stats()
        probe.close(); del probe_rays
        nodes_per_cast = pc["nodes_visited"] / max(pc["casts"], 1)
        leaf_per_cast = pc["leaf_tests"] / max(pc["casts"], 1)
        bytes_per_cast = 24.0 * nodes_per_cast + 42.0 * leaf_per_cast + 36.0
        alg_bytes_launch = casts_per_launch * bytes_per_cast + paths_per_launch / args.spp * 168.0
        achieved_gbs = alg_bytes_launch / avg_kernel_s / 1e9
        cyc = ph["cycles"]
        lanes = {k: (ph[k]["lanes"] / (64.0 * ph[k]["iters"]) if ph[k]["iters"] else 0.0) for k in ("node", "leaf", "shade", "gen")}
        # K1w's gathers: one 32-byte device node per node visited, 48 bytes (1.5 x 32) of primitive record per leaf test
        gathers_per_cast = nodes_per_cast + 1.5 * leaf_per_cast
        gathers_per_s = casts_per_launch * gathers_per_cast / avg_kernel_s
        node_lanes = max(1, min(64, int(round(lanes["node"] * 64))))
        gather = {"gathers_per_cast": gathers_per_cast, "gathers_per_s": gathers_per_s, "active_lanes": node_lanes,
                  "unit": "32-byte lane-gathers per second, whole chip"}
        try:
            sys.path.insert(0, str(ROOT / "tools"))
            import gather_probe as gp
            lib = gp.build()
            nodes32 = gp.device_nodes(scene)
            gather["attainable_gathers_per_s"] = gp.measure(lib, nodes32, 0, 1, node_lanes)[1]
            gather["attainable_full_wave"] = gp.measure(lib, nodes32, 0, 1, 64)[1]             # the same walk with all 64 lanes of every wave gathering
            gather["attainable_uniform_random"] = gp.measure(lib, nodes32, 0, 0, node_lanes)[1]
            gather["attainable_from_lds"] = gp.measure(lib, nodes32, 1, 1, node_lanes)[1]
            gather["attainable_from_lds_full_wave"] = gp.measure(lib, nodes32, 1, 1, 64)[1]
            gather["measured"] = "in this run (ipu_ray_lib_amd/libmi_gather_probe.so)"
        except (OSError, ge.StaleBinary) as e:      # no probe library here and nothing to build it with: the committed run of the same probe.
            # (Anything else - a HIP error inside the probe, say - is a failure of this run and propagates.)
            gather["probe_error"] = f"{type(e).__name__}: {e}"
            pj = json.loads(GATHER_PROBE.read_text()) if GATHER_PROBE.exists() else {"rows": []}
            row = min((r for r in pj["rows"] if r["path"].startswith("L1") and r["walk"] == "tree-shaped" and r["wg_per_cu"] == 6),
                      key=lambda r: abs(r["active_lanes"] - node_lanes), default=None)
            if row:
                gather["attainable_gathers_per_s"] = row["lane_gathers_per_s"]
                gather["measured"] = f"offline: {GATHER_PROBE.name} @ {pj.get('source_commit')}, {row['active_lanes']} lanes ({type(e).__name__}: probe not runnable here)"
        att = gather.get("attainable_gathers_per_s")
        att64 = gather.get("attainable_full_wave")
        roof.update({
            "bound": "l1", "achieved": achieved_gbs, "unit": "GB/s",
            "peak": (att * bytes_per_cast / gathers_per_cast / 1e9) if att else None,
            # three readings of the same launch, from the most forgiving roof to the strictest:
            #   frac            against the gather rate of the probe run at K1w's OWN lane count (~33 of 64: the roof embeds the kernel's half-empty waves)
            #   frac_full_wave  against the same probe with all 64 lanes of every wave gathering (what the L1 path serves a kernel without divergence)
            #   frac_l2         algorithmic bytes against the guide's 34.5 TB/s aggregate L2 figure (the only on-chip global-load rate in the guide)
            "frac": (gathers_per_s / att) if att else None,
            "frac_full_wave": (gathers_per_s / att64) if att64 else None,
            "frac_l2": achieved_gbs / L2_PEAK_GBS,
            "peak_source": "measured: the node-gather microbenchmark csrc/probe/gather_probe.hip (tools/gather_probe.py) - tree-shaped dependent walk over this "
                           "scene's node array, two 16-byte loads per lane, K1w's occupancy and lane count - converted to algorithmic bytes with this "
                           "frame's bytes per gather; MI355X_MICROARCH.md gives no vector-L1 gather rate",
            "gather_roof": gather,
            "l2": {"peak": L2_PEAK_GBS, "frac": achieved_gbs / L2_PEAK_GBS, "source": "MI355X_MICROARCH.md (34.5 TB/s aggregate)"},
            "bytes_per_cast": bytes_per_cast, "nodes_per_cast": nodes_per_cast, "leaf_tests_per_cast": leaf_per_cast,
            "lanes_active": dict(lanes, note="instrumented build, same frame at 64 spp: lanes in the phase / 64 per wave turn"),
            "cycle_share": {k: cyc[k] / max(cyc["total"], 1) for k in ("traverse", "shade", "gen")},
        })
        traffic = None
        if PMC_SUMMARY.exists():
            # measured offline by rocprofv3 --pmc passes of this same command (separate passes per counter group,
            # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); only quoted for the profiled workload,
            # every such field marked with the file and the commit it was collected at
            pm = json.loads(PMC_SUMMARY.read_text())
            if pm.get("workload") == [args.scene, width, height, args.spp, world]:
                mark = f"offline, {PMC_SUMMARY.name} @ {pm.get('source_commit')}"
                traffic = pm.get("hbm_bytes_per_launch")
                roof["hbm"] = {"traffic_bytes_per_launch": traffic, "achieved": traffic / avg_kernel_s / 1e9, "peak": HBM_PEAK_GBS,
                               "frac": traffic / avg_kernel_s / 1e9 / HBM_PEAK_GBS, "unit": "GB/s", "measured": mark}
                roof["valu"] = dict(pm.get("valu", {}), measured=mark)
                if pm.get("l1_accesses_per_clk_per_cu") is not None:
                    roof["l1_accesses_per_clk_per_cu"] = {"k1w": pm["l1_accesses_per_clk_per_cu"], "ta_busy": pm.get("ta_busy"),
                                                          "probe_at_saturation": pm.get("probe_l1_accesses_per_clk_per_cu"), "measured": mark}
        roof["traffic"] = traffic
    out["roofline"] = roof

    if not args.no_extras and world == 1:
        # ---------------- K3, the NIF MLP on MFMA: 1440^2 rays, synthetic weights of the reference's shapes ----------------
        ks, bs, relu, dims = nif_weights(np.random.default_rng(0))
        nif_scene = irl.HostScene.builtin("spheres")          # (kept alive: the desc points into it)
        ns = irl.IpuScene(nif_scene.desc)
        ns.setNif(ks, bs, relu, 12, 3.43, np.array([-2.35, -2.27, -1.96], np.float32), True)
        nr = 1440 * 1440
        u = torch.rand(nr, device="cuda"); v = torch.rand(nr, device="cuda"); bgr = torch.empty(nr, 3, device="cuda")
        # the default kernel for this network (nif_shape auto: K3a, the hand-scheduled register-resident kernel) - 20 launches, the
        # clock of the last one from the kernel's own counter reads (mi_get_nif_clock) - and nif_mlp_kernel (w6) beside it
        ms = time_launches(torch, lambda: ns.nif_infer_device(u.data_ptr(), v.data_ptr(), bgr.data_ptr(), nr, stream.cuda_stream), 20, stream, ahead=3)
        ghz = ns.nif_clock_ghz()
        ns.set_option("nif_shape", "w6")
        ms_w6 = time_launches(torch, lambda: ns.nif_infer_device(u.data_ptr(), v.data_ptr(), bgr.data_ptr(), nr, stream.cuda_stream), 20, stream, ahead=3)
        ns.set_option("nif_shape", "auto")
        flops_per_ray = 2 * sum(k * c for k, c in dims)
        tf = nr * flops_per_ray / (ms * 1e-3) / 1e12
        mfma_cycles = nr / 16.0 * (flops_per_ray / 2.0 / (16 * 32)) * 16.0 / (4 * cus)         # 16 cycles per v_mfma_f32_16x16x32_f16 per SIMD (real MACs only)
        out["nif"] = {"kernel": "nif_asm_kernel (K3a)" if ghz else "nif_mlp_kernel (w6)", "rays": nr, "avg_launch_ms": ms, "rays_per_s": nr / (ms * 1e-3), "flops_per_ray": flops_per_ray,
                      "shader_clock_ghz": ghz, "shader_clock_note": "delta s_memtime / delta s_memrealtime x 100 MHz of the kernel's first wave, which lives as long as the launch (mi_get_nif_clock): "
                                                                      "the clock the chip's power management left this launch; boxes of the pool differ",
                      "matrix_pipe_busy_at_that_clock": (mfma_cycles / (ms * 1e-3 * ghz * 1e9)) if ghz else None,
                      "nif_mlp_kernel_w6_ms": ms_w6,
                      "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS,
                                   "dtype": "f16 in / f32 accumulate"}}
        ns.close(); del u, v, bgr
        # ---------------- the 16-spp preview frame (launch ramp-up and drain dominate) ----------------
        pd = irl.SceneDesc.from_buffer_copy(d)
        pd.samples_per_pixel = 16
        pv = irl.IpuScene(pd)
        pv_rays = to_device(torch, irl, host_rays)
        ms = time_launches(torch, lambda: pv.run_device(pv_rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream), 10, stream)
        pcnt = pv.counters()
        out["preview_16spp"] = {"ms_per_frame": ms, "rays_per_s": pcnt["casts"] / 11.0 / (ms * 1e-3), "workload": f"{width}x{height} x 16 spp"}
        pv.close(); del pv_rays
        # ---------------- the tolerance tier (scene option "fast": FMA box / triangle tests), same frame; never the headline ----------------
        fd = irl.IpuScene(irl.SceneDesc.from_buffer_copy(d)).set_option("fast", 1)
        f_rays = to_device(torch, irl, host_rays)
        ms = time_launches(torch, lambda: fd.run_device(f_rays.data_ptr(), n, irl.MODE_PATH_TRACE, stream.cuda_stream), 2, stream)
        fcnt = fd.counters()
        out["fast_tier"] = {"ms_per_frame": ms, "rays_per_s": fcnt["casts"] / 3.0 / (ms * 1e-3), "workload": f"{width}x{height} x {args.spp} spp",
                            "note": "not bit-exact: results within the tolerance stated in tests/test_gpu_parity.py::test_fast_tier_within_its_stated_tolerance"}
        fd.close(); del f_rays
        # ---------------- BASELINE configs 3 and 5 at a fraction of their samples (rates as trace.cpp:328-333 defines them) ----------------
        try:
            ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
        except AttributeError:
            ncores = max(1, min(os.cpu_count() or 1, 16))
        out["config3"] = config3_record(torch, irl, stream)
        out["config5"] = config5_record(torch, irl, stream, ncores, check=not args.no_cpu_baseline)

    rc = 0
    if final_frame is not None:
        # ---------------- N > 1: parity of the gathered frame (rgb is what the gather moves) ----------------
        import oracle_lib
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, 16))
        stride = max(1, (width * height) // 1009)
        flat = np.arange(0, width * height, stride)
        rr, cc = flat // width, flat % width
        want = make_stream(irl, rr, cc)
        tp = time.perf_counter()
        for _ in range(frames_rendered):
            oracle_lib.path_trace_pixel_rng(d, want, cores)
        got_rgb = np.ascontiguousarray(final_frame[rr, cc]).astype(np.float32)
        want_rgb = np.stack([want["rgb"][k] for k in "xyz"], 1).astype(np.float32)
        bad = int((got_rgb.view(np.uint32) != want_rgb.view(np.uint32)).any(axis=1).sum())
        out["parity_checked_pixels"] = int(flat.size)
        out["parity_mismatches"] = bad
        out["parity_note"] = (f"every {stride}th pixel of the frame gathered on rank 0 after {frames_rendered} accumulated frames x {args.spp} spp: rgb, "
                              f"bit for bit, vs oracle/ray_oracle.c ({time.perf_counter() - tp:.1f} s); the other 72 bytes of a TraceResult stay on their rank")
        if bad:
            rc = 1
    if not args.no_cpu_baseline and world == 1:
        import oracle_lib
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, 16))            # the GPU box grants 16 host cores per GPU
        # ---------------- parity of the frames that were timed ----------------
        # Every pixel owns its RNG streams, so any subset of pixels is exact on its own: ~2 000 pixels of the LAST frame,
        # all 84 bytes, against the oracle run `frames_rendered` times over them (rgb accumulates from frame to frame,
        # exactly as it did on the device) at the full sample count.
        stride = max(1, n // 2011)
        idx = np.arange(0, n, stride)
        got = np.frombuffer(d_rays[torch.from_numpy(idx).cuda()].cpu().numpy().tobytes(), dtype=irl.TRACE_RESULT)
        want = host_rays[idx].copy()
        tp = time.perf_counter()
        for _ in range(frames_rendered):
            oracle_lib.path_trace_pixel_rng(d, want, cores)
        gb = got.view(np.uint8).reshape(got.size, -1); wb = want.view(np.uint8).reshape(want.size, -1)
        bad = int((gb != wb).any(axis=1).sum())
        out["parity_checked_pixels"] = int(idx.size)
        out["parity_mismatches"] = bad
        out["parity_note"] = (f"every {stride}th pixel of the last timed frame, all 84 bytes of the TraceResult, vs oracle/ray_oracle.c over "
                              f"{frames_rendered} accumulated frames x {args.spp} spp ({time.perf_counter() - tp:.1f} s)")
        if bad:
            rc = 1
        # ---------------- CPU baseline: the oracle on a bounded sample of the same frame ----------------
        cpu_desc = irl.SceneDesc.from_buffer_copy(d)
        cpu_desc.samples_per_pixel = min(args.spp, 250)
        # calibrate on a coarse pixel grid, then size the sample for about 15 s of CPU work
        rr, cc = np.meshgrid(np.arange(0, height, 48), np.arange(0, width, 48), indexing="ij")
        cal = make_stream(irl, rr.reshape(-1), cc.reshape(-1))
        tc = time.perf_counter()
        oracle_lib.path_trace_pixel_rng(cpu_desc, cal, cores)
        cal_s = max(time.perf_counter() - tc, 1e-3)
        want_px = cal.size * 15.0 / cal_s
        step_px = int(min(48, max(2, round((width * height / want_px) ** 0.5))))
        rr, cc = np.meshgrid(np.arange(0, height, step_px), np.arange(0, width, step_px), indexing="ij")
        cpu_rays = make_stream(irl, rr.reshape(-1), cc.reshape(-1))
        tc = time.perf_counter()
        st = oracle_lib.path_trace_pixel_rng(cpu_desc, cpu_rays, cores)
        cpu_s = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": st.casts / cpu_s, "unit": "rays/s", "cores": cores, "kind": "port",
                               "sample": f"every {step_px}th pixel of the {width}x{height} frame ({cpu_rays.size} pixels) x "
                                         f"{cpu_desc.samples_per_pixel} spp, {st.casts} casts in {cpu_s:.1f} s, oracle/ray_oracle.c (-O3, OpenMP)"}
    if "config5" in out and "parity" in out["config5"] and not out["config5"]["parity"]["ok"]:
        rc = 1
    emit(out)
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
