#!/usr/bin/env python3
"""Node-gather roof of K1w's access shape: builds csrc/probe/gather_probe.hip and runs it on the
built-in scene's device node array for N = 16 / 35 / 64 active lanes, the L1, all-LDS and mixed data paths, uniform and
walk-shaped index sequences, at K1w's occupancy (6 workgroups of 256 threads per CU; 5 up to round 3) and at 5, 4 and 8.

    python tools/gather_probe.py [--scene box] [--out profiles/r03_gather_probe.json] [--quick]

Prints a table and writes the JSON. Units: lane-gathers (one lane fetching one 32-byte node = two 16-byte loads) per
shader clock per CU, with the clock the run measured (GRBM-free: kernel time x the device's nominal clock is NOT used;
the probe reports rates per second and, from --clock-ghz (default: the nominal 2.4), per clock)."""
import argparse, ctypes as C, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ipu_ray_lib_amd as irl

LIB = ROOT / "ipu_ray_lib_amd" / "libmi_gather_probe.so"       # built by __graft_entry__.build_probe() from csrc/probe/gather_probe.hip


def build():
    """The probe library (built in-tree next to the product library so that it travels to the GPU box; it is NOT part of
    the product: nothing under ipu_ray_lib_amd/*.py loads it)."""
    import __graft_entry__ as ge
    ge.build_probe()
    lib = C.CDLL(str(LIB))
    lib.gp_run.restype = C.c_double
    lib.gp_run.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
    return lib


def measure(lib, nodes, path, walk, lanes, wg=6, steps=20000, reps=3, lds_nodes=512):
    """One configuration: (average launch ms, lane-gathers per second chip-wide)."""
    blocks = C.c_uint32()
    ms = lib.gp_run(nodes.ctypes.data, nodes.size, path, walk, steps, lanes, lds_nodes if path else 1, wg, reps, C.byref(blocks))
    if ms <= 0:
        raise RuntimeError("gather probe failed")
    return ms, blocks.value * 4 * lanes * steps / (ms * 1e-3)


def device_nodes(scene):
    """The 32-byte device node array exactly as raylib.hip builds it (min, max = min + (float)half extent interleaved,
    link = next(i) in preorder, leaf index or 0xFFFFFFFF)."""
    n = scene.nodes
    N = n.size
    out = np.zeros(N, np.dtype([("minx", "<f4"), ("maxx", "<f4"), ("miny", "<f4"), ("maxy", "<f4"), ("minz", "<f4"), ("maxz", "<f4"), ("link", "<u4"), ("leaf", "<u4")]))
    ext = [n[k].view(np.float16).astype(np.float32) for k in ("dx", "dy", "dz")]
    for a, lo, e in (("x", n["min_x"], ext[0]), ("y", n["min_y"], ext[1]), ("z", n["min_z"], ext[2])):
        out["min" + a] = lo; out["max" + a] = lo + e
    leaf = n["geomID"] != 0xFFFF
    skip = np.zeros(N + 1, np.uint32)
    second = n["link"]
    for i in range(N - 1, -1, -1):
        skip[i] = i + 1 if leaf[i] else skip[second[i]]
    out["link"] = skip[:N]
    out["leaf"] = np.where(leaf, np.cumsum(leaf) - 1, 0xFFFFFFFF).astype(np.uint32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="box"); ap.add_argument("--out", default="")
    ap.add_argument("--steps", type=int, default=20000); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="path,walk,lanes,wg (one configuration, e.g. for a rocprofv3 --pmc pass)")
    args = ap.parse_args()
    lib = build()
    s = irl.HostScene.builtin(args.scene)
    nodes = device_nodes(s)
    rows = []
    paths = {0: "L1 (global)", 1: "LDS", 2: "mixed"}
    cfgs = []
    if args.only:
        p, w, l, g = [int(x) for x in args.only.split(",")]
        cfgs = [(p, w, l, g, 512)]
    else:
        for wg in ((6,) if args.quick else (6, 5, 4, 8)):
            for walk in (1, 0):
                for path in (0, 1, 2):
                    for lanes in (16, 35, 64):
                        cfgs.append((path, walk, lanes, wg, 512))
    for path, walk, lanes, wg, ldsn in cfgs:
        ldsn_eff = ldsn if path else 0
        ms, per_s = measure(lib, nodes, path, walk, lanes, wg, args.steps, args.reps, ldsn)
        per_clk_cu = per_s / 256 / (args.clock_ghz * 1e9)
        rows.append({"path": paths[path], "walk": "tree-shaped" if walk else "uniform", "active_lanes": lanes, "wg_per_cu": wg, "lds_nodes": ldsn_eff,
                     "ms": ms, "lane_gathers_per_s": per_s, "lane_gathers_per_clk_per_cu": per_clk_cu,
                     "wave_steps_per_clk_per_cu": per_clk_cu / lanes})
        print(f"wg/CU {wg}  {rows[-1]['walk']:<11} {paths[path]:<12} lanes {lanes:>2}: {ms:8.3f} ms  {per_s:.3e} lane-gathers/s  "
              f"{per_clk_cu:.3f} /clk/CU  (wave steps {per_clk_cu / lanes * 1e3:.2f} per 1000 clk per CU)", flush=True)
    if args.out:
        import subprocess
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or None
        except OSError:
            commit = None
        Path(args.out).write_text(json.dumps({"scene": args.scene, "nodes": int(nodes.size), "steps": args.steps, "clock_ghz_assumed": args.clock_ghz,
                                              "source_commit": commit, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
