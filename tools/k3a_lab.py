#!/usr/bin/env python3
"""Placements and timing-only knock-outs of K3a's generated body (csrc/nif_asm_gen.py), timed in interleaved rounds in ONE process.

    python tools/k3a_lab.py build NAME=GENARGS ...     (here: cross-compiles build/k3a/lib_NAME.so, a few at a time)
    python tools/k3a_lab.py run [--rounds 5] [--reps 20] [--out FILE]     (on the GPU box: every library under build/k3a)

GENARGS are nif_asm_gen.py options (and -D definitions for the C++ round the body) with ',' for ' ', e.g. base= ring6=--ring,6 nodma=--ko,1. `run` times K3 (w6) from the first
library as the yardstick, then every K3a placement, round after round; it prints median and minimum per build and - for builds
without a knock-out - compares the results with the first build's (bit for bit: placements must not change a result)."""
import ctypes as C, json, os, subprocess, sys, statistics
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "build" / "k3a"
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-shared", "-Wno-unused-function"]


def weights(rng, hidden=320, embed=12, layers=6):
    F = 4 * embed
    dims = [(F, hidden)] + [((hidden + F) if l == layers // 2 else hidden, hidden) for l in range(1, layers)] + [(hidden, 3)]
    ks = [(rng.normal(size=d) * np.sqrt(2.0 / d[0])).astype(np.float32) for d in dims]
    bs = [(rng.normal(size=d[1]) * 0.05).astype(np.float32) for d in dims]
    return ks, bs, [1] * (len(dims) - 1) + [0], dims


def build(specs):
    OUT.mkdir(parents=True, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, args = spec.partition("=")
        inc = OUT / f"body_{name}.inc"
        words = [a for a in args.replace(",", " ").split() if a]
        defs = [w for w in words if w.startswith("-D")]      # (-DMI_K3A_KO=n: knock-outs in the C++ around the body)
        subprocess.run([sys.executable, str(ROOT / "ipu_ray_lib_amd" / "csrc" / "nif_asm_gen.py"), str(inc)] + [w for w in words if not w.startswith("-D")], check=True)
        (OUT / f"lib_{name}.args").write_text(args)
        which_b = "--mt" in words and words[words.index("--mt") + 1] == "4"      # K3b: four waves x four ray tiles (use with --waves,4,--tag,_B)
        other = ROOT / "ipu_ray_lib_amd" / "csrc" / ("nif_asm_body.inc" if which_b else "nif_asm_body_b.inc")
        body_defs = [f'-DMI_NIF_ASM_BODY_B_INC="{inc}"', f'-DMI_NIF_ASM_BODY_INC="{other}"', "-DMI_LAB_WHICH=4u"] if which_b else [f'-DMI_NIF_ASM_BODY_INC="{inc}"', f'-DMI_NIF_ASM_BODY_B_INC="{other}"']
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *defs, *body_defs, "-I", str(ROOT / "include"), "-o", str(OUT / f"lib_{name}.so"),
               str(ROOT / "ipu_ray_lib_amd" / "csrc" / "probe" / "k3a_lab.hip")]
        procs.append(subprocess.Popen(cmd))
        if len(procs) >= 4:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0


def run(rounds, reps, out, cus=()):
    import torch  # noqa: F401  (one HIP runtime per process, loaded first)
    libs = sorted(OUT.glob("lib_*.so"), key=lambda p: (p.stem != "lib_base", p.stem))
    rng = np.random.default_rng(0)
    ks, bs, relu, dims = weights(rng)
    n = 1440 * 1440
    u = rng.random(n).astype(np.float32); v = rng.random(n).astype(np.float32)
    L = len(ks)
    kp = (C.c_void_p * L)(*[k.ctypes.data for k in ks]); bp = (C.c_void_p * L)(*[b.ctypes.data for b in bs])
    rows = np.array([k.shape[0] for k in ks], np.uint32); cols = np.array([k.shape[1] for k in ks], np.uint32); rl = np.array(relu, np.uint8)
    mean = np.array([-2.35, -2.27, -1.96], np.float32)
    hs = {}
    for p in libs:
        lib = C.CDLL(str(p))
        lib.lab_create.restype = C.c_void_p
        lib.lab_create.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32]
        lib.lab_time.restype = C.c_double; lib.lab_time.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.lab_result.argtypes = [C.c_void_p, C.c_void_p]; lib.lab_destroy.argtypes = [C.c_void_p]
        h = lib.lab_create(L, kp, bp, rows.ctypes.data, cols.ctypes.data, rl.ctypes.data, 12, 3.43, mean.ctypes.data, 1, u.ctypes.data, v.ctypes.data, n)
        assert h, p
        hs[p.stem[4:]] = (lib, C.c_void_p(h))
    names = list(hs)
    times = {"K3(w6)": []}; times.update({k: [] for k in names})
    for r in range(rounds):
        lib, h = hs[names[0]]
        times["K3(w6)"].append(lib.lab_time(h, 0, reps, 256))
        for k in names:
            lib, h = hs[k]
            times[k].append(lib.lab_time(h, 1, reps, 256))
    # fewer workgroups than compute units (the rest of the chip idle): does the clock of a power-limited launch make up for the units left out?
    if cus:
        lib, h = hs[names[0]]
        for c in cus:
            ts = [lib.lab_time(h, 1, reps, c) for _ in range(3)]
            buf = (C.c_ulonglong * 4)()
            ghz = buf[0] / buf[1] * 0.1 if lib.lab_stamps(buf) == 0 and buf[1] else float("nan")
            print(f"{names[0]} on {c:3d} workgroups: median {statistics.median(ts):.4f} ms  ({statistics.median(ts) * c / 256:.4f} ms x units / 256)  clock {ghz:.3f} GHz", flush=True)
    ref = None
    rec = {"rays": n, "rounds": rounds, "reps": reps, "builds": {}}
    for k, ts in times.items():
        args = (OUT / f"lib_{k}.args").read_text() if (OUT / f"lib_{k}.args").exists() else ""
        same = None
        if k in hs and "--ko" not in args and "MI_K3A_KO" not in args:
            lib, h = hs[k]
            got = np.zeros((n, 3), np.float32); lib.lab_result(h, got.ctypes.data)
            if ref is None: ref = got
            same = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
        st = None
        if k in hs:
            lib, h = hs[k]
            buf = (C.c_ulonglong * 4)()
            if lib.lab_stamps(buf) == 0 and buf[1]:
                st = {"clock_ghz": buf[0] / buf[1] * 0.1, "kernel_cycles": buf[0], "body_cycles": buf[2], "passes": buf[3],
                      "body_cycles_per_pass": buf[2] / max(buf[3], 1), "outside_body_cycles_per_pass": (buf[0] - buf[2]) / max(buf[3], 1)}
                print(f"    stamps (workgroup 0, wave 0): {st}")
        rec["builds"][k] = {"args": args, "stamps": st, "median_ms": statistics.median(ts), "min_ms": min(ts), "all_ms": ts, "same_bits_as_first": same}
        print(f"{k:<14} median {statistics.median(ts):.4f} ms   min {min(ts):.4f} ms   {args}   {'' if same is None else ('bits == first' if same else 'BITS DIFFER')}", flush=True)
    if out:
        Path(out).write_text(json.dumps(rec, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        import argparse
        ap = argparse.ArgumentParser(); ap.add_argument("cmd"); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--out", default="")
        ap.add_argument("--cus", default="", help="also time the first build on these workgroup counts, e.g. 256,240,224")
        a = ap.parse_args()
        run(a.rounds, a.reps, a.out, [int(c) for c in a.cus.split(",") if c])
