#!/usr/bin/env python3
"""K3's k-loop skeleton in the two f16 MFMA shapes (csrc/probe/mfma_shape_probe.hip): TFLOP/s of
  0: v_mfma_f32_16x16x32_f16, 4 waves x (5 x 6 tiles)   - the shipped formulation's operand traffic
  1: v_mfma_f32_32x32x16_f16, 5 waves x (2 x 3 tiles)
  2: v_mfma_f32_32x32x16_f16, 4 waves x ({3,3,2,2} x 3 tiles)
on 6 layers of 320 x 320, 96 rays per workgroup, two workgroups per CU, random operands.
    python tools/mfma_shape_probe.py [--out profiles/r03_mfma_shape_probe.txt] [--only SHAPE]"""
import argparse, ctypes as C, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "ipu_ray_lib_amd" / "csrc" / "probe" / "mfma_shape_probe.hip"
LIB = ROOT / "build" / "probe" / "libmfma_shape_probe.so"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=""); ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--passes", type=int, default=200); ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    LIB.parent.mkdir(parents=True, exist_ok=True)
    if not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-shared", "-o", str(LIB), str(SRC)], check=True)
    lib = C.CDLL(str(LIB))
    lib.msp_run.restype = C.c_double
    lib.msp_run.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    names = {0: "16x16x32, 4 waves x (5 x 6 tiles)", 1: "32x32x16, 5 waves x (2 x 3 tiles)", 2: "32x32x16, 4 waves x ({3,3,2,2} x 3 tiles)"}
    lines = []
    for rnd in (1, 2):
        for shape in ([a.only] if a.only >= 0 else [0, 1, 2]):
            fl = C.c_double(); bl = C.c_uint32()
            ms = lib.msp_run(shape, 6, a.passes, a.reps, C.byref(fl), C.byref(bl))
            if ms <= 0:
                raise SystemExit("probe failed")
            lines.append(f"round {rnd}  shape {shape} ({names[shape]}): {ms:8.3f} ms  {fl.value / (ms * 1e-3) / 1e12:8.1f} TFLOP/s  ({fl.value / (ms * 1e-3) / 2.5e15:.3f} of 2.5 PF)")
            print(lines[-1], flush=True)
    if a.out:
        Path(a.out).write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
