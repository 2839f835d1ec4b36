#!/usr/bin/env python3
"""Throughput of K3 (the NIF MLP on MFMA): rays/s and TFLOP/s against the dense f16 MFMA peak.
Weights: synthetic, the reference's shapes (48->320->320->320->(320+48)->320->320->3, nif_metadata.txt)."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl

MFMA_F16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense


def weights(rng, hidden=320, embed=12, layers=6):
    F = 4 * embed
    dims = [(F, hidden)] + [((hidden + F) if l == layers // 2 else hidden, hidden) for l in range(1, layers)] + [(hidden, 3)]
    ks = [(rng.normal(size=d) * np.sqrt(2.0 / d[0])).astype(np.float32) for d in dims]
    bs = [(rng.normal(size=d[1]) * 0.05).astype(np.float32) for d in dims]
    return ks, bs, [1] * (len(dims) - 1) + [0], dims


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=1440 * 1440)
    ap.add_argument("--shape", default="auto", help="scene option nif_shape: auto (default: K3a for this network) | w6 | t6 | t4 (nif_mlp_kernel) | r8 | r8s (K3r, nif_regs_kernel.hpp) | a8 (K3a, nif_asm_kernel.hpp)")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    n = a.n
    rng = np.random.default_rng(0)
    ks, bs, relu, dims = weights(rng)
    s = irl.HostScene.builtin("spheres")
    dev = irl.IpuScene(s.desc, variants=a.shape.startswith("r")).set_option("nif_shape", a.shape)      # (K3r: the variants build)
    dev.setNif(ks, bs, relu, 12, 3.43, np.array([-2.35, -2.27, -1.96], np.float32), True)
    u = torch.rand(n, device="cuda"); v = torch.rand(n, device="cuda"); out = torch.empty(n, 3, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    dev.nif_infer_device(u.data_ptr(), v.data_ptr(), out.data_ptr(), n, st)
    torch.cuda.synchronize()
    for _ in range(3):      # (no wait between these and the timed launches: an idle device needs ~1 ms to get its clocks back)
        dev.nif_infer_device(u.data_ptr(), v.data_ptr(), out.data_ptr(), n, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = a.reps
    e0.record()
    for _ in range(reps):
        dev.nif_infer_device(u.data_ptr(), v.data_ptr(), out.data_ptr(), n, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops_per_ray = 2 * sum(k * c for k, c in dims)
    tf = n * flops_per_ray / (ms * 1e-3) / 1e12
    print(json.dumps({"kernel": "nif_regs_kernel" if a.shape in ("r8", "r8s") else "nif_asm_kernel" if a.shape in ("a8", "b4", "auto") else "nif_mlp_kernel", "shape": a.shape, "rays": n, "ms": ms, "rays_per_s": n / (ms * 1e-3), "flops_per_ray": flops_per_ray,
                      "tflops": tf, "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS}}))


if __name__ == "__main__":
    main()
