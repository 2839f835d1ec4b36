#!/usr/bin/env python3
"""Where one wave of K3 spends its cycles (diagnostic build: hipcc ... -DMI_NIF_STAMPS=1 -o build/ab/lib_stamps.so,
run with MI_RAYLIB_LIB=build/ab/lib_stamps.so). Prints the shares of the s_memtime brackets in nif_kernels.hpp summed
over the second wave of every workgroup's row groups; the build's run time itself is not a measurement.
MI_NIF_DIAG_ONE_WG=1 (stamp builds only) asks for the whole LDS, i.e. one workgroup per CU = one wave per SIMD for the
4-wave shape. -DMI_NIF_KO=1|2|3 builds knock the weight loads (1) / the LDS reads (2) out of the k-loop (wrong results:
timing and stamps only) to price them."""
import ctypes as C, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl
from tools.bench_nif import weights


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1440 * 1440
    rng = np.random.default_rng(0)
    ks, bs, relu, dims = weights(rng)
    s = irl.HostScene.builtin("spheres")
    dev = irl.IpuScene(s.desc)
    dev.setNif(ks, bs, relu, 12, 3.43, np.array([-2.35, -2.27, -1.96], np.float32), True)
    u = torch.rand(n, device="cuda"); v = torch.rand(n, device="cuda"); out = torch.empty(n, 3, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib = irl.device_lib()
    buf = (C.c_ulonglong * 8)()
    for _ in range(2):
        dev.nif_infer_device(u.data_ptr(), v.data_ptr(), out.data_ptr(), n, st)
    torch.cuda.synchronize()
    assert lib.mi_debug_nif_stamps(buf) == 0
    dev.nif_infer_device(u.data_ptr(), v.data_ptr(), out.data_ptr(), n, st)
    torch.cuda.synchronize()
    assert lib.mi_debug_nif_stamps(buf) == 0
    names = ["k_loops", "barrier_after_k_loop", "epilogue_body", "barrier_after_epilogue", "staging", "total", "waves"]
    vals = dict(zip(names, [int(x) for x in buf[:7]]))
    tot = vals["total"]
    print(json.dumps({k: (v if k == "waves" else round(v / tot, 4)) for k, v in vals.items()} | {"ticks_per_wave": tot / max(1, vals["waves"])}))


if __name__ == "__main__":
    main()
