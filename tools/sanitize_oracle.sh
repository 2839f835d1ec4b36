#!/bin/bash
# The CPU oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer, on the oracle calls that ran right before
# and inside the test in which round 4's one unexplained abort happened (gpurun_out/r5z: the plain oracle render of
# test_launch_grids_follow_the_compute_unit_count - box scene, 200 x 120 x 70 spp, 16 threads - and the NIF path trace the two
# config-5 tests before it make): a heap overrun there would have been found by glibc inside the next native call, mi_render.
#   tools/sanitize_oracle.sh        (CPU only; prints two "ok" lines when clean)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/san
gcc -std=c11 -O1 -g -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -o $R/build/san/libray_oracle.so $R/oracle/ray_oracle.c -lm
cat > $R/build/san/oracle_asan.py <<'PY'
import sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as ol
ol.ORACLE_SO = ROOT / "build" / "san" / "libray_oracle.so"
import ipu_ray_lib_amd as irl
s = irl.HostScene.builtin("box"); d = s.desc
d.set_image(200, 120); d.samples_per_pixel = 70; d.path_trace = 1
want = s.init_ray_stream(); st = ol.path_trace_pixel_rng(d, want, 16)
print("plain oracle render ok:", st.casts, "casts")
rng = np.random.default_rng(8)
F = 48; dims = [(F, 320)] + [((320 + F) if l == 3 else 320, 320) for l in range(1, 6)] + [(320, 3)]
ks = [(rng.normal(size=dd) * np.sqrt(2.0 / dd[0])).astype(np.float16).astype(np.float32) for dd in dims]
bs = [(rng.normal(size=dd[1]) * 0.05).astype(np.float32) for dd in dims]
m = irl.HostScene.builtin("monkey"); dm = m.desc
dm.set_image(1440, 1440); dm.samples_per_pixel = 64; dm.path_trace = 1
sub = m.init_ray_stream()[::30011].copy()
nif, keep = ol.make_nif(ks, bs, [1] * 6 + [0], 12, 3.43, np.array([-2.35, -2.27, -1.96], np.float32), True, half_features=True, half_weights_acts=True)
st2 = ol.Stats()
ol.lib().o_path_trace_nif_pixel_rng(C.byref(dm), C.byref(nif), 0.0, sub.ctypes.data, sub.size, 16, C.byref(st2))
print("NIF oracle render ok:", sub.size, "pixels,", st2.casts, "casts")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  python $R/build/san/oracle_asan.py
