#!/bin/bash
# Timing-only builds of K3r (nif_regs_kernel.hpp, MI_NIF_REGS_KO) next to the real one, all on ONE box:
#   tools/k3r_knockouts.sh build     (here: cross-compiles build/ko/libmi_raylib_ko<N>.so)
#   tools/k3r_knockouts.sh run       (on the GPU box: times each with tools/bench_nif.py)
R=$(cd $(dirname $0)/.. && pwd)
KOS="0 1 2 3 4 8 15"
if [ "$1" = build ]; then
  mkdir -p $R/build/ko
  for k in $KOS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -pragma-unroll-threshold=10000000 -fPIC -shared -Wall -Wno-unused-function \
      -DMI_RAYLIB_VARIANTS=1 -DMI_NIF_REGS_KO=$k -I $R/include -o $R/build/ko/libmi_raylib_ko$k.so $R/ipu_ray_lib_amd/csrc/raylib.hip 2>&1 | grep -E "error" &
  done
  wait
else
  for k in $KOS; do
    echo -n "KO $k: "
    MI_NO_BUILD=1 MI_RAYLIB_VARIANTS_LIB=$R/build/ko/libmi_raylib_ko$k.so timeout -k 10 120 python3 $R/tools/bench_nif.py --shape ${SHAPE:-r8} --reps 20 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms" % d["ms"])'
  done
fi
