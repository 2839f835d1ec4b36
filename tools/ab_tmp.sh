#!/bin/bash
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/c5.log
for spl in 32 64 128; do
  for spp in 128 1024; do
    echo "== spl $spl spp $spp" >> gpurun_out/ab/c5.log
    MI_RAYLIB_NIF_SPL=$spl timeout -k 10 200 python tools/bench_config5.py $spp 2>/dev/null >> gpurun_out/ab/c5.log || exit 1
  done
done
