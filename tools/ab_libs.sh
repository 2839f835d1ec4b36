#!/bin/bash
# A/B of several builds of libmi_raylib.so on one box: tools/ab_libs.sh OUT lib1.so lib2.so ... (each timed twice, interleaved)
out=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)" >> $out
    MI_RAYLIB_LIB=$lib python tools/k_sweep.py --reps 3 kernel=1 >> $out 2>&1 || exit 1
  done
done
