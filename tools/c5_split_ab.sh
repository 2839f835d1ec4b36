#!/bin/bash
# tools/c5_split_ab.sh [spp] - config 5 (monkey + NIF) with the trace launches on compute units of their own (scene option nif_split = units
# of the trace stream; 0 = the launches share the chip), same box, one process per setting, the unsplit form first and last.
set -e
spp=${1:-1024}
for o in "" nif_split=8 nif_split=16 nif_split=24 nif_split=32 ""; do
  echo "== opts '$o'"
  timeout -k 10 180 python tools/bench_config5.py $spp --steps 2 --warmup 1 --opts "$o" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print({k:r[k] for k in ('ms_per_frame','mlp_ms_per_frame','rgb_sum')})"
done
