#!/bin/bash
# Re-sweep of K1w's scheduling weights on the round-3 kernel (runtime-weight instantiation; the first line is the default set).
out=$1
python tools/k_sweep.py --reps 2 tune=8,16,24,48,3,4,5 tune=8,16,24,48,3,3,5 tune=8,16,24,48,3,5,5 tune=8,16,24,48,3,6,5 tune=8,16,24,48,3,4,4 tune=8,16,24,48,3,4,3 \
  tune=6,16,24,48,3,4,5 tune=10,16,24,48,3,4,5 tune=12,16,24,48,3,4,5 tune=8,12,24,48,3,4,5 tune=8,20,24,48,3,4,5 tune=8,16,16,48,3,4,5 tune=8,16,32,48,3,4,5 \
  tune=8,16,24,32,3,4,5 tune=8,16,24,64,3,4,5 tune=8,16,24,96,3,4,5 tune=8,16,24,48,2,4,5 tune=8,16,24,48,4,4,5 tune=8,16,24,48,3,4,5,0 tune=8,16,24,48,3,4,5,1,0 kernel=1 > $out 2>&1
