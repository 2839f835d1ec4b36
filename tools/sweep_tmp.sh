specs=""
for r in 8 16 24 32; do for f in 4 16; do specs="$specs kernel=3:pool_tune=4,48,$r,$f,4,4"; done; done
for sw in 3 6; do for gw in 4 8; do specs="$specs kernel=3:pool_tune=4,48,16,8,$sw,$gw"; done; done
for la in 2 3 6 8; do specs="$specs kernel=3:pool_tune=$la,48,16,8,4,4"; done
for b in 24 96; do specs="$specs kernel=3:pool_tune=4,$b,16,8,4,4"; done
python tools/k_sweep.py --reps 1 kernel=1 $specs
