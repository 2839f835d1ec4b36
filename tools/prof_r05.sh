#!/bin/bash
# Round-5 profile collection on one MI355X (one gpurun call). Counters-only passes, one --pmc group per run; kernel
# times from separate --kernel-trace --stats runs. Everything lands under gpurun_out/prof_r05/; tools/collect_profiles.py
# turns it into the files under profiles/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r05
mkdir -p $OUT
# One un-profiled build first, then no compiling inside a profiled process: under `rocprofv3 --pmc` the preloaded tool has
# initialised the GPU before python starts, and a compiler child (make -> sh -> gcc, hipcc -> clang) would be an exec after
# GPU init, which the pool forbids. MI_NO_BUILD=1 turns every build_* helper into "fail fast on a stale binary".
(cd $R && python3 -c 'import __graft_entry__ as ge; ge.build()') || { echo "build failed"; exit 1; }
export MI_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 3 --warmup 1 --no-cpu-baseline --no-extras"
GROUPS_K1=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum")

run_set() {   # tag, then the program and its arguments
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/stats -o st -- "$@" > $OUT/${tag}_stats.log 2>&1 || echo "$tag: stats failed"
  local i=0
  for grp in "${GROUPS_K1[@]}"; do
    i=$((i+1))
    rocprofv3 --output-format csv --pmc $grp -d $OUT/$tag/g$i -o pmc -- "$@" > $OUT/${tag}_g$i.log 2>&1 || echo "$tag: group $i ($grp) failed"
  done
}

# config 2 (headline): bench.py itself
run_set c2 python3 $R/bench.py $BENCH
if [ "$ONLY" = c2 ]; then      # (re-collection of the headline set alone, e.g. after a change that only touches K1w's launch)
  python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
  echo done; exit 0
fi
# config 3 scene (test_scene.dae with vertex normals) and the 16-spp box frame: one timed launch each
run_set c3 python3 $R/tools/k_sweep.py --file $R/assets/test_scene.dae --spp 4000 --reps 1 kernel=1
run_set spp16 python3 $R/tools/k_sweep.py --spp 16 --reps 8 kernel=1
# K3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nif/stats -o st -- python3 $R/tools/bench_nif.py --shape auto --reps 20 > $OUT/nif_stats.log 2>&1 || echo "nif stats failed"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $grp -d $OUT/nif/g$i -o pmc -- python3 $R/tools/bench_nif.py --shape auto --reps 20 > $OUT/nif_g$i.log 2>&1 || echo "nif group $i failed"
done
# nif_mlp_kernel (w6), the default up to round 4, next to K3a (the default now: the "nif" set above) - kernel stats + counters of the same workload
$R/tools/prof_nif.sh gpurun_out/prof_r05/nif_w6 --shape w6 --reps 20 > $OUT/nif_w6.log 2>&1 || echo "nif w6 failed"
# config 5 at its real size on one GPU: monkey + NIF, 1440^2 x 4000 spp, un-profiled for the wall time and with --stats for the kernel shares
python3 $R/tools/bench_config5.py 4000 > $OUT/c5_full.json 2> $OUT/c5_full.err || echo "config 5 run failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5/stats -o st -- python3 $R/tools/bench_config5.py 1024 > $OUT/c5_stats.log 2>&1 || echo "c5 stats failed"
# the node-gather microbenchmark (the measured roof K1w's record quotes): the table, and PMC passes at saturation
python3 $R/tools/gather_probe.py --out $OUT/gather_probe.json > $OUT/gather_probe.txt 2>&1 || echo "gather probe failed"
for cfg in "tree35 0,1,35,6" "uni35 0,0,35,6" "tree64 0,1,64,6" "lds35 1,1,35,6"; do
  set -- $cfg
  $R/tools/prof_gather_probe.sh gpurun_out/prof_r05/probe_$1 $2
done
# the tolerance tier and the double-fallback variant next to the default kernel, same box
python3 $R/tools/k_sweep.py --reps 3 kernel=1 fast=1 double_fallback=1 kernel=1 fast=1 > $OUT/variants.txt 2>&1 || echo "variants failed"
# the un-profiled bench line of the same build
python3 $R/bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
echo done
