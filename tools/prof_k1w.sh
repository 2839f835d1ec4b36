#!/bin/bash
# rocprofv3 passes for the path-trace kernel at BASELINE config 2 (one --pmc group per run; counters only).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/k1wprof
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k1w -- python3 $R/bench.py $ARGS > $R/gpurun_out/k1w_stats.log 2>&1 || echo "stats failed"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python3 $R/bench.py $ARGS > $R/gpurun_out/k1w_pmc_g$i.log 2>&1 || echo "group $i ($grp) failed"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(list)
for f in glob.glob(R+"/gpurun_out/k1wprof/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "path_trace_wavefront_kernel<false" in row["Kernel_Name"] or "path_trace_wavefront_kernelILb0" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v), len(v))
PY
