#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $grp -d $R/gpurun_out/c5pmc/g$i -o pmc -- python3 $R/tools/bench_config5.py 8 > $R/gpurun_out/c5pmc_g$i.log 2>&1 || echo "group $i ($grp) failed"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(list)
for f in glob.glob(R+"/gpurun_out/c5pmc/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "path_trace_wavefront_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v), len(v))
PY
