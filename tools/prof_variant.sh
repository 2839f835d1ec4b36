#!/bin/bash
# rocprofv3 counter passes for one path-trace kernel variant on the headline frame (counters only, one group per run).
#   tools/prof_variant.sh <tag> <spec> [extra k_sweep args]       e.g.  tools/prof_variant.sh k3 kernel=3
R=$GRAFT_REPO_ROOT
TAG=$1; SPEC=$2; shift 2
OUT=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python3 $R/tools/k_sweep.py --reps 1 "$@" $SPEC > $R/gpurun_out/prof_${TAG}_g$i.log 2>&1 || echo "group $i ($grp) failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "path_trace" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:34s} {sum(v) / len(v):.6g}  ({len(v)} launches)")
PY
