#!/bin/bash
# rocprofv3 passes over one configuration of the gather probe (tools/gather_probe.py --only path,walk,lanes,wg):
# kernel trace + the counters K1w's record quotes (TCP accesses, TA busy, the clock). Run on the GPU box:
#   tools/prof_gather_probe.sh gpurun_out/<dir> 0,1,35,6
out=$1; cfg=$2
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/$out
# One un-profiled build first, then no compiling inside a profiled process: under `rocprofv3 --pmc` the preloaded tool has
# initialised the GPU before python starts, and a compiler child (make -> sh -> gcc, hipcc -> clang) would be an exec after
# GPU init, which the pool forbids. MI_NO_BUILD=1 turns every build_* helper into "fail fast on a stale binary".
(cd $R && python3 -c 'import __graft_entry__ as ge; ge.build()') || { echo "build failed"; exit 1; }
export MI_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o st -- python3 $R/tools/gather_probe.py --only $cfg > $R/$out/stats.log 2>&1 || echo "stats failed"
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $grp -d $R/$out/g$i -o pmc -- python3 $R/tools/gather_probe.py --only $cfg > $R/$out/g$i.log 2>&1 || echo "group $i failed"
done
