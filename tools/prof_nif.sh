#!/bin/bash
# rocprofv3 passes over the NIF MLP kernel alone (tools/bench_nif.py): kernel stats + the counters K3's record quotes.
#   tools/prof_nif.sh gpurun_out/<dir> [bench_nif.py arguments]
out=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/$out
(cd $R && python3 -c 'import __graft_entry__ as ge; ge.build()') || { echo "build failed"; exit 1; }
export MI_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o st -- python3 $R/tools/bench_nif.py "$@" > $R/$out/stats.log 2>&1 || echo "stats failed"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $grp -d $R/$out/g$i -o pmc -- python3 $R/tools/bench_nif.py "$@" > $R/$out/g$i.log 2>&1 || echo "group $i failed"
done
echo done
