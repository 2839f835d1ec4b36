#!/bin/bash
# tools/prof_c5_trace.sh OUTDIR - counters of config 5's trace launch (monkey + NIF, 256 spp = two launches of 128 samples), one --pmc group per run
R=$GRAFT_REPO_ROOT
OUT=$R/$1
mkdir -p $OUT
export MI_NO_BUILD=1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" "TCC_ATOMIC_sum TCC_WRITE_sum TCC_READ_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $grp -d $OUT/g$i -o pmc -- python3 $R/tools/bench_config5.py 256 --steps 1 --warmup 0 > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed"
done
echo done
