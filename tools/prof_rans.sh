#!/bin/bash
# rocprofv3 recipe for the headline bench (run on the GPU box through gpurun; outputs under gpurun_out/prof)
set -x
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
# PMC passes, each in its own run (never combined with tracing)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TA_BUSY_avr TD_BUSY_avr" "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pmc$i -o pmc -- $CMD > $OUT/pmc$i.log 2>&1
done
rocprofv3 -L > $OUT/counters_list.txt 2>&1
ls -R $OUT | head -50
