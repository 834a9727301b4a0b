#!/bin/bash
# rocprofv3 recipe for the headline bench (run on the GPU box through gpurun).
# Writes small text summaries to gpurun_out/prof/ (raw traces stay in /tmp on the box).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
RAW=/tmp/prof_raw
rm -rf $RAW; mkdir -p $OUT $RAW
# the kernel trace runs bench.py's default step counts (50 timed after 20 warm-up: the durations it reports are the ones
# bench.py prints); the counter passes only need a few dispatches
TRACE_CMD="python $REPO/bench.py --no-cpu-baseline ${BENCH_ARGS}"
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $TRACE_CMD > $OUT/trace.log 2>&1
find $RAW/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $RAW/trace -name '*kernel_trace.csv' -exec python3 $REPO/tools/summarize_trace.py {} $OUT/kernel_trace_summary.txt \;
# PMC passes, each in its own run (never combined with tracing)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $RAW/pmc$i -o pmc -- $CMD > $RAW/pmc$i.log 2>&1
  find $RAW/pmc$i -name '*counter_collection.csv' -exec python3 $REPO/tools/summarize_pmc.py {} \; >> $OUT/pmc_summary.txt 2>&1
done
tail -3 $OUT/trace.log
cat $OUT/kernel_trace_summary.txt
cat $OUT/pmc_summary.txt
python3 $REPO/tools/make_traffic_json.py $OUT/pmc_summary.txt $OUT/traffic.json
