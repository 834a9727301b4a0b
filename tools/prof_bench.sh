#!/bin/bash
# prof_bench.sh NAME  -- rocprofv3 evidence for one bench.py workload (BENCH_ARGS), run on the GPU box through gpurun.
# Writes gpurun_out/prof_NAME/{bench.json, kernel_trace_summary.txt, kernel_stats.csv, pmc_summary.txt, traffic_entry.json};
# raw traces stay in /tmp on the box.  The kernel trace runs bench.py's default step counts (the durations it reports are
# the ones bench.py prints); each PMC group is its own rocprofv3 run (never combined with tracing).
NAME=$1
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$NAME
RAW=/tmp/prof_raw_$NAME
rm -rf $RAW $OUT; mkdir -p $OUT $RAW
python $REPO/bench.py --full-line --no-cpu-baseline --no-other-configs ${BENCH_ARGS} 2>/dev/null | grep '^{' > $OUT/bench.json
# (--no-dense-pipeline: that measurement launches the encode kernel on half batches; per-kernel means must not mix them in)
TRACE_CMD="python $REPO/bench.py --no-cpu-baseline --no-other-configs --no-dense-pipeline ${BENCH_ARGS}"
CMD="python $REPO/bench.py --steps 3 --warmup 1 --min-warm-ms 0 --no-cpu-baseline --no-other-configs --no-dense-pipeline ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $TRACE_CMD > $OUT/trace.log 2>&1
find $RAW/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $RAW/trace -name '*kernel_trace.csv' -exec python3 $REPO/tools/summarize_trace.py {} $OUT/kernel_trace_summary.txt \;
grep -v -E "at::native|rocprim|Cijk|elementwise|distribution|index_" $OUT/kernel_trace_summary.txt > $OUT/k.txt; mv $OUT/k.txt $OUT/kernel_trace_summary.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  # counters only for the library's kernels: the device-side data generators of bench.py launch thousands of tiny torch
  # kernels, and a counter pass over all of them does not finish
  rocprofv3 --pmc $set --kernel-include-regex "rans_|tans_|range_|aec_|cp_" --output-format csv -d $RAW/pmc$i -o pmc -- $CMD > $RAW/pmc$i.log 2>&1
  [ -d $RAW/pmc$i ] || { echo "pass $i ($set) produced nothing:"; tail -3 $RAW/pmc$i.log; } >> $OUT/pmc_errors.txt
  find $RAW/pmc$i -name '*counter_collection.csv' -exec python3 $REPO/tools/summarize_pmc.py {} \; >> $OUT/pmc_summary.txt 2>&1
done
python3 $REPO/tools/make_traffic_json.py $OUT/pmc_summary.txt $OUT/bench.json $OUT/traffic_entry.json
rm -f $OUT/trace.log
cat $OUT/kernel_trace_summary.txt | head -6
