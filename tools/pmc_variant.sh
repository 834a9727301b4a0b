#!/bin/bash
# pmc_variant.sh VARIANT  -- hardware counters of the rANS kernels of one library build (libscl_hip_VARIANT.so, see
# build_variant.sh) on tools/ablate_enc.py; one rocprofv3 --pmc pass per counter group (never combined with tracing).
V=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
P=$REPO/stanford_compression_library_amd
cp $P/libscl_hip.so /tmp/keep_pmc.so
[ -n "$V" ] && cp $P/libscl_hip_$V.so $P/libscl_hip.so
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/pmc_raw_$V; rm -rf $RAW; mkdir -p $RAW
OUT=$REPO/gpurun_out/pmc_$V.txt; : > $OUT
i=0
SETS=${SETS:-"1 2 3 4 5 6 7 8"}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  case " $SETS " in *" $i "*) ;; *) continue;; esac
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $RAW/p$i -o pmc -- python $REPO/tools/ablate_enc.py > $RAW/p$i.log 2>&1
  find $RAW/p$i -name '*counter_collection.csv' -exec python3 $REPO/tools/summarize_pmc.py {} \; >> $OUT 2>&1
done
cp /tmp/keep_pmc.so $P/libscl_hip.so
grep -E "encode" $OUT | sed 's/void rans_encode_fast_kernel<0, 10>//' | awk '{print $1, $2}'
