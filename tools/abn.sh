#!/bin/bash
# abn.sh "v1 v2 ..." : same-box timing of several builds libscl_hip_<v>.so (tools/build_one_variant.sh); CMD = what to time
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
CMD=${CMD:-python tools/ablate_aec.py}
cp $P/libscl_hip.so /tmp/keep_abn.so
for r in $(seq ${REPS:-2}); do for v in $1; do cp $P/libscl_hip_$v.so $P/libscl_hip.so; echo "$v: $(ABL=$v timeout 300 $CMD 2>&1 | tail -1)"; done; done
cp /tmp/keep_abn.so $P/libscl_hip.so
