#!/bin/bash
# ab_variants.sh V1 V2 ... : same-box timing of libscl_hip_<V>.so builds against the shipped library ("base"), three rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
cp $P/libscl_hip.so /tmp/keep.so
for r in 1 2 3; do for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/keep.so $P/libscl_hip.so; else cp $P/libscl_hip_$v.so $P/libscl_hip.so; fi
  ABL=$v WARM=${WARM:-60} timeout 200 ${CMD:-python tools/ablate_enc.py} 2>/dev/null | tail -1
done; done
cp /tmp/keep.so $P/libscl_hip.so
