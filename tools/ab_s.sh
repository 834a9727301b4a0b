#!/bin/bash
# time the rANS encoder with either writer (SCL_RANS_ENC_WRITER=L|S) for the builds in VARIANTS, on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
cp $P/libscl_hip.so /tmp/keep.so
for r in 1 2; do
for v in $VARIANTS; do cp $P/libscl_hip_$v.so $P/libscl_hip.so; for w in ${WRITERS:-L S}; do SCL_RANS_ENC_WRITER=$w ABL=$v-$w python tools/ablate_enc.py 2>/dev/null | tail -1; done; done; done
cp /tmp/keep.so $P/libscl_hip.so
