#!/bin/bash
# time several builds of the library on the same box: VARIANTS="old new abl1" tools/ab_multi.sh  (see build_variant.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
CMD=${CMD:-python tools/ablate_enc.py}
cp $P/libscl_hip.so /tmp/keep.so
for r in 1 2; do for v in $VARIANTS; do cp $P/libscl_hip_$v.so $P/libscl_hip.so; ABL=$v timeout 200 $CMD 2>/dev/null | tail -1; done; done
cp /tmp/keep.so $P/libscl_hip.so
