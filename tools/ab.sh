#!/bin/bash
# A/B timing of two builds of the library on the same box: libscl_hip_old.so vs libscl_hip_new.so
# (box-to-box variation between gpurun machines is about +-4 %).  CMD = what to time, default the rANS kernels.
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
CMD=${CMD:-python tools/ablate_enc.py}
cp $P/libscl_hip.so /tmp/keep.so
for r in 1 2 3; do for v in old new; do cp $P/libscl_hip_$v.so $P/libscl_hip.so; echo -n "$v: "; ABL=$v timeout 200 $CMD | tail -1; done; done
cp /tmp/keep.so $P/libscl_hip.so
