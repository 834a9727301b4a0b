#!/bin/bash
# bench_brief on the uniform table and on T256 for the builds in VARIANTS, on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
cp $P/libscl_hip.so /tmp/keep.so
for r in 1 2; do for v in $VARIANTS; do cp $P/libscl_hip_$v.so $P/libscl_hip.so; echo "$v uniform: $(bash tools/bench_brief.sh --table uniform)"; echo "$v t256:    $(bash tools/bench_brief.sh)"; done; done
cp /tmp/keep.so $P/libscl_hip.so
