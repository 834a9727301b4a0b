#!/bin/bash
# timing experiments on the LDS-table arithmetic coder (run on the GPU box): builds -DAF_ABLATE=n variants of
# scl_aec_fast.hip and times encode / decode of 65536 x 4 KiB order-1 K=16 chunks (outputs are NOT valid streams)
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
cp $P/libscl_hip.so /tmp/base.so
for a in 0 ${ABLS:-1}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAF_ABLATE=$a -Iinclude -c $P/csrc/scl_aec_fast.hip -o /tmp/abl.o || exit 1
  objs=$(ls $P/csrc/_build/*.o | grep -v scl_aec_fast)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libscl_hip.so $objs /tmp/abl.o || exit 1
  ABL=$a python tools/ablate_aec.py
done
cp /tmp/base.so $P/libscl_hip.so
