cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
mkdir -p gpurun_out/r5e
cp $P/libscl_hip.so /tmp/keep.so
for r in 1 2 3; do for v in base abl2 abl8 abl10 dabl1 dabl3; do
  if [ $v = base ]; then cp /tmp/keep.so $P/libscl_hip.so; else cp $P/libscl_hip_$v.so $P/libscl_hip.so; fi
  ABL=$v WARM=60 timeout 200 python tools/ablate_enc.py | tail -1
done; done 2>&1 | tee gpurun_out/r5e/ablations.txt
cp /tmp/keep.so $P/libscl_hip.so
