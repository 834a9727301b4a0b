cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
cp $P/libscl_hip.so /tmp/keep.so
for v in newb r192b newb3f; do cp $P/libscl_hip_$v.so $P/libscl_hip.so; for n in 131072 196608 262144 393216 524288; do NCHUNKS=$n ABL=$v timeout 200 python tools/ablate_enc.py 2>/dev/null | tail -1; done; done
cp /tmp/keep.so $P/libscl_hip.so
