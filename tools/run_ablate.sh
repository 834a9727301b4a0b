#!/bin/bash
P=stanford_compression_library_amd
python tools/ablate_enc.py
cp $P/libscl_hip.so /tmp/base.so
for a in 11 12 13; do cp $P/libscl_hip_abl$a.so $P/libscl_hip.so; ABL=abl$a python tools/ablate_enc.py; done
cp /tmp/base.so $P/libscl_hip.so
