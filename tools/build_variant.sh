#!/bin/bash
# build_variant.sh NAME "EXTRA_FLAGS"  ->  stanford_compression_library_amd/libscl_hip_NAME.so
# (a second build of the library with extra -D / -mllvm switches, for same-box A/B timing with tools/ab.sh)
set -e
NAME=$1; EXTRA=$2
cd "$(dirname "$0")/../stanford_compression_library_amd/csrc"
B=/tmp/scl_variant_$NAME; mkdir -p $B
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for f in scl_*.hip; do
  o=$B/${f%.hip}.o
  # only the files an experiment touches need the flags, but rebuilding all keeps the variant self-consistent
  ( $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA -c $f -o $o ) &
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libscl_hip_$NAME.so $B/*.o
echo built ../libscl_hip_$NAME.so
