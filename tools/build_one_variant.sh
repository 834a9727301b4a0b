#!/bin/bash
# build_one_variant.sh NAME FILE.hip "EXTRA_FLAGS" -> libscl_hip_NAME.so = the current build with FILE rebuilt with EXTRA_FLAGS
set -e
NAME=$1; F=$2; EXTRA=$3
cd "$(dirname "$0")/../stanford_compression_library_amd/csrc"
make -s -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
o=/tmp/variant_${NAME}_${F%.hip}.o
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA -c $F -o $o
objs=$(ls _build/*.o | grep -v "_build/${F%.hip}.o")
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libscl_hip_$NAME.so $objs $o -ldl
echo built libscl_hip_$NAME.so
