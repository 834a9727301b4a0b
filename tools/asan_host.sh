#!/bin/bash
# asan_host.sh -- host-side sanitizer pass (SURVEY 5 / VERDICT r2 #7).  The image ships no compute-sanitizer equivalent and
# no device ASan runtime (/opt/rocm/lib/asan is absent), so the DEVICE side is checked by construction
# (tests/test_gpu_guard_bands.py: guard bands around every output buffer) and the HOST side here:
#   1. libscl_hip.so rebuilt with its host code under AddressSanitizer + UndefinedBehaviorSanitizer.  On this image the
#      ASan runtime's HSA interceptor fails inside hipInit ("out of memory ... hsa_amd_memory_pool_allocate", the 20 TB
#      shadow reservation is not available in the container) -- recorded, then
#   2. the same tests under UBSan alone (-fno-sanitize-recover: any report aborts) with glibc's heap checker
#      (MALLOC_CHECK_=3, MALLOC_PERTURB_) watching the host allocations.
# Run on the GPU box:  bash tools/asan_host.sh  ->  gpurun_out/asan_host.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=stanford_compression_library_amd
HIPCC=/opt/rocm/bin/hipcc
TESTS="tests/test_abi.py tests/test_gpu_goldens.py tests/test_gpu_stream_goldens.py tests/test_gpu_sharded.py tests/test_gpu_guard_bands.py tests/test_gpu_batch.py"
SEL="not full_occupancy and not full_size and not random_"
cp $P/libscl_hip.so /tmp/keep_asan.so
mkdir -p gpurun_out; : > gpurun_out/asan_host.txt
build() {  # $1 = sanitizer list
  B=/tmp/scl_san; rm -rf $B; mkdir -p $B
  for f in $P/csrc/scl_*.hip; do
    ( $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer -Xarch_host -fsanitize=$1 \
        -Xarch_host -fno-sanitize-recover=undefined -I$P/csrc -c $f -o $B/$(basename ${f%.hip}).o ) &
  done; wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=$1 -o $P/libscl_hip.so $B/*.o -ldl
}
{
echo "== pass 1: host ASan + UBSan =="
build address,undefined || echo "build failed"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
rm -f /tmp/asan_log*
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:log_path=/tmp/asan_log UBSAN_OPTIONS=print_stacktrace=1 \
  LD_PRELOAD=$RT timeout 900 python -m pytest $TESTS -q -x -k "$SEL" -p no:cacheprovider > /tmp/asan_pytest.txt 2>&1
echo "pytest exit code: $?"; grep -E "passed|failed|error" /tmp/asan_pytest.txt | tail -3
echo "sanitizer report files: $(ls /tmp/asan_log* 2>/dev/null | wc -l)"; cat /tmp/asan_log* 2>/dev/null | grep -E "ERROR|SUMMARY|#0|#1" | head -8
echo
echo "== pass 2: host UBSan (abort on any report) + glibc heap checker =="
build undefined || echo "build failed"
RTU=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165 LD_PRELOAD=$RTU \
  timeout 1500 python -m pytest $TESTS -q -x -k "$SEL" -p no:cacheprovider > /tmp/ubsan_pytest.txt 2>&1
echo "pytest exit code: $?"; grep -E "passed|failed|error" /tmp/ubsan_pytest.txt | tail -3
echo "runtime error reports: $(grep -c 'runtime error' /tmp/ubsan_pytest.txt)"; grep "runtime error" /tmp/ubsan_pytest.txt | head -10
nm -D $P/libscl_hip.so | grep -c __ubsan | sed 's/^/ubsan symbols referenced by the tested library: /'
} >> gpurun_out/asan_host.txt 2>&1
cp /tmp/keep_asan.so $P/libscl_hip.so
cat gpurun_out/asan_host.txt
