// occ.hip -- resident 256-lane workgroups per CU as a function of their LDS size (runtime's occupancy calculator and a
// direct measurement: blocks spin until every block of a one-wave-per-SIMD-per-block launch has started).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(unsigned *counter, unsigned target, unsigned *ok) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(counter, 1u);
        // wait (bounded) until `target` blocks are resident at once
        for (long i = 0; i < 20000000; ++i) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
                *ok = 1;
                break;
            }
        }
    }
}
int main() {
    unsigned *c, *ok;
    (void)hipMalloc(&c, 4);
    (void)hipMalloc(&ok, 4);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int kib : {32, 36, 40, 48, 50, 52, 53, 54, 64, 80}) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, kib * 1024);
        int resident = 0;
        for (int per_cu = 1; per_cu <= 6; ++per_cu) {
            unsigned z = 0, h = 0;
            (void)hipMemcpy(c, &z, 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(ok, &z, 4, hipMemcpyHostToDevice);
            k<<<256 * per_cu, 256, kib * 1024>>>(c, 256 * per_cu, ok);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&h, ok, 4, hipMemcpyDeviceToHost);
            if (h) resident = per_cu; else break;
        }
        printf("LDS %3d KiB per workgroup: calculator %d, measured >= %d workgroups per CU\n", kib, nb, resident);
    }
    return 0;
}
