// lds_write.hip -- ds_write_b32 throughput for the encoder's ring layouts: lane stride 256 vs 192 bytes, per-lane
// offsets random or in lockstep (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
#define ITERS 2048
template <int STRIDE, int WORDS, int LOCKSTEP>
__global__ __launch_bounds__(256) void k_write(u32 *out, u32 seed) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    u32 a[8];
    u32 s = seed + (LOCKSTEP ? 0u : threadIdx.x * 747796405u) + blockIdx.x;
    const u32 rot = LOCKSTEP == 2 ? (STRIDE == 192 ? 16u * ((5u * (threadIdx.x >> 2)) % 12u) : 16u * (threadIdx.x & 15u)) : 0u;
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        a[j] = threadIdx.x * STRIDE + (((s >> 11) % WORDS) * 4 + rot) % STRIDE;
    }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *(u32 *)(lds + a[j]) = i;
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (((u32 *)lds)[threadIdx.x] == 0x1234567u) out[threadIdx.x] = 1;
}
template <typename K>
static void run(const char *name, K kern, u32 *d, int waves, int ldsbytes) {
    const int blocks = 256 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsbytes);
    kern<<<blocks, 256, ldsbytes>>>(d, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, 256, ldsbytes>>>(d, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-40s %7.3f ms  %6.2f ns per wave-write per CU\n", name, ms, ms * 1e6 / ((double)waves * 4 * ITERS * 8));
}
int main() {
    u32 *d;
    (void)hipMalloc(&d, 4096);
    const int w = 2;
    run("stride 256, random offsets", k_write<256, 64, 0>, d, w, 65536);
    run("stride 192, random offsets", k_write<192, 48, 0>, d, w, 49152);
    run("stride 256, lockstep", k_write<256, 64, 1>, d, w, 65536);
    run("stride 192, lockstep", k_write<192, 48, 1>, d, w, 49152);
    run("stride 256, lockstep + rotation", k_write<256, 64, 2>, d, w, 65536);
    run("stride 192, lockstep + rotation", k_write<192, 48, 2>, d, w, 49152);
    return 0;
}
