// lds_write_unaligned.hip -- what does a ds_write_b32 to a byte address that is NOT a multiple of four cost on gfx950?
// (round 5: the range encoder releases 0..3 bytes per symbol; writing bswap(low) as one 4-byte store at the lane's BYTE
// position in its ring -- the bytes beyond the released ones are overwritten by the next symbol's store -- would replace
// its byte accumulator, the pair merge and the word-completed branch.)  [thread][256 B] rings as in AnsBackWriterL.
//   hipcc --offload-arch=gfx950 -O3 -o lds_write_unaligned lds_write_unaligned.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
typedef u32 __attribute__((aligned(1))) u32_u;
#define ITERS 2048
// MODE 0: aligned word offsets, random per lane; 1: random BYTE offsets (3 of 4 unaligned); 2: a byte cursor per lane that
// advances by 1 (70 %), 0 (15 %) or 2 (15 %) per store, random start -- the range encoder's pattern; 3: the same cursor,
// stores of 8 bytes (ds_write_b64)
template <int MODE>
__global__ __launch_bounds__(256) void k_write(u32 *out, u32 seed) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const u32 base = threadIdx.x * 256u;
    u32 s = seed + threadIdx.x * 747796405u + blockIdx.x;
    u32 pos = (s >> 9) & ((MODE == 0 || MODE == 4) ? 252u : 255u);  // modes 0 and 4 stay on word boundaries
    u32 step[8];
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        const u32 r = (s >> 13) % 100u;
        step[j] = (MODE == 2 || MODE == 3 || MODE == 6) ? (r < 70 ? 1u : (r < 85 ? 0u : 2u)) : ((MODE == 1 || MODE == 5) ? ((s >> 11) & 255u) : ((s >> 11) & 252u));
    }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pos = (MODE == 2 || MODE == 3 || MODE == 6) ? ((pos + step[j]) & 255u) : ((pos ^ step[j]) & 255u);
            const u32 a = base | (MODE == 3 ? (pos & 247u) | (pos & 7u) : pos);  // (same address either way)
            const u32 lim = MODE == 3 ? min(a, base + 248u) : min(a, base + 252u);  // stay inside the lane's ring
            if (MODE == 3)
                asm volatile("ds_write_b64 %0, %1" : : "v"(lim), "v"((unsigned long long)i) : "memory");
            else if (MODE == 4)
                { *(u32 *)(lds + (lim & ~3u)) = (u32)i; asm volatile("" ::: "memory"); }
            else if (MODE == 5 || MODE == 6)
                { *(u32_u *)(lds + lim) = (u32)i; asm volatile("" ::: "memory"); }
            else
                asm volatile("ds_write_b32 %0, %1" : : "v"(lim), "v"((u32)i) : "memory");
        }
    }
    __syncthreads();
    if (((u32 *)lds)[threadIdx.x] == 0x1234567u) out[threadIdx.x] = 1;
}
template <typename K>
static void run(const char *name, K kern, u32 *d, int waves) {
    const int blocks = 256 * waves, ldsbytes = 65536;
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
    printf("%-58s %7.3f ms  %6.2f ns per wave-write per CU\n", name, ms, ms * 1e6 / ((double)waves * 4 * ITERS * 8));
}
int main() {
    u32 *d;
    (void)hipMalloc(&d, 4096);
    const int w = 2;
    run("ds_write_b32, aligned random word of a 256-byte ring", k_write<0>, d, w);
    run("ds_write_b32, random BYTE offset", k_write<1>, d, w);
    run("ds_write_b32, byte cursor +0/1/2 per store", k_write<2>, d, w);
    run("ds_write_b64, byte cursor +0/1/2 per store", k_write<3>, d, w);
    run("compiler-made store, aligned random word", k_write<4>, d, w);
    run("compiler-made store of an align-1 u32, random byte", k_write<5>, d, w);
    run("compiler-made store of an align-1 u32, byte cursor", k_write<6>, d, w);
    return 0;
}
