// lds_rate.hip -- throughput of ds_read_b128 / ds_read_b64 / ds_read_b32 table lookups per CU for different address
// patterns (gfx950).  hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip && ./lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
#define ITERS 2048

template <int BYTES, int PATTERN>  // PATTERN 0: linear (lane i -> entry i), 1: random entries, 2: all lanes one entry
__global__ __launch_bounds__(256) void k_read(u32 *out, u32 seed, u32 entries) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (u32 i = threadIdx.x; i < entries * BYTES / 4; i += 256) ((u32 *)lds)[i] = i * 2654435761u;
    __syncthreads();
    u32 idx[8];
    u32 s = seed + threadIdx.x * 747796405u + blockIdx.x;
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        u32 e = PATTERN == 0 ? (threadIdx.x + j * 7) : PATTERN == 1 ? (s >> 11) : j;
        idx[j] = (e % entries) * BYTES;
    }
    u32 acc = 0;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (BYTES == 16) {
                uint4 v = *(const uint4 *)(lds + idx[j]);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else if (BYTES == 8) {
                uint2 v = *(const uint2 *)(lds + idx[j]);
                acc += v.x ^ v.y;
            } else {
                acc += *(const u32 *)(lds + idx[j]);
            }
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 0x1234567u) out[threadIdx.x] = acc;
}

template <typename K>
static void run(const char *name, K kern, u32 *d, int waves, int bytes, int entries) {
    const int blocks = 256 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kern<<<blocks, 256, entries * bytes>>>(d, 1, entries);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, 256, entries * bytes>>>(d, 1, entries);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double per_cu = (double)waves * 4 * ITERS * 8;  // wave-level read instructions per CU
    printf("%-28s waves/SIMD %d: %7.3f ms  %6.2f ns per wave-read per CU  (%5.1f B/ns/CU)\n", name, waves, ms,
           ms * 1e6 / per_cu, 64.0 * bytes / (ms * 1e6 / per_cu));
}

int main() {
    u32 *d;
    (void)hipMalloc(&d, 4096);
    for (int w = 2; w <= 8; w *= 2) {
        run("b128 linear  256 entries", k_read<16, 0>, d, w, 16, 256);
        run("b128 random  256 entries", k_read<16, 1>, d, w, 16, 256);
        run("b128 one entry", k_read<16, 2>, d, w, 16, 256);
        run("b64  linear  256 entries", k_read<8, 0>, d, w, 8, 256);
        run("b64  random  256 entries", k_read<8, 1>, d, w, 8, 256);
        run("b32  linear  256 entries", k_read<4, 0>, d, w, 4, 256);
        run("b32  random  256 entries", k_read<4, 1>, d, w, 4, 256);
        run("b32  random 4096 entries", k_read<4, 1>, d, w, 4, 4096);
        run("b64  random 4096 entries", k_read<8, 1>, d, w, 8, 4096);
    }
    return 0;
}
