// seq_rate.hip -- the per-symbol VALU sequences of the rANS encoder as straight-line loops (no LDS, no memory):
// what the SIMD alone needs per symbol at 1, 2, 3, 4 waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
#define ITERS 8192

#define SEQ_OLD                                                                                                     \
    "v_sub_u32 %[t], %[x], %[e1]\n\t"                                                                               \
    "v_lshrrev_b32 %[t], 31, %[t]\n\t"                                                                              \
    "v_sub_u32_sdwa %[k], %[e3], %[t] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"       \
    "v_mul_hi_u32 %[q], %[x], %[e0]\n\t"                                                                            \
    "v_sub_u32 %[t], 10, %[t]\n\t"                                                                                  \
    "v_lshrrev_b32 %[q], %[t], %[q]\n\t"                                                                            \
    "v_alignbit_b32 %[lo], %[hi], %[lo], %[k]\n\t"                                                                  \
    "v_alignbit_b32 %[hi], %[x], %[hi], %[k]\n\t"                                                                   \
    "v_lshrrev_b32 %[t], %[k], %[x]\n\t"                                                                            \
    "v_add_u32 %[t], %[t], %[e2]\n\t"                                                                               \
    "v_mad_u32_u24 %[x], %[q], %[e3], %[t]\n\t"

#define SEQ_NEW                                                                                                     \
    "v_mul_hi_u32 %[q], %[x], %[e0]\n\t"                                                                            \
    "v_lshrrev_b32 %[t], 26, %[q]\n\t"                                                                              \
    "v_lshrrev_b32 %[q], 9, %[q]\n\t"                                                                               \
    "v_add_u32_sdwa %[k], %[t], %[e3] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"       \
    "v_lshrrev_b32 %[q], %[t], %[q]\n\t"                                                                            \
    "v_alignbit_b32 %[lo], %[hi], %[lo], %[k]\n\t"                                                                  \
    "v_alignbit_b32 %[hi], %[x], %[hi], %[k]\n\t"                                                                   \
    "v_lshrrev_b32 %[t], %[k], %[x]\n\t"                                                                            \
    "v_add_u32 %[t], %[t], %[e2]\n\t"                                                                               \
    "v_mad_u32_u24 %[x], %[q], %[e3], %[t]\n\t"

#define KERNEL(name, SEQ)                                                                                 \
    __global__ __launch_bounds__(256) void k_##name(u32 *out, u32 a, u32 b, u32 c, u32 d) {               \
        u32 x = a + threadIdx.x, hi = 0, lo = 0, t = 0, k = 0, q = 0;                                     \
        u32 e0 = b | 0x80000000u, e1 = c << 17, e2 = d & 4095u, e3 = (4096u - c) | (12u << 24);           \
        for (int i = 0; i < ITERS; ++i) {                                                                 \
            asm volatile(SEQ SEQ SEQ SEQ                                                                  \
                         : [x] "+v"(x), [hi] "+v"(hi), [lo] "+v"(lo), [t] "+v"(t), [k] "+v"(k), [q] "+v"(q) \
                         : [e0] "v"(e0), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3));                       \
        }                                                                                                 \
        if ((x ^ hi ^ lo) == 0x12345u) out[threadIdx.x] = x;                                              \
    }
KERNEL(old, SEQ_OLD)
KERNEL(new, SEQ_NEW)

template <typename K>
static void run(const char *name, K kern, u32 *d, int waves) {
    const int blocks = 256 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 1, 3, 100, 7);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, 256>>>(d, 1, 3, 100, 7);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double wave_syms_per_simd = (double)waves * ITERS * 4;
    printf("%-6s waves/SIMD %d: %7.3f ms  %6.2f ns per wave-symbol per SIMD\n", name, waves, ms,
           ms * 1e6 / wave_syms_per_simd);
}

int main() {
    u32 *d;
    (void)hipMalloc(&d, 4096);
    for (int w = 1; w <= 4; ++w) {
        run("old", k_old, d, w);
        run("new", k_new, d, w);
    }
    return 0;
}
