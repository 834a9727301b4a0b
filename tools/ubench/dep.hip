// micro-benchmark 3: dependent-chain issue latency on gfx950 (wave64): how often can ONE wave issue an
// instruction that depends on its previous one, and how does that scale with waves per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
#define REP32(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x
#define DEF(NAME, ASM)                                                                 \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {        \
        uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = 5;                         \
        for (int i = 0; i < ITER; ++i) asm volatile(REP32(ASM) : "+v"(a), "+v"(b), "+v"(c)); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;                        \
    }
DEF(d_add,   "v_add_u32 %0, %0, %1\n")
DEF(d_lshr,  "v_lshrrev_b32 %0, 1, %0\n")
DEF(d_and,   "v_and_b32 %0, %0, %1\n")
DEF(d_mad24, "v_mad_u32_u24 %0, %0, %1, %2\n")
DEF(d_mulhi, "v_mul_hi_u32 %0, %0, %1\n")
DEF(d_bfe,   "v_bfe_u32 %0, %0, 1, %2\n")
DEF(d_add3,  "v_add3_u32 %0, %0, %1, %2\n")
DEF(d_ffbh,  "v_ffbh_u32 %0, %0\n")
DEF(d_mix,   "v_add_u32 %0, %0, %1\n v_lshrrev_b32 %0, 1, %0\n")   /* counts as 2 per ASM */
typedef void (*kern_t)(uint32_t *, uint32_t);
static void run(const char *name, kern_t fn, double per_asm, int waves_per_simd) {
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_wave = (double)ITER * 32 * per_asm;           // dependent instrs per wave
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-8s waves/SIMD=%d  %.3f ms  %.2f cyc per dependent instr per wave,  %.2f cyc/instr/SIMD\n", name, waves_per_simd, ms,
           cyc / per_wave, cyc / (per_wave * waves_per_simd));
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run("add", d_add, 1, w); run("lshr", d_lshr, 1, w); run("and", d_and, 1, w); run("mad24", d_mad24, 1, w);
        run("mulhi", d_mulhi, 1, w); run("bfe", d_bfe, 1, w); run("add3", d_add3, 1, w); run("ffbh", d_ffbh, 1, w);
        run("add+lshr", d_mix, 2, w);
    }
    return 0;
}
