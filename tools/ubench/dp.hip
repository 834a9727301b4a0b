// micro-benchmark 4: issue cost of the binary64 / conversion / 64-bit-shift instructions the arithmetic-coder
// fast path leans on (gfx950, wave64), as a dependent chain and as two interleaved independent chains, at 1 and
// 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 512
#define REP32(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x
#define DEF(NAME, ASM)                                                                                 \
    __global__ void __launch_bounds__(256) NAME(double *out, uint32_t seed) {                           \
        double a = 1.0 + threadIdx.x * 1e-3 + seed, b = 0.999999, c = 1e-9, d = 2.0 + seed;             \
        uint32_t u = threadIdx.x + seed, v = 3;                                                         \
        float f = 1.5f + seed;                                                                          \
        for (int i = 0; i < ITER; ++i)                                                                  \
            asm volatile(REP32(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(u), "+v"(v), "+v"(f));    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + u + v + f;                         \
    }
DEF(k_fma_dep,   "v_fma_f64 %0, %0, %1, %2\n")
DEF(k_fma_2ch,   "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %3, %3, %1, %2\n")
DEF(k_mul_dep,   "v_mul_f64 %0, %0, %1\n")
DEF(k_add_dep,   "v_add_f64 %0, %0, %2\n")
DEF(k_cvt_u2d,   "v_cvt_f64_u32 %0, %4\n v_add_u32 %4, %4, %5\n")
DEF(k_cvt_d2u,   "v_cvt_u32_f64 %4, %0\n")
DEF(k_cvt_rt,    "v_cvt_u32_f64 %4, %0\n v_cvt_f64_u32 %0, %4\n")
DEF(k_cvt_f2d,   "v_cvt_f64_f32 %0, %6\n")
DEF(k_rcp_f32,   "v_rcp_f32 %6, %6\n")
DEF(k_lshl64,    "v_lshlrev_b64 %0, 1, %0\n")
DEF(k_fma32_dep, "v_fma_f32 %6, %6, %6, %6\n")
typedef void (*kern_t)(double *, uint32_t);
static void run(const char *name, kern_t fn, double per_asm, int waves_per_simd) {
    double *d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_wave = (double)ITER * 32 * per_asm;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f cyc per instr per wave,  %.2f cyc/instr/SIMD\n", name, waves_per_simd, ms,
           cyc / per_wave, cyc / (per_wave * waves_per_simd));
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 4}) {
        run("fma64 dep", k_fma_dep, 1, w); run("fma64 2ch", k_fma_2ch, 2, w); run("mul64 dep", k_mul_dep, 1, w);
        run("add64 dep", k_add_dep, 1, w); run("cvt u->d", k_cvt_u2d, 2, w); run("cvt d->u", k_cvt_d2u, 1, w);
        run("cvt d->u->d", k_cvt_rt, 2, w); run("cvt f->d", k_cvt_f2d, 1, w); run("rcp f32", k_rcp_f32, 1, w);
        run("lshl b64", k_lshl64, 1, w); run("fma32 dep", k_fma32_dep, 1, w);
    }
    return 0;
}
