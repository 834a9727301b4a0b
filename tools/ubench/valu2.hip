// micro-benchmark 2: per-instruction issue cost on gfx950 (wave64), exact instructions via inline asm.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 2048
#define REP8(x) x x x x x x x x
#define DEF(NAME, ASM)                                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {                \
        uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = b + 7;               \
        uint32_t e = a + 11, f = b ^ c, g = c + d, h = 5;                                      \
        for (int i = 0; i < ITER; ++i) {                                                       \
            asm volatile(REP8(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "vcc"); \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;            \
    }
// each ASM block = 4 independent instructions on a,c,e,g (sources b,d,f,h)
DEF(k_add,    "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n")
DEF(k_mulhi,  "v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %4, %4, %5\n v_mul_hi_u32 %6, %6, %7\n")
DEF(k_mullo,  "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %6, %6, %7\n")
DEF(k_mul24,  "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %6, %6, %7\n")
DEF(k_mad24,  "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %6, %6, %7, %0\n")
DEF(k_add3,   "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %6, %6, %7, %0\n")
DEF(k_bfe,    "v_bfe_u32 %0, %0, 1, %7\n v_bfe_u32 %2, %2, 1, %7\n v_bfe_u32 %4, %4, 1, %7\n v_bfe_u32 %6, %6, 1, %7\n")
DEF(k_lshlor, "v_lshl_or_b32 %0, %0, %7, %1\n v_lshl_or_b32 %2, %2, %7, %3\n v_lshl_or_b32 %4, %4, %7, %5\n v_lshl_or_b32 %6, %6, %7, %1\n")
DEF(k_perm,   "v_perm_b32 %0, %0, %1, %7\n v_perm_b32 %2, %2, %3, %7\n v_perm_b32 %4, %4, %5, %7\n v_perm_b32 %6, %6, %7, %7\n")
DEF(k_align,  "v_alignbit_b32 %0, %0, %1, %7\n v_alignbit_b32 %2, %2, %3, %7\n v_alignbit_b32 %4, %4, %5, %7\n v_alignbit_b32 %6, %6, %7, %7\n")
DEF(k_lshr,   "v_lshrrev_b32 %0, %7, %0\n v_lshrrev_b32 %2, %7, %2\n v_lshrrev_b32 %4, %7, %4\n v_lshrrev_b32 %6, %7, %6\n")
DEF(k_sdwa,   "v_lshlrev_b32_sdwa %0, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %2, %7, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %4, %7, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %6, %7, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n")
DEF(k_cmpaddc,"v_cmp_ge_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_ge_u32 vcc, %2, %3\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n")
DEF(k_cmpe64, "v_cmp_ge_u32_e64 s[20:21], %0, %1\n v_addc_co_u32_e64 %0, s[22:23], 0, %0, s[20:21]\n v_cmp_ge_u32_e64 s[24:25], %2, %3\n v_addc_co_u32_e64 %2, s[26:27], 0, %2, s[24:25]\n")
DEF(k_ffbh,   "v_ffbh_u32 %0, %0\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %4, %4\n v_ffbh_u32 %6, %6\n")

DEF(k_and,    "v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n v_and_b32 %6, %6, %7\n")
DEF(k_or,     "v_or_b32 %0, %0, %1\n v_or_b32 %2, %2, %3\n v_or_b32 %4, %4, %5\n v_or_b32 %6, %6, %7\n")
DEF(k_xor,    "v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7\n")
DEF(k_sub,    "v_sub_u32 %0, %0, %1\n v_sub_u32 %2, %2, %3\n v_sub_u32 %4, %4, %5\n v_sub_u32 %6, %6, %7\n")
DEF(k_lshl,   "v_lshlrev_b32 %0, %7, %0\n v_lshlrev_b32 %2, %7, %2\n v_lshlrev_b32 %4, %7, %4\n v_lshlrev_b32 %6, %7, %6\n")
DEF(k_ashr,   "v_ashrrev_i32 %0, %7, %0\n v_ashrrev_i32 %2, %7, %2\n v_ashrrev_i32 %4, %7, %4\n v_ashrrev_i32 %6, %7, %6\n")
DEF(k_mov,    "v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n")
DEF(k_cndmask,"v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n")
DEF(k_max,    "v_max_u32 %0, %0, %1\n v_max_u32 %2, %2, %3\n v_max_u32 %4, %4, %5\n v_max_u32 %6, %6, %7\n")
DEF(k_cmp,    "v_cmp_ge_u32 vcc, %0, %1\n v_cmp_ge_u32 vcc, %2, %3\n v_cmp_ge_u32 vcc, %4, %5\n v_cmp_ge_u32 vcc, %6, %7\n")
DEF(k_addc,   "v_addc_co_u32 %0, vcc, %0, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %3, vcc\n v_addc_co_u32 %4, vcc, %4, %5, vcc\n v_addc_co_u32 %6, vcc, %6, %7, vcc\n")
DEF(k_addco,  "v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %4, vcc, %4, %5\n v_add_co_u32 %6, vcc, %6, %7\n")
DEF(k_lshladd,"v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %2, %2, 2, %3\n v_lshl_add_u32 %4, %4, 2, %5\n v_lshl_add_u32 %6, %6, 2, %7\n")
DEF(k_andor,  "v_and_or_b32 %0, %0, %1, %7\n v_and_or_b32 %2, %2, %3, %7\n v_and_or_b32 %4, %4, %5, %7\n v_and_or_b32 %6, %6, %7, %1\n")
DEF(k_mulhi24,"v_mul_hi_u32_u24 %0, %0, %1\n v_mul_hi_u32_u24 %2, %2, %3\n v_mul_hi_u32_u24 %4, %4, %5\n v_mul_hi_u32_u24 %6, %6, %7\n")
DEF(k_bfeimm, "v_bfe_u32 %0, %0, 8, 12\n v_bfe_u32 %2, %2, 8, 12\n v_bfe_u32 %4, %4, 8, 12\n v_bfe_u32 %6, %6, 8, 12\n")
DEF(k_andimm, "v_and_b32 %0, 0xfff, %0\n v_and_b32 %2, 0xfff, %2\n v_and_b32 %4, 0xfff, %4\n v_and_b32 %6, 0xfff, %6\n")
DEF(k_addimm, "v_add_u32 %0, 0x12345, %0\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %6, 0x12345, %6\n")
DEF(k_lshrimm,"v_lshrrev_b32 %0, 12, %0\n v_lshrrev_b32 %2, 12, %2\n v_lshrrev_b32 %4, 12, %4\n v_lshrrev_b32 %6, 12, %6\n")
DEF(k_pkadd16,"v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %4, %4, %5\n v_pk_add_u16 %6, %6, %7\n")
DEF(k_addsgpr,"v_add_u32 %0, s20, %0\n v_add_u32 %2, s21, %2\n v_add_u32 %4, s22, %4\n v_add_u32 %6, s23, %6\n")
DEF(k_cvtf,   "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %6, %6\n")
DEF(k_fma,    "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %6, %6, %7, %0\n")
DEF(k_fmul,   "v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %3\n v_mul_f32 %4, %4, %5\n v_mul_f32 %6, %6, %7\n")
DEF(k_mbcnt,  "v_mbcnt_lo_u32_b32 %0, %0, %1\n v_mbcnt_lo_u32_b32 %2, %2, %3\n v_mbcnt_lo_u32_b32 %4, %4, %5\n v_mbcnt_lo_u32_b32 %6, %6, %7\n")

__global__ void __launch_bounds__(256) k_lshl64(uint32_t *out, uint32_t seed) {
    uint64_t a = threadIdx.x + seed, c = a * 3 + 1, e = a ^ 0x55, g = c + 7;
    uint32_t h = 5;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n")
                     : "+v"(a), "+v"(c), "+v"(e), "+v"(g) : "v"(h));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + c + e + g);
}
__global__ void __launch_bounds__(256) k_ldsread(uint32_t *out, uint32_t seed) {
    __shared__ uint4 tab[256];
    tab[threadIdx.x] = make_uint4(threadIdx.x * 2654435761u, seed, 3, 4);
    __syncthreads();
    uint32_t s = (threadIdx.x * 7 + seed) & 255, acc = 0;
    for (int i = 0; i < ITER * 4; ++i) {
        uint4 e = tab[s];
        acc += e.y;
        s = (e.x >> 7) & 255;   // data-dependent pseudo-random index
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s;
}
typedef void (*kern_t)(uint32_t *, uint32_t);
static void run(const char *name, kern_t fn, double instr_per_iter, int waves_per_simd, int iters) {
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)blocks * 4 * iters * instr_per_iter / 1024.0;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f cyc/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / per_simd);
    (void)hipFree(d);
}
int main() {
    for (int w : {4}) {
#define R(k) run(#k, k, 32, w, ITER)
        R(k_add); R(k_mulhi); R(k_mullo); R(k_mul24); R(k_mad24); R(k_add3); R(k_bfe); R(k_lshlor); R(k_perm); R(k_align);
        R(k_lshr); R(k_and); R(k_or); R(k_xor); R(k_sub); R(k_lshl); R(k_ashr); R(k_mov); R(k_cndmask); R(k_max); R(k_cmp); R(k_addc); R(k_addco); R(k_lshladd); R(k_andor); R(k_mulhi24); R(k_bfeimm); R(k_andimm); R(k_addimm); R(k_lshrimm); R(k_pkadd16); R(k_addsgpr); R(k_cvtf); R(k_fma); R(k_fmul); R(k_mbcnt); R(k_sdwa); R(k_cmpaddc); R(k_cmpe64); R(k_ffbh); R(k_lshl64);
        run("k_ldsread(b128)", k_ldsread, 1, w, ITER * 4);
    }
    return 0;
}
