// valu_rate.hip -- issue rate of the integer VALU instructions the coders are built from (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CHAINS 8
#define ITERS 16384

#define KERNEL(name, body)                                                     \
    __global__ __launch_bounds__(256) void k_##name(u32 *out, u32 a0, u32 b0) { \
        u32 v[CHAINS];                                                         \
        for (int c = 0; c < CHAINS; ++c) v[c] = a0 + threadIdx.x * 977u + c;   \
        u32 b = b0 | 1u, tmp = 0, b2 = b0 + 7;                                 \
        asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(b), "v"(b2) : "vcc");  \
        for (int i = 0; i < ITERS; ++i) {                                      \
            _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) {               \
                u32 x = v[c];                                                  \
                unsigned long long xx = x;                                     \
                double dd = __longlong_as_double(0x3ff0000000000000ull | x);   \
                double db = 1.0000001;                                         \
                unsigned long long mask64 = 0x5555555555555555ull ^ b0, mask64w = 0;  \
                body;                                                          \
                v[c] = x;                                                      \
            }                                                                  \
        }                                                                      \
        u32 s = 0;                                                             \
        for (int c = 0; c < CHAINS; ++c) s ^= v[c];                            \
        if (s == 0x12345u) out[blockIdx.x * 256 + threadIdx.x] = s;            \
    }

KERNEL(add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(b)))
KERNEL(mul_hi_u32_u24, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b)))
KERNEL(mul_f32, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(cvt_f32_u32, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x)))
KERNEL(lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(b)))
KERNEL(bfe, asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x)))
KERNEL(sdwa_shl, asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(b)))
KERNEL(mov_dpp, asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(x)))
KERNEL(ffbh, asm volatile("v_ffbh_u32 %0, %0" : "+v"(x)))
KERNEL(and_b32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(or_b32, asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(xor_b32, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(lshlrev, asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x)))
KERNEL(lshrrev, asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(b)))
KERNEL(ashrrev, asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x)))
KERNEL(sub_u32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(subrev_u32, asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x) : "v"(b)))
KERNEL(min_u32, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(mov, asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b)))
KERNEL(cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b)))
KERNEL(add3, asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(b)))
KERNEL(and_or, asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b)))
KERNEL(lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(b)))
KERNEL(perm, asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b)))
KERNEL(add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(b) : "vcc"))
KERNEL(fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b)))
KERNEL(mac_f32, asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x) : "v"(b)))
KERNEL(add_e64, asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(add_sgpr, asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(b0)))
KERNEL(add_const, asm volatile("v_add_u32 %0, 17, %0" : "+v"(x)))
KERNEL(mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(xx) : "v"(b), "v"(b2) : "vcc"); x = (u32)xx)
KERNEL(lshl_b64, asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(xx) : "v"(b)); x = (u32)xx)
KERNEL(mul_u32_u24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd) : "v"(db)); x = (u32)__double_as_longlong(dd))
KERNEL(mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd) : "v"(db)); x = (u32)__double_as_longlong(dd))
KERNEL(add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd) : "v"(db)); x = (u32)__double_as_longlong(dd))
KERNEL(cvt_f64_u32, asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(dd) : "v"(x)); x = (u32)__double_as_longlong(dd))
KERNEL(cvt_u32_f64, asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(x) : "v"(dd)))
KERNEL(rcp_f32, asm volatile("v_rcp_f32 %0, %0" : "+v"(x)))
KERNEL(cvt_u32_f32, asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x)))
KERNEL(cnd_vcc_dst, asm volatile("v_cndmask_b32 %1, %0, %2, vcc\n\tv_add_u32 %0, %0, %1" : "+v"(x), "+v"(tmp) : "v"(b) : "vcc"))
KERNEL(cnd_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(mask64)))
KERNEL(cnd_sgpr_dst, asm volatile("v_cndmask_b32_e64 %1, %0, %2, %3\n\tv_add_u32 %0, %0, %1" : "+v"(x), "+v"(tmp) : "v"(b), "s"(mask64)))
KERNEL(cmp_only, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
KERNEL(cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc"))
KERNEL(cmp_cnd_sgpr, asm volatile("v_cmp_lt_u32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x), "+v"(b) , "+s"(mask64w)))
KERNEL(add_waitcnt, asm volatile("v_add_u32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(b)))
KERNEL(add_snop, asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(x) : "v"(b)))
KERNEL(alignbit_waitcnt, asm volatile("v_alignbit_b32 %0, %0, %1, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(b)))
KERNEL(alignbit_branch, asm volatile("v_alignbit_b32 %0, %0, %1, %1\n\ts_cbranch_execz 0" : "+v"(x) : "v"(b)))
KERNEL(sub_co, asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(b) : "vcc"))

static int g_waves = 8;
template <typename K>
static void run(const char *name, K kern, u32 *d) {
    const int blocks = 256 * g_waves;  // g_waves waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, 256>>>(d, 1, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, 256>>>(d, 1, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double waveinstr = (double)blocks * 4 * ITERS * CHAINS;  // per-wave instructions
    const double per_simd = waveinstr / 1024.0;                    // 256 CUs x 4 SIMDs
    printf("%-16s %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / per_simd);
}

KERNEL(cmp_eq, asm volatile("v_cmp_eq_u32 vcc, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(x) : "v"(b) : "vcc"))
KERNEL(cmp_cnd_add, asm volatile("v_cmp_eq_u32 vcc, %0, %1\n\tv_cndmask_b32 %2, %1, %3, vcc\n\tv_add_u32 %0, %0, %2" : "+v"(x) : "v"(b), "v"(tmp), "v"(b2) : "vcc"))
KERNEL(cnd_only, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc"))
KERNEL(ring256, asm volatile("v_add_u32 %1, 0xfc, %0\n\tv_and_or_b32 %0, %1, %2, %3" : "+v"(x) : "v"(tmp), "s"(b0), "v"(b)))
KERNEL(ring_c, asm volatile("v_sub_u32 %1, %0, %2\n\tv_ashrrev_i32 %1, 31, %1\n\tv_and_b32 %1, 0xc0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, -4, %0" : "+v"(x) : "v"(tmp), "v"(b)))
KERNEL(ring_d, asm volatile("v_add_u32 %1, -4, %0\n\tv_add_u32 %0, 0xbc, %0\n\tv_min_u32 %0, %0, %1\n\tv_add_u32 %1, %0, %2" : "+v"(x) : "v"(tmp), "v"(b)))
KERNEL(mix_fs, asm volatile("v_add_u32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b)))
int main(int argc, char **argv) {
    if (argc > 1) {
        u32 *dd;
        hipMalloc(&dd, 256 * 8 * 256 * 4);
        for (int w = 1; w <= 8; w *= 2) {
            g_waves = w;
            printf("-- %d waves per SIMD\n", w);
            run("add", k_add, dd);
            run("alignbit", k_alignbit, dd);
            run("mul_hi_u32", k_mul_hi_u32, dd);
            run("mix_fs(2 instr)", k_mix_fs, dd);
            run("add + s_waitcnt", k_add_waitcnt, dd);
            run("alignbit + s_waitcnt", k_alignbit_waitcnt, dd);
            run("alignbit + s_cbranch", k_alignbit_branch, dd);
            run("cmp_eq+add (2)", k_cmp_eq, dd);
            run("cmp+cndmask+add (3)", k_cmp_cnd_add, dd);
            run("cndmask", k_cnd_only, dd);
            run("ring256 add+and_or (2)", k_ring256, dd);
            run("ring_c (5 fast)", k_ring_c, dd);
            run("ring_d (4)", k_ring_d, dd);
        }
        return 0;
    }
    u32 *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
#define RUN(n) run(#n, k_##n, d);
    RUN(add) RUN(mul_hi_u32) RUN(mul_lo_u32) RUN(mad_u32_u24) RUN(mul_hi_u32_u24) RUN(alignbit) RUN(mul_f32)
    RUN(cvt_f32_u32) RUN(lshl_add) RUN(bfe) RUN(sdwa_shl) RUN(mov_dpp) RUN(ffbh) RUN(sub_co)
    RUN(and_b32) RUN(or_b32) RUN(xor_b32) RUN(lshlrev) RUN(lshrrev) RUN(ashrrev) RUN(sub_u32) RUN(subrev_u32) RUN(min_u32)
    RUN(mov) RUN(cndmask) RUN(add3) RUN(and_or) RUN(lshl_or) RUN(perm) RUN(add_co) RUN(fma_f32) RUN(mac_f32) RUN(add_e64)
    RUN(add_sgpr) RUN(add_const) RUN(add) RUN(mad_u64_u32) RUN(lshl_b64) RUN(mul_u32_u24) RUN(fma_f64) RUN(mul_f64) RUN(add_f64) RUN(cvt_f64_u32) RUN(cvt_u32_f64) RUN(rcp_f32) RUN(cvt_u32_f32) RUN(cndmask) RUN(cnd_vcc_dst) RUN(cnd_sgpr) RUN(cnd_sgpr_dst) RUN(cmp_only) RUN(cmp_cnd) RUN(cmp_cnd_sgpr) RUN(add) RUN(add_waitcnt) RUN(add_snop) RUN(alignbit) RUN(alignbit_waitcnt) RUN(alignbit_branch)
    return 0;
}
