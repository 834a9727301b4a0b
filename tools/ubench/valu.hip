// micro-benchmark: integer VALU issue rate on gfx950 (wave64), to price the rANS inner loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = b + 7, e = a + 11, f = b ^ c, g = c + d, h = d ^ 99;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) { a += b; c += d; e += f; g += h; b += a; d += c; f += e; h += g; }            // v_add_u32 x8
            if (OP == 1) { a = __umulhi(a, b) + 1; c = __umulhi(c, d) + 1; e = __umulhi(e, f) + 1; g = __umulhi(g, h) + 1; } // mulhi+add x4
            if (OP == 2) { a = (a << (b & 7)) | 1; c = (c >> (d & 7)) + 1; e = (e << (f & 7)) | 1; g = (g >> (h & 7)) + 1; } // and,shift,or x4 = 12
            if (OP == 3) { a = __umul24(a, b) + c; e = __umul24(e, f) + g; b = __umul24(b, d) + h; f = __umul24(f, h) + d; } // mad24 x4
            if (OP == 4) { a = __builtin_amdgcn_ubfe(a, 3, 9) + b; c = __builtin_amdgcn_ubfe(c, 3, 9) + d; e = __builtin_amdgcn_ubfe(e, 2, 11) + f; g = __builtin_amdgcn_ubfe(g, 1, 13) + h; } // bfe+add x4 = 8
            if (OP == 5) { uint64_t t = ((uint64_t)a << (b & 31)); a = (uint32_t)t + 1; c = (uint32_t)(t >> 32) + c; uint64_t u = ((uint64_t)e << (f & 31)); e = (uint32_t)u + 1; g += (uint32_t)(u >> 32); } // lshl_b64
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int OP>
void run(const char *name, int instr_per_inner, int waves_per_simd) {
    uint32_t *d; hipMalloc(&d, 256 * 4 * 1024 * 4 * 8);
    int blocks = 256 * waves_per_simd;  // 256 CUs x (4 SIMD x w waves / 4 waves per block)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * ITER * 8 * instr_per_inner;
    double per_simd = wave_instr / 1024.0;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32 (8 indep)", 8, w);
        run<1>("v_mul_hi_u32+add (4 chains)", 8, w);
        run<2>("and+shift+or (12)", 12, w);
        run<3>("v_mad_u32_u24 (4)", 4, w);
        run<4>("v_bfe_u32+add (8)", 8, w);
        run<5>("lshl_b64 mix (~8)", 8, w);
    }
    return 0;
}
