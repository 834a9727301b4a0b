// rcp64.hip -- is v_rcp_f64 + ONE Newton step enough for the arithmetic coder's exact quotients, and what does it cost?
// (a) exhaustive: for every integer v in [1, 2^32]: x0 = v_rcp_f64(v), x1 = x0 + x0 (1 - v x0); the residual |1 - v x| is
//     computed with one FMA (one rounding of an exactly representable difference's neighbourhood) and its maximum reported,
//     together with the same for the seven-instruction sequence the kernels use now (v_rcp_f32 + two steps).
// (b) timing: dependent chains of both sequences, 1 wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32;
typedef unsigned long long u64;

__device__ __forceinline__ double recip_f32_2(double v) {
    double x = (double)__builtin_amdgcn_rcpf((float)v);
    double e = __builtin_fma(-v, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-v, x, 1.0);
    x = __builtin_fma(x, e, x);
    return x;
}
__device__ __forceinline__ double recip_f64_1(double v) {
    double x = __builtin_amdgcn_rcp(v);
    const double e = __builtin_fma(-v, x, 1.0);
    return __builtin_fma(x, e, x);
}
__device__ __forceinline__ u64 dbits(double d) { return __builtin_bit_cast(u64, d); }

// max over the range of |1 - v x| as raw double bits (positive doubles order like their bit patterns)
__global__ void sweep(u64 first, u64 count, u64 *out) {
    u64 m_raw = 0, m_new = 0, m_old = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (u64)gridDim.x * blockDim.x) {
        const double v = (double)(first + i);
        const double x0 = __builtin_amdgcn_rcp(v);
        const double e0 = __builtin_fabs(__builtin_fma(-v, x0, 1.0));
        const double e1 = __builtin_fabs(__builtin_fma(-v, recip_f64_1(v), 1.0));
        const double e2 = __builtin_fabs(__builtin_fma(-v, recip_f32_2(v), 1.0));
        m_raw = max(m_raw, dbits(e0));
        m_new = max(m_new, dbits(e1));
        m_old = max(m_old, dbits(e2));
    }
    atomicMax(&out[0], m_raw);
    atomicMax(&out[1], m_new);
    atomicMax(&out[2], m_old);
}

template <int WHICH>
__global__ void chain(double *out, int iters) {
    double v = 3.0 + threadIdx.x, acc = 0.0;
    for (int i = 0; i < iters; ++i) {
        const double x = WHICH ? recip_f64_1(v) : recip_f32_2(v);
        acc += x;
        v += 1.0 + x;  // dependent
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    u64 *d_out;
    hipMalloc(&d_out, 24);
    hipMemset(d_out, 0, 24);
    sweep<<<4096, 256>>>(1ull, 1ull << 32, d_out);
    u64 h[3];
    hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
    double e[3];
    for (int i = 0; i < 3; ++i) e[i] = *reinterpret_cast<double *>(&h[i]);
    printf("max |1 - v x| over v = 1 .. 2^32:  v_rcp_f64 alone %.3e (2^%.1f)   + one step %.3e (2^%.1f)   v_rcp_f32 + two steps %.3e (2^%.1f)\n",
           e[0], log2(e[0]), e[1], log2(e[1]), e[2], log2(e[2]));
    double *d_acc;
    hipMalloc(&d_acc, 256 * 256 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (which) chain<1><<<256, 256>>>(d_acc, iters); else chain<0><<<256, 256>>>(d_acc, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("%s: %.2f ns per reciprocal (+2 adds) per wave, 1 wave per SIMD, dependent chain\n",
               which ? "v_rcp_f64 + one step    " : "v_rcp_f32 + two steps   ", ms * 1e6 / iters);
    }
    return 0;
}
