// scl_aec_math.h -- exact interval arithmetic shared by the arithmetic-coder fast kernels (scl_aec_fast.hip:
// adaptive models with per-lane tables, scl_aec_static.hip: static model).  Internal to csrc/.
// Reference: ArithmeticEncoder.shrink_range scl/compressors/arithmetic_coding.py:58-78 and the renormalisation
// loops :126-150 / :245-275.  See the header of scl_aec_fast.hip for the exactness arguments.
#pragma once
#include "scl_common.h"

#define AF_HALF 0x80000000u
#define AF_QTR 0x40000000u

// 1 / v for an integer 1 <= v <= 2^32 held exactly in a double, relative error < 2^-50
__device__ __forceinline__ double af_recip(double v) {
    double x = (double)__builtin_amdgcn_rcpf((float)v);
    double e = __builtin_fma(-v, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-v, x, 1.0);
    x = __builtin_fma(x, e, x);
    return x;
}

// shrink_range (:58-78) on (low, hm = high - 1); c, d = c + f, T from the model, x = 1/T
__device__ __forceinline__ void af_shrink(u32 &low, u32 &hm, u32 c, u32 d, u32 T, double x) {
    const double rd = (double)(hm - low) + 1.0;
    const u32 q1 = (u32)(__builtin_fma(rd, (double)c, 0.5) * x);
    const u32 q2 = (u32)(__builtin_fma(rd, (double)d, 0.5) * x);
    hm = (d == T) ? hm : low + q2 - 1;  // (rng*T)//T == rng: high is unchanged (and rng may be 2^32)
    low = low + q1;
}

// closed-form renormalisation counts; returns true if the literal loops must be used for this symbol:
// ctz(low) + k + m + 1 >= 32 for low != 0, likewise for high (v_ffbl of 0 is -1, which wraps to "no")
__device__ __forceinline__ bool af_renorm_counts(u32 low, u32 hm, u32 &k, u32 &m) {
    k = (u32)__builtin_clz(low ^ hm);  // low != hm: the interval holds more than one value
    const u32 z = ((low & ~hm) << k) << 1;
    m = (u32)__builtin_clz(~z);
    const u32 sh = k + m + 1;  // <= 32
    const u32 e_lo = (u32)(__builtin_ffs((int)low) - 1) + sh;
    const u32 e_hi = (u32)(__builtin_ffs((int)(hm + 1)) - 1) + sh;
    return max(e_lo, e_hi) >= 32;
}

