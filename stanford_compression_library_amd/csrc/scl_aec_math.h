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

// the same for a small integer given as float AND as double (v < 2^24, both exact): the caller gets the float with one
// SDWA conversion out of a packed field and the double from the float
__device__ __forceinline__ double af_recip_fd(float vf, double v) {
    double x = (double)__builtin_amdgcn_rcpf(vf);
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


// ---- round-3 forms (scl_aec_split.hip first, then every tuned arithmetic-coder kernel) ------------------------------
// shrink_range without the d == T special case: high' - 1 = low + ((rng d) // T - 1) with the "- 1" inside the FMA, so
// the value converted is at most 2^32 - 1 even when the quotient is 2^32 (rng = 2^32, d = T).  (rng d) // T >= 1 always
// (rng > 2^30 after a renormalisation, d >= 1, T <= 2^16), so trunc(v - 1) = trunc(v) - 1; precision as in af_shrink.
__device__ __forceinline__ void af_shrink2(u32 &low, u32 &hm, u32 c, u32 d, double x) {
    const double rd = (double)(hm - low) + 1.0;
    const u32 q1 = (u32)(__builtin_fma(rd, (double)c, 0.5) * x);
    const u32 q2m1 = (u32)__builtin_fma(__builtin_fma(rd, (double)d, 0.5), x, -1.0);
    hm = low + q2m1;
    low = low + q1;
}

// the same with c and d already converted
__device__ __forceinline__ void af_shrink2_d(u32 &low, u32 &hm, double c, double d, double x) {
    const double rd = (double)(hm - low) + 1.0;
    const u32 q1 = (u32)(__builtin_fma(rd, c, 0.5) * x);
    const u32 q2m1 = (u32)__builtin_fma(__builtin_fma(rd, d, 0.5), x, -1.0);
    hm = low + q2m1;
    low = low + q1;
}

// shrink_range for a power-of-two total T = 2^t (static models; tANS-style tables): rng c >> t in integers.  rng c =
// (rng - 1) c + c is one v_mad_u64_u32 (rng itself may be 2^32), the quotient one v_alignbit; (rng d) >> t can be 2^32
// (rng = 2^32, d = T) but high' - 1 = low + that - 1 is below 2^32, so arithmetic mod 2^32 is exact.  Seven integer
// instructions instead of thirteen with six binary64 ones.
__device__ __forceinline__ void af_shrink_pow2(u32 &low, u32 &hm, u32 c, u32 d, u32 t) {
    const u32 r1 = hm - low;
    const u64 p1 = (u64)r1 * c + c;
    const u64 p2 = (u64)r1 * d + d;
    const u32 q1 = __builtin_amdgcn_alignbit((u32)(p1 >> 32), (u32)p1, t);
    const u32 q2 = __builtin_amdgcn_alignbit((u32)(p2 >> 32), (u32)p2, t);
    hm = low + q2 - 1u;
    low = low + q1;
}

// the same with c and d already zero-extended to 64 bits in register pairs (their low words are the multiplicands)
__device__ __forceinline__ void af_shrink_pow2_wide(u32 &low, u32 &hm, u64 c, u64 d, u32 t) {
    const u32 r1 = hm - low;
    const u64 p1 = (u64)r1 * (u32)c + c;
    const u64 p2 = (u64)r1 * (u32)d + d;
    const u32 q1 = __builtin_amdgcn_alignbit((u32)(p1 >> 32), (u32)p1, t);
    const u32 q2 = __builtin_amdgcn_alignbit((u32)(p2 >> 32), (u32)p2, t);
    hm = low + q2 - 1u;
    low = low + q1;
}

// closed-form step counts AND the renormalised interval; true if the literal loops must be used for this symbol.
// The corner test of af_renorm_counts on the SHIFTED values: ctz(low) + k + m + 1 >= 32 with low != 0 <=> every bit of
// low leaves, i.e. (low << (k + m)) & 0x7FFFFFFF == 0; for high = hm + 1 != 2^32: (high << (k + m + 1)) mod 2^32 == 0 <=>
// ((hm << kt) | ones(kt)) + 1 is 0 or 2^31 <=> the new hm (with its top bit set) is all ones.
__device__ __forceinline__ bool af_renorm2(u32 low, u32 hm, u32 &k, u32 &m, u32 &nlow, u32 &nhm) {
    k = (u32)__builtin_clz(low ^ hm);  // low != hm: the interval holds more than one value
    const u32 z = ((low & ~hm) << k) << 1;
    m = (u32)__builtin_clz(~z);
    const u32 kt = k + m;  // <= 31
    nlow = (low << kt) & 0x7FFFFFFFu;
    nhm = (hm << kt) | ((1u << kt) - 1u) | AF_HALF;
    return (nlow == 0 && low != 0) || (nhm == 0xFFFFFFFFu && hm != 0xFFFFFFFFu);
}

// The decoders' variant: the corner test without the "low != 0" / "hm != 2^32 - 1" exclusions -- two compares fewer per
// symbol.  It sends a few more symbols through the literal loops (exact for every interval): those coded while low is still
// 0 or high still 2^32, i.e. the first symbols of a chunk.
__device__ __forceinline__ bool af_renorm2_dec(u32 low, u32 hm, u32 &k, u32 &m, u32 &nlow, u32 &nhm) {
    k = (u32)__builtin_clz(low ^ hm);
    const u32 z = ((low & ~hm) << k) << 1;
    m = (u32)__builtin_clz(~z);
    const u32 kt = k + m;  // <= 31
    nlow = (low << kt) & 0x7FFFFFFFu;
    nhm = ~(~hm << kt) | AF_HALF;  // (hm << kt) | ones(kt) | HALF
    return nlow == 0 || nhm == 0xFFFFFFFFu;
}
