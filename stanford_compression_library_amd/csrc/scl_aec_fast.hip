// scl_aec_fast.hip -- adaptive arithmetic coding with per-lane context tables in LDS (BASELINE.json configs[3]),
// one wavefront lane per chunk.  Same streams, bit for bit, as scl_aec.hip and the reference:
//   ArithmeticEncoder.shrink_range / encode_block   scl/compressors/arithmetic_coding.py:58-78, :80-161
//   ArithmeticDecoder.decode_step_core / decode_block                               :177-201, :203-287
//   AdaptiveIIDFreqModel / AdaptiveOrderKFreqModel   scl/compressors/probability_models.py:70-92, :95-160
//
// Served models (aec_fast_ok): PRECISION = 32, DATA_BLOCK_SIZE_BITS = 32, adaptive IID or order-k model with
// alphabet 2..16 and at most 16 contexts (K^k <= 16: order-1 K <= 16, order-2 K <= 4, ...), totals that stay
// below 2^15 and below the model's rescale threshold for the whole chunk.  Everything else -> scl_aec.hip.
//
// What is different from the generic kernel (13.7 / 21.7 ms per 256 MiB, profiles/r01_bench_aec_k16.json):
//  * Model layout.  A context row is stored as 16 u16 INCLUSIVE cumulative counts E[j] = count[0] + .. + count[j]
//    (E[15] = total; unused symbols are padded with the total), 32 bytes per row, row `ctx` of thread t at
//    LDS [ctx][half][t] as two uint4: c = E[s-1], d = E[s], T = E[15] are single 2-byte reads, the update
//    `count[s] += 1` is E[j] += 1 for j >= s = eight v_pk_add_u16 on the row with a mask row from a 512-byte LUT,
//    and the decoder's search max{s : c[s] <= target} is a packed compare-and-count over the 8 row registers.
//    16 contexts x 32 B x 256 lanes = 128 KiB: one workgroup of 256 lanes per CU (one wave per SIMD).
//  * Arithmetic.  (rng*c)//T and ((state-low+1)*T-1)//rng are evaluated in binary64: every operand is an integer
//    below 2^47, q = trunc((num + 0.5) * x) with x = 1/den refined by two Newton steps from v_rcp_f32 is exact
//    because the quotient's error (< 2^-18 resp. 2^-35) is below the distance 0.5/den of (num + 0.5)/den from the
//    nearest integer (den < 2^15 resp. <= 2^32).  low and high-1 are kept as u32.
//  * Renormalisation in closed form.  With hm = high - 1: k = clz(low ^ hm) E1/E2 steps emit the k common leading
//    bits, then m = number of leading (1,0) bit pairs of (low, hm) below them E3 steps.  The reference's STRICT
//    comparisons (quirk Q1: `high < HALF`, `low > HALF`, `low > QTR and high < 3*QTR`) differ from this only when
//    low or high hits a power-of-two boundary inside the shifted-out prefix, i.e. when
//    (low << (k+m+1)) == 0 or (high << (k+m+1)) == 0; those symbols (probability ~2^-20) and runs of more than
//    32 - k pending bits take the literal loops of the reference.
//  * Encoder software pipeline: the model reads/updates of symbol i+1 are issued before the arithmetic of
//    symbol i, so LDS latency is off the critical path; output words collect in a 16-register FIFO and leave as
//    64-byte bursts; input arrives as 16-byte loads, one block ahead.
#include "scl_aec_internal.h"

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// LDS rows are read as 2-byte elements and written as 16-byte halves: the accesses must not be reordered by
// type-based alias analysis
typedef u16 __attribute__((may_alias)) u16_lds;
typedef u32 __attribute__((may_alias)) u32_lds;
typedef uint4 __attribute__((may_alias)) uint4_lds;

#define AF_THREADS 256
#define AF_HALF_BYTES (AF_THREADS * 16)  // one 16-byte half row of every thread
#define AF_CTX_BYTES (2 * AF_HALF_BYTES)
#define AF_TABLE_BYTES (16 * AF_CTX_BYTES)
#define AF_LUT_BYTES 512
#define AF_LDS_BYTES (AF_TABLE_BYTES + AF_LUT_BYTES)
#define AF_HALF 0x80000000u
#define AF_QTR 0x40000000u

struct AecFastDev {
    u32 K;          // alphabet size 2..16
    u32 nctx;       // K^k <= 16
    u32 ctx_magic;  // ceil(2^16 / nctx): (v * magic) >> 16 == v / nctx for v < 272
    u32 initE[8];   // 16 packed u16: inclusive cumulative initial counts, padded with the total
};

__device__ __forceinline__ u32 af_pk_add(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
// acc += (e > t) ? -1 : 0 per 16-bit half (all values < 2^15)
__device__ __forceinline__ u32 af_pk_count_gt(u32 acc, u32 t, u32 e) {
    s16x2 d = __builtin_bit_cast(s16x2, t) - __builtin_bit_cast(s16x2, e);
    d = d >> (s16x2)(15);
    return __builtin_bit_cast(u32, (s16x2)(__builtin_bit_cast(s16x2, acc) + d));
}

// 1 / v for an integer 1 <= v <= 2^32 held exactly in a double, relative error < 2^-50
__device__ __forceinline__ double af_recip(double v) {
    double x = (double)__builtin_amdgcn_rcpf((float)v);
    double e = __builtin_fma(-v, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-v, x, 1.0);
    x = __builtin_fma(x, e, x);
    return x;
}

struct AfRow {
    uint4 a, b;
};

__device__ __forceinline__ void af_setup_tables(char *lds, const AecFastDev &P, u32 tid) {
    // mask rows: LUT[s] = packed (j >= s) for j = 0..15
    if (tid < 128) {
        const u32 s = tid >> 3, r = tid & 7;  // register r holds elements 2r, 2r+1
        const u32 v = ((2 * r >= s) ? 1u : 0u) | ((2 * r + 1 >= s) ? 0x10000u : 0u);
        *reinterpret_cast<u32_lds *>(lds + AF_TABLE_BYTES + s * 32 + r * 4) = v;
    }
    const uint4 a = make_uint4(P.initE[0], P.initE[1], P.initE[2], P.initE[3]);
    const uint4 b = make_uint4(P.initE[4], P.initE[5], P.initE[6], P.initE[7]);
    for (u32 c = 0; c < P.nctx; ++c) {
        *reinterpret_cast<uint4_lds *>(lds + c * AF_CTX_BYTES + tid * 16) = a;
        *reinterpret_cast<uint4_lds *>(lds + c * AF_CTX_BYTES + AF_HALF_BYTES + tid * 16) = b;
    }
    __syncthreads();
}

__device__ __forceinline__ u32 af_elem_addr(u32 rowbase, u32 j) {  // byte address of E[j] of this thread's row
    return rowbase + (j >> 3) * AF_HALF_BYTES + (j & 7) * 2;
}
__device__ __forceinline__ u32 af_next_ctx(const AecFastDev &P, u32 ctx, u32 s) {  // past_k[1:] + [s], :146-151
    const u32 v = ctx * P.K + s;
    return v - ((v * P.ctx_magic) >> 16) * P.nctx;
}
__device__ __forceinline__ AfRow af_row_plus_mask(const AfRow &R, const char *lds, u32 s) {
    const uint4 ia = *reinterpret_cast<const uint4_lds *>(lds + AF_TABLE_BYTES + s * 32);
    const uint4 ib = *reinterpret_cast<const uint4_lds *>(lds + AF_TABLE_BYTES + s * 32 + 16);
    AfRow o;
    o.a = make_uint4(af_pk_add(R.a.x, ia.x), af_pk_add(R.a.y, ia.y), af_pk_add(R.a.z, ia.z), af_pk_add(R.a.w, ia.w));
    o.b = make_uint4(af_pk_add(R.b.x, ib.x), af_pk_add(R.b.y, ib.y), af_pk_add(R.b.z, ib.z), af_pk_add(R.b.w, ib.w));
    return o;
}

// shrink_range (:58-78) on (low, hm = high - 1); c, d = c + f, T from the model
__device__ __forceinline__ void af_shrink(u32 &low, u32 &hm, u32 c, u32 d, u32 T) {
    const double rd = (double)(hm - low) + 1.0;
    const double x = af_recip((double)T);
    const u32 q1 = (u32)(__builtin_fma(rd, (double)c, 0.5) * x);
    const u32 q2 = (u32)(__builtin_fma(rd, (double)d, 0.5) * x);
    hm = (d == T) ? hm : low + q2 - 1;  // (rng*T)//T == rng: high is unchanged (and rng may be 2^32)
    low = low + q1;
}

// closed-form renormalisation counts; returns true if the literal loops must be used for this symbol
__device__ __forceinline__ bool af_renorm_counts(u32 low, u32 hm, u32 &k, u32 &m) {
    k = (u32)__builtin_clz(low ^ hm);  // low != hm: the interval holds more than one value
    const u32 z = ((low & ~hm) << k) << 1;
    m = (u32)__builtin_clz(~z);
    const u32 sh = k + m + 1;  // <= 32
    const u32 hi = hm + 1;      // low 32 bits of high
    const bool lo_edge = (low != 0) & (((low << (sh & 31)) == 0) | (sh >= 32));
    const bool hi_edge = (hi != 0) & (((hi << (sh & 31)) == 0) | (sh >= 32));
    return lo_edge | hi_edge;
}

// ---- forward bit writer: completed big-endian words collect in a 16-register FIFO, 64-byte bursts ------------
struct AfWriter {
    u64 acc;
    u32 nacc;  // < 32 pending bits in acc
    u32 cnt;   // words in the FIFO
    u32 w[16];
    uint4 *dst;
    u64 nwords;  // words already stored

    __device__ __forceinline__ void init(u8 *slot) {
        acc = 0;
        nacc = 0;
        cnt = 0;
        nwords = 0;
        dst = reinterpret_cast<uint4 *>(slot);
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = 0;
    }
    __device__ __forceinline__ void push(u32 word) {
#pragma unroll
        for (int i = 15; i > 0; --i) w[i] = w[i - 1];
        w[0] = __builtin_bswap32(word);
        if (++cnt == 16) {
            dst[0] = make_uint4(w[15], w[14], w[13], w[12]);
            dst[1] = make_uint4(w[11], w[10], w[9], w[8]);
            dst[2] = make_uint4(w[7], w[6], w[5], w[4]);
            dst[3] = make_uint4(w[3], w[2], w[1], w[0]);
            dst += 4;
            nwords += 16;
            cnt = 0;
        }
    }
    __device__ __forceinline__ void put(u32 v, u32 nb) {  // v < 2^nb, nb <= 32
        acc = (acc << nb) | v;
        nacc += nb;
        if (nacc >= 32) {
            nacc -= 32;
            push((u32)(acc >> nacc));
        }
    }
    __device__ __forceinline__ void put_run(u32 bit, u64 count) {
        while (count >= 32) {
            put(bit ? 0xFFFFFFFFu : 0u, 32);
            count -= 32;
        }
        if (count) put(bit ? ((1u << count) - 1u) : 0u, (u32)count);
    }
    __device__ __forceinline__ u64 finish() {  // zero-pads to the next 64-byte boundary
        const u64 total = (nwords + cnt) * 32 + nacc;
        if (nacc) put(0, 32 - nacc);
        while (cnt) push(0);
        return total;
    }
};

__global__ void __launch_bounds__(AF_THREADS)
    aec_fast_encode_kernel(AecFastDev P, const u8 *__restrict__ sym, u64 sym_stride, const u32 *__restrict__ lens,
                           u32 chunk_len, u64 n_chunks, u8 *__restrict__ out, u64 out_stride,
                           u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits, u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[AF_LDS_BYTES];
    const u32 tid = threadIdx.x;
    af_setup_tables(lds, P, tid);
    const u64 chunk = (u64)blockIdx.x * AF_THREADS + tid;
    if (chunk >= n_chunks) return;
    const u32 n = lens ? lens[chunk] : chunk_len;
    const uint4 *src = reinterpret_cast<const uint4 *>(sym + chunk * sym_stride);
    AfWriter wr;
    wr.init(out + chunk * out_stride);
    wr.put(n, 32);
    u32 st = 0;
    u32 low = 0, hm = 0xFFFFFFFFu;
    u64 pending = 0;
    u32 ctx = 0;
    uint4 cur = make_uint4(0, 0, 0, 0), pf = make_uint4(0, 0, 0, 0);
    if (n > 0) pf = src[0];
    u32 word = 0;
    u32 c_nx = 0, d_nx = 0, T_nx = 1;
    for (u32 p = 0; p <= n; ++p) {
        const u32 cc = c_nx, dd = d_nx, TT = T_nx;
        if (p < n) {
            // ---- model access for symbol p (its arithmetic happens in the next iteration) ----
            if ((p & 15) == 0) {
                cur = pf;
                if (p + 16 < n) pf = src[(p >> 4) + 1];
            }
            if ((p & 3) == 0) {
                word = cur.x;
                cur.x = cur.y;
                cur.y = cur.z;
                cur.z = cur.w;
            }
            u32 s = word & 0xFFu;
            word >>= 8;
            if (s >= P.K) {
                st |= SCL_ST_SYMBOL;
                s = 0;
            }
            const u32 rowbase = ctx * AF_CTX_BYTES + tid * 16;
            const u32 sm1 = (s == 0) ? 0 : s - 1;
            const u32 c_raw = *reinterpret_cast<const u16_lds *>(lds + af_elem_addr(rowbase, sm1));
            d_nx = *reinterpret_cast<const u16_lds *>(lds + af_elem_addr(rowbase, s));
            c_nx = (s == 0) ? 0 : c_raw;
            AfRow R;
            R.a = *reinterpret_cast<const uint4_lds *>(lds + rowbase);
            R.b = *reinterpret_cast<const uint4_lds *>(lds + rowbase + AF_HALF_BYTES);
            T_nx = R.b.w >> 16;
            const AfRow R2 = af_row_plus_mask(R, lds, s);  // update_model, :118 (after this symbol's lookup)
            *reinterpret_cast<uint4_lds *>(lds + rowbase) = R2.a;
            *reinterpret_cast<uint4_lds *>(lds + rowbase + AF_HALF_BYTES) = R2.b;
            ctx = af_next_ctx(P, ctx, s);
        }
        if (p > 0) {
            // ---- arithmetic of symbol p-1 ----
            af_shrink(low, hm, cc, dd, TT);
            u32 k, m;
            const bool edge = af_renorm_counts(low, hm, k, m);
            if (__builtin_expect(edge || (k + pending > 32), 0)) {
                // literal loops of the reference, :126-150
                u64 lo = low, hi = (u64)hm + 1;
                while (hi < AF_HALF || lo > AF_HALF) {
                    if (hi < AF_HALF) {
                        wr.put(0, 1);
                        wr.put_run(1, pending);
                        lo <<= 1;
                        hi <<= 1;
                    } else {
                        wr.put(1, 1);
                        wr.put_run(0, pending);
                        lo = (lo - AF_HALF) << 1;
                        hi = (hi - AF_HALF) << 1;
                    }
                    pending = 0;
                }
                while (lo > AF_QTR && hi < 3ull * AF_QTR) {
                    pending += 1;
                    lo = (lo - AF_QTR) << 1;
                    hi = (hi - AF_QTR) << 1;
                }
                low = (u32)lo;
                hm = (u32)(hi - 1);
            } else {
                if (k > 0) {
                    // b0, then `pending` copies of !b0, then the other k-1 common bits
                    const u32 top = low >> (32 - k);
                    const u32 b0 = top >> (k - 1);
                    const u32 rest = top & ((1u << (k - 1)) - 1u);
                    const u32 pn = (u32)pending;  // <= 31 here
                    const u32 pat = (1u << pn) - (b0 ^ 1u);
                    wr.put((pat << (k - 1)) | rest, k + pn);
                    pending = 0;
                }
                pending += m;
                const u32 kt = k + m;  // <= 31
                low = (low << kt) & 0x7FFFFFFFu;
                hm = (hm << kt) | ((1u << kt) - 1u) | AF_HALF;
            }
        }
    }
    pending += 1;  // termination, :153-159
    if (low <= AF_QTR) {
        wr.put(0, 1);
        wr.put_run(1, pending);
    } else {
        wr.put(1, 1);
        wr.put_run(0, pending);
    }
    const u64 total = wr.finish();
    out_bit_off[chunk] = chunk * out_stride * 8;
    out_nbits[chunk] = (u32)total;
    if (status) status[chunk] = st;
}

// ---- forward bit reader: 16-byte loads one block ahead, bits past the end of the stream read as 0 -------------
struct AfReader {
    const uint4 *base;
    u64 nblk;      // readable 16-byte blocks
    u64 blk;       // next block to prefetch
    uint4 cur, pf;
    u32 wleft;     // words left in cur
    u64 win;       // bit window, left-aligned
    u32 nwin;      // valid bits in win (>= 32 between calls)
    i64 rem;       // stream bits not yet moved into the window (may go negative)

    __device__ __forceinline__ uint4 load(u64 j) const { return (j < nblk) ? base[j] : make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ u32 next_word() {
        if (wleft == 0) {
            cur = pf;
            pf = load(blk++);
            wleft = 4;
        }
        u32 v = __builtin_bswap32(cur.x);
        cur.x = cur.y;
        cur.y = cur.z;
        cur.z = cur.w;
        --wleft;
        if (rem < 32) v = (rem <= 0) ? 0u : (v & ~(0xFFFFFFFFu >> rem));
        rem -= 32;
        return v;
    }
    __device__ __forceinline__ void init(const u8 *in, u64 in_size_bytes, u64 bit_off, u32 nbits) {
        base = reinterpret_cast<const uint4 *>(in);
        nblk = in_size_bytes >> 4;
        const u64 b0 = bit_off >> 7;
        cur = load(b0);
        pf = load(b0 + 1);
        blk = b0 + 2;
        const u32 skipw = (u32)(bit_off >> 5) & 3u;
        wleft = 4;
        for (u32 i = 0; i < skipw; ++i) {
            cur.x = cur.y;
            cur.y = cur.z;
            cur.z = cur.w;
            --wleft;
        }
        const u32 skipb = (u32)bit_off & 31u;
        rem = (i64)nbits + skipb;
        const u64 hiw = next_word();
        const u64 low_ = next_word();
        win = (hiw << 32) | low_;
        nwin = 64;
        if (skipb) {  // drop the bits in front of the stream (>= 33 valid bits remain)
            win <<= skipb;
            nwin -= skipb;
        }
    }
    __device__ __forceinline__ u32 get(u32 nb) {  // nb <= 32
        if (nb == 0) return 0;
        const u32 v = (u32)(win >> (64 - nb));
        win <<= nb;
        nwin -= nb;
        if (nwin < 32) {
            win |= (u64)next_word() << (32 - nwin);
            nwin += 32;
        }
        return v;
    }
};

__global__ void __launch_bounds__(AF_THREADS)
    aec_fast_decode_kernel(AecFastDev P, const u8 *__restrict__ in, u64 in_size_bytes,
                           const u64 *__restrict__ bit_off, const u32 *__restrict__ in_nbits, u64 n_chunks,
                           u8 *__restrict__ out_sym, u64 out_stride, u32 out_cap, u32 *__restrict__ out_lens,
                           u32 *__restrict__ consumed, u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[AF_LDS_BYTES];
    const u32 tid = threadIdx.x;
    af_setup_tables(lds, P, tid);
    const u64 chunk = (u64)blockIdx.x * AF_THREADS + tid;
    if (chunk >= n_chunks) return;
    const u32 nbits = in_nbits[chunk];
    u32 st = 0;
    AfReader rd;
    rd.init(in, in_size_bytes, bit_off[chunk], nbits);
    u32 n = rd.get(32);
    if (nbits < 32) {
        st |= SCL_ST_TRUNCATED;
        n = 0;
    }
    out_lens[chunk] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    if (n == 0) {  // quirk Q5, as in scl_aec.hip
        consumed[chunk] = (st == 0) ? 32 + 2 : 0;
        if (status) status[chunk] = st;
        return;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out_sym + chunk * out_stride);
    u64 used = 32;
    u32 state = rd.get(32);
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 ctx = 0;
    AfRow R;
    R.a = *reinterpret_cast<const uint4_lds *>(lds + tid * 16);
    R.b = *reinterpret_cast<const uint4_lds *>(lds + AF_HALF_BYTES + tid * 16);
    uint4 ob = make_uint4(0, 0, 0, 0);
    u32 oword = 0;
    for (u32 i = 0;; ++i) {
        // ---- decode_step_core, :177-201 ----
        const u32 T = R.b.w >> 16;
        const double rdd = (double)(hm - low) + 1.0;
        const double xr = af_recip(rdd);
        const double Td = (double)T;
        // target = ((state - low + 1) * T - 1) // rng  (see scl_aec.hip), clamped for corrupt streams
        const double num = __builtin_fma((double)(state - low) + 1.0, Td, -0.5);
        u32 tgt = (u32)(num * xr);
        tgt = min(tgt, T - 1);
        const u32 tp = tgt | (tgt << 16);
        u32 acc = 0;
        acc = af_pk_count_gt(acc, tp, R.a.x);
        acc = af_pk_count_gt(acc, tp, R.a.y);
        acc = af_pk_count_gt(acc, tp, R.a.z);
        acc = af_pk_count_gt(acc, tp, R.a.w);
        acc = af_pk_count_gt(acc, tp, R.b.x);
        acc = af_pk_count_gt(acc, tp, R.b.y);
        acc = af_pk_count_gt(acc, tp, R.b.z);
        acc = af_pk_count_gt(acc, tp, R.b.w);
        // s = #{j : E[j] <= target} = 16 - #{E[j] > target}
        u32 s = 16u + (u32)((int32_t)(acc << 16) >> 16) + (u32)((int32_t)acc >> 16);
        s = min(s, P.K - 1);
        const u32 rowbase = ctx * AF_CTX_BYTES + tid * 16;
        const u32 sm1 = (s == 0) ? 0 : s - 1;
        const u32 c_raw = *reinterpret_cast<const u16_lds *>(lds + af_elem_addr(rowbase, sm1));
        const u32 d = *reinterpret_cast<const u16_lds *>(lds + af_elem_addr(rowbase, s));
        const u32 c = (s == 0) ? 0 : c_raw;
        const u32 nctx = af_next_ctx(P, ctx, s);
        const u32 nbase = nctx * AF_CTX_BYTES + tid * 16;
        AfRow Rn;  // issued before this row's write-back; patched below when it is the same row
        Rn.a = *reinterpret_cast<const uint4_lds *>(lds + nbase);
        Rn.b = *reinterpret_cast<const uint4_lds *>(lds + nbase + AF_HALF_BYTES);
        const AfRow R2 = af_row_plus_mask(R, lds, s);  // update_model
        *reinterpret_cast<uint4_lds *>(lds + rowbase) = R2.a;
        *reinterpret_cast<uint4_lds *>(lds + rowbase + AF_HALF_BYTES) = R2.b;
        const bool same = (nctx == ctx);
        R.a = same ? R2.a : Rn.a;
        R.b = same ? R2.b : Rn.b;
        ctx = nctx;
        af_shrink(low, hm, c, d, T);
        // ---- symbol out ----
        oword |= s << (8 * (i & 3));
        if ((i & 3) == 3) {
            const u32 q = (i >> 2) & 3;
            if (q == 0) ob.x = oword;
            if (q == 1) ob.y = oword;
            if (q == 2) ob.z = oword;
            if (q == 3) {
                ob.w = oword;
                dst[i >> 4] = ob;
            }
            oword = 0;
        }
        if (i + 1 == n) break;  // before the renormalisation, :242-243
        // ---- renormalisation, :245-275 ----
        u32 k, m;
        const bool edge = af_renorm_counts(low, hm, k, m);
        if (__builtin_expect(edge, 0)) {
            u64 lo = low, hi = (u64)hm + 1, stt = state;
            while (hi < AF_HALF || lo > AF_HALF) {
                if (hi < AF_HALF) {
                    lo <<= 1;
                    hi <<= 1;
                    stt <<= 1;
                } else {
                    lo = (lo - AF_HALF) << 1;
                    hi = (hi - AF_HALF) << 1;
                    stt = (stt - AF_HALF) << 1;
                }
                stt += rd.get(1);
                used++;
            }
            while (lo > AF_QTR && hi < 3ull * AF_QTR) {
                lo = (lo - AF_QTR) << 1;
                hi = (hi - AF_QTR) << 1;
                stt = (stt - AF_QTR) << 1;
                stt += rd.get(1);
                used++;
            }
            low = (u32)lo;
            hm = (u32)(hi - 1);
            state = (u32)stt;
        } else {
            const u32 kt = k + m;  // <= 31
            const u32 bits = rd.get(kt);
            const u32 keep = (state << k) & AF_HALF;
            state = (((state << kt) | bits) & 0x7FFFFFFFu) | keep;
            low = (low << kt) & 0x7FFFFFFFu;
            hm = (hm << kt) | ((1u << kt) - 1u) | AF_HALF;
            used += kt;
        }
    }
    // tail of the last (partial) 16-symbol group
    if ((n & 15) != 0) {
        const u32 q = ((n - 1) >> 2) & 3;
        if ((n & 3) != 0) {
            if (q == 0) ob.x = oword;
            if (q == 1) ob.y = oword;
            if (q == 2) ob.z = oword;
            if (q == 3) ob.w = oword;
        }
        if (q < 3) ob.w = 0;
        if (q < 2) ob.z = 0;
        if (q < 1) ob.y = 0;
        dst[(n - 1) >> 4] = ob;
    }
    // how many of the last PRECISION bits belonged to the encoder (:277-282)
    const u64 lo = low, hi = (u64)hm + 1;
    u32 e = 0;
    for (; e < 32; ++e) {
        const u64 slo = ((u64)state >> e) << e, shi = slo + (1ull << e);
        if (slo < lo || shi > hi) break;
    }
    if (e == 32) e = 31;
    consumed[chunk] = (u32)((i64)(used + 32) - ((i64)e - 1));
    if (status) status[chunk] = st;
}

// ---- host side ----------------------------------------------------------------------------------------------
bool aec_fast_ok(const scl_aec_model *m, u64 max_symbols) {
    const AecDev &d = m->dev;
    if (d.kind != SCL_MODEL_IID && d.kind != SCL_MODEL_ORDERK) return false;
    if (d.K < 2 || d.K > 16 || d.ctx_mod > 16 || d.P != 32 || d.size_bits != 32) return false;
    const u64 total_max = (u64)d.total0 + max_symbols;  // IID: total; ORDERK: bound on a row total and on any count
    if (total_max >= 32768 || total_max >= d.max_total) return false;
    return true;
}

static AecFastDev aec_fast_dev(const scl_aec_model *m) {
    AecFastDev f;
    f.K = m->dev.K;
    f.nctx = (u32)m->dev.ctx_mod;
    f.ctx_magic = (65536u + f.nctx - 1) / f.nctx;
    u32 E[16], acc = 0;
    for (u32 j = 0; j < 16; ++j) {
        if (j < f.K) acc += m->h_freq[j];
        E[j] = acc;
    }
    for (u32 r = 0; r < 8; ++r) f.initE[r] = E[2 * r] | (E[2 * r + 1] << 16);
    return f;
}

void aec_fast_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                            u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                            u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AF_THREADS - 1) / AF_THREADS);
    hipLaunchKernelGGL(aec_fast_encode_kernel, dim3(blocks), dim3(AF_THREADS), 0, st, aec_fast_dev(m), d_sym,
                       sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits,
                       d_status);
}

void aec_fast_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                            const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                            u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AF_THREADS - 1) / AF_THREADS);
    hipLaunchKernelGGL(aec_fast_decode_kernel, dim3(blocks), dim3(AF_THREADS), 0, st, aec_fast_dev(m), d_in,
                       in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                       d_consumed, d_status);
}
