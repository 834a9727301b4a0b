// scl_aec_fast.hip -- adaptive arithmetic coding with per-lane context tables in LDS (BASELINE.json configs[3]),
// one wavefront lane per chunk.  Same streams, bit for bit, as scl_aec.hip and the reference:
//   ArithmeticEncoder.shrink_range / encode_block   scl/compressors/arithmetic_coding.py:58-78, :80-161
//   ArithmeticDecoder.decode_step_core / decode_block                               :177-201, :203-287
//   AdaptiveIIDFreqModel / AdaptiveOrderKFreqModel   scl/compressors/probability_models.py:70-92, :95-160
//
// Served models (aec_fast_ok): PRECISION = 32, adaptive IID or order-k model with
// alphabet 2..16 and at most 16 contexts (K^k <= 16: order-1 K <= 16, order-2 K <= 4, ...), totals that stay
// below 2^15 and below the model's rescale threshold for the whole chunk.  Everything else -> scl_aec.hip.
//
// What is different from the generic kernel (13.7 / 21.7 ms per 256 MiB of order-1 K = 16 data, now 1.9 / 2.6 ms):
//  * Model layout.  A context row is 16 u16 EXCLUSIVE cumulative counts X[j] = count[0] + .. + count[j-1]
//    (entries past the alphabet hold the total), 32 bytes per row, row `ctx` of thread t at LDS [ctx][t], plus a
//    u32 total per row: c = X[s], d = X[s+1] (or the total) are two 2-byte reads at one address, the update
//    `count[s] += 1` is X[j] += 1 for j > s = eight v_pk_add_u16 on the row with a mask row from a 512-byte LUT
//    and one ds_add_u32 on the total, and the decoder's search max{s : c[s] <= target} is a packed
//    compare-and-count over the 8 row registers.  16 contexts x 32 B x 256 lanes = 128 KiB (+ 16 KiB totals):
//    one workgroup of 256 lanes per CU, i.e. ONE wave per SIMD -- the kernels are bound by how fast a lone wave
//    issues instructions (~6 clk each, profiles/r01_ubench_dp_issue.txt), so every instruction counts.
//  * Arithmetic.  (rng*c)//T and ((state-low+1)*T-1)//rng are evaluated in binary64: every operand is an integer
//    below 2^47, q = trunc((num + 0.5) * x) with x = 1/den refined by two Newton steps from v_rcp_f32 is exact
//    because the quotient's error (< 2^-18 resp. 2^-35) is below the distance 0.5/den of (num + 0.5)/den from the
//    nearest integer (den < 2^15 resp. <= 2^32).  low and high-1 are kept as u32.
//  * Renormalisation in closed form.  With hm = high - 1: k = clz(low ^ hm) E1/E2 steps emit the k common leading
//    bits, then m = number of leading (1,0) bit pairs of (low, hm) below them E3 steps.  The reference's STRICT
//    comparisons (quirk Q1: `high < HALF`, `low > HALF`, `low > QTR and high < 3*QTR`) differ from this only when
//    low or high hits a power-of-two boundary inside the shifted-out prefix, i.e. when
//    ctz(low) + k + m + 1 >= 32 or ctz(high) + k + m + 1 >= 32; those symbols (rare after the first few of a
//    chunk) and runs of more than 32 - k pending bits take the literal loops of the reference.
//  * Encoder software pipeline: the model reads of symbol i+1 are in flight while symbol i is coded; words go
//    straight to the slot (4-byte stores, L2 merges them); symbols arrive four per 32-bit load, one word ahead,
//    the load unconditional so that it is not waited for at once.
#include <stdlib.h>
#include <type_traits>

#include "scl_aec_internal.h"
#include "scl_aec_math.h"
#include "scl_aec_lane_io.h"

#define AF_THREADS 256
#define AF_CTX_BYTES (AF_THREADS * 32)      // one 32-byte row of every thread
#define AF_TABLE_BYTES (16 * AF_CTX_BYTES)  // 128 KiB
#define AF_TOT_BASE AF_TABLE_BYTES          // u32 totals [ctx][thread]
#define AF_TOT_BYTES (16 * AF_THREADS * 4)
#define AF_LUT_BASE (AF_TOT_BASE + AF_TOT_BYTES)
#define AF_LUT_BYTES 512
#define AF_LDS_BYTES (AF_LUT_BASE + AF_LUT_BYTES)

struct AecFastDev {
    u32 K;          // alphabet size 2..16
    u32 nctx;       // K^k <= 16
    u32 ctx_magic;  // ceil(2^16 / nctx): (v * magic) >> 16 == v / nctx for v < 272
    u32 total0;     // initial total of a row
    u32 size_bits;  // DATA_BLOCK_SIZE_BITS (1..32)
    u32 initX[8];   // 16 packed u16: EXCLUSIVE cumulative initial counts X[j] = sum_{i<j}, padded with the total
};

struct AfRow {
    uint4 a, b;
};

// LDS image of one workgroup: rows (exclusive cumulative counts, 32 bytes per context and thread), row totals,
// and the update masks LUT[s][j] = (j > s)
__device__ __forceinline__ void af_setup_tables(char *lds, const AecFastDev &P, u32 tid) {
    if (tid < 128) {
        const u32 s = tid >> 3, r = tid & 7;  // register r holds elements 2r, 2r+1
        const u32 v = ((2 * r > s) ? 1u : 0u) | ((2 * r + 1 > s) ? 0x10000u : 0u);
        *reinterpret_cast<u32_lds *>(lds + AF_LUT_BASE + s * 32 + r * 4) = v;
    }
    const uint4 a = make_uint4(P.initX[0], P.initX[1], P.initX[2], P.initX[3]);
    const uint4 b = make_uint4(P.initX[4], P.initX[5], P.initX[6], P.initX[7]);
    for (u32 c = 0; c < P.nctx; ++c) {
        *reinterpret_cast<uint4_lds *>(lds + c * AF_CTX_BYTES + tid * 32) = a;
        *reinterpret_cast<uint4_lds *>(lds + c * AF_CTX_BYTES + tid * 32 + 16) = b;
        *reinterpret_cast<u32_lds *>(lds + AF_TOT_BASE + c * (AF_THREADS * 4) + tid * 4) = P.total0;
    }
    __syncthreads();
}

template <bool ORDER1>
__device__ __forceinline__ u32 af_next_ctx(const AecFastDev &P, u32 ctx, u32 s) {  // past_k[1:] + [s], :146-151
    if (ORDER1) return s;
    // (24-bit multiplies: v < 272, magic <= 2^15, nctx <= 16 -- the 32-bit v_mul_lo_u32 is a quarter-rate instruction)
    const u32 v = __umul24(ctx, P.K) + s;
    return v - __umul24(__umul24(v, P.ctx_magic) >> 16, P.nctx);
}
__device__ __forceinline__ AfRow af_row_load(const char *lds, u32 rowbase) {
    AfRow R;
    R.a = *reinterpret_cast<const uint4_lds *>(lds + rowbase);
    R.b = *reinterpret_cast<const uint4_lds *>(lds + rowbase + 16);
    return R;
}
// update_model with the mask row already loaded
__device__ __forceinline__ void af_row_store_updated(const AfRow &R, const uint4 &ia, const uint4 &ib, char *lds,
                                                     u32 rowbase, u32 totaddr) {
    *reinterpret_cast<uint4_lds *>(lds + rowbase) =
        make_uint4(af_pk_add(R.a.x, ia.x), af_pk_add(R.a.y, ia.y), af_pk_add(R.a.z, ia.z), af_pk_add(R.a.w, ia.w));
    *reinterpret_cast<uint4_lds *>(lds + rowbase + 16) =
        make_uint4(af_pk_add(R.b.x, ib.x), af_pk_add(R.b.y, ib.y), af_pk_add(R.b.z, ib.z), af_pk_add(R.b.w, ib.w));
    __hip_atomic_fetch_add(reinterpret_cast<u32 *>(lds + totaddr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// update_model: count[s] += 1  <=>  X[j] += 1 for j > s, total += 1
__device__ __forceinline__ AfRow af_row_update(const AfRow &R, char *lds, u32 rowbase, u32 totaddr, u32 s) {
    const uint4 ia = *reinterpret_cast<const uint4_lds *>(lds + AF_LUT_BASE + s * 32);
    const uint4 ib = *reinterpret_cast<const uint4_lds *>(lds + AF_LUT_BASE + s * 32 + 16);
    AfRow o;
    o.a = make_uint4(af_pk_add(R.a.x, ia.x), af_pk_add(R.a.y, ia.y), af_pk_add(R.a.z, ia.z), af_pk_add(R.a.w, ia.w));
    o.b = make_uint4(af_pk_add(R.b.x, ib.x), af_pk_add(R.b.y, ib.y), af_pk_add(R.b.z, ib.z), af_pk_add(R.b.w, ib.w));
    *reinterpret_cast<uint4_lds *>(lds + rowbase) = o.a;
    *reinterpret_cast<uint4_lds *>(lds + rowbase + 16) = o.b;
    __hip_atomic_fetch_add(reinterpret_cast<u32 *>(lds + totaddr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return o;
}

template <bool ORDER1>
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
    const u32 *src = reinterpret_cast<const u32 *>(sym + chunk * sym_stride);
    AfWriter wr;
    wr.init(out + chunk * out_stride);
    wr.put(P.size_bits < 32 ? (n & ((1u << P.size_bits) - 1u)) : n, P.size_bits);  // header, :92-99
    u32 st = (P.size_bits < 32 && (n >> P.size_bits)) ? SCL_ST_SIZE : 0u;
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 pending = 0;  // E3 steps not yet resolved (<= 32 * n < 2^20)
    u32 ctx = 0;
    u32 nextw = 0;
    const u32 last_word = n ? (n - 1) >> 2 : 0;
    u32 c_nx = 0, d_nx = 1, T_nx = 1;
    double x_nx = 1.0;

    // model stage, split in two so that the arithmetic of the previous symbol runs while the LDS reads are in
    // flight: model_issue = freqs_current lookup for symbol s (loads only), model_finish = update_model (:118)
    // and 1/T.  Everything the arithmetic of a symbol needs (c, d, T, 1/T) is ready one iteration ahead.
    u32 m_rowbase = 0, m_totaddr = 0, m_s = 0, m_draw = 0;
    AfRow m_R;
    uint4 m_ia, m_ib;
    auto model_issue = [&](u32 s) {
        m_rowbase = ctx * AF_CTX_BYTES + tid * 32;
        m_totaddr = AF_TOT_BASE + ctx * (AF_THREADS * 4) + tid * 4;
        m_s = s;
        const u32 ea = m_rowbase + 2 * s;
        c_nx = *reinterpret_cast<const u16_lds *>(lds + ea);
        m_draw = *reinterpret_cast<const u16_lds *>(lds + ea + 2);
        T_nx = *reinterpret_cast<const u32_lds *>(lds + m_totaddr);
        m_R = af_row_load(lds, m_rowbase);
        m_ia = *reinterpret_cast<const uint4_lds *>(lds + AF_LUT_BASE + s * 32);
        m_ib = *reinterpret_cast<const uint4_lds *>(lds + AF_LUT_BASE + s * 32 + 16);
        ctx = af_next_ctx<ORDER1>(P, ctx, s);
    };
    auto model_finish = [&]() {
        af_row_store_updated(m_R, m_ia, m_ib, lds, m_rowbase, m_totaddr);
        d_nx = (m_s == 15) ? T_nx : m_draw;
        x_nx = af_recip((double)T_nx);
    };
    // arithmetic stage: shrink_range (:58-78) and the renormalisation loops (:126-150) of one symbol
    auto code = [&](u32 cc, u32 dd, u32 TT, double xx) {
        af_shrink(low, hm, cc, dd, TT, xx);
        u32 k, m;
        const bool edge = af_renorm_counts(low, hm, k, m);
        if (__builtin_expect(edge || (k + pending > 32), 0)) {
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
                const u32 pat = (1u << pending) - (b0 ^ 1u);  // pending <= 31 here
                wr.put((pat << (k - 1)) | rest, k + pending);
                pending = 0;
            }
            pending += m;
            const u32 kt = k + m;  // <= 31
            low = (low << kt) & 0x7FFFFFFFu;
            hm = (hm << kt) | ((1u << kt) - 1u) | AF_HALF;
        }
    };

    // Before the first symbol (c, d, T, 1/T) = (0, 1, 1, 1.0) makes the arithmetic stage a no-op (d == T keeps
    // high, low += 0, nothing to renormalise), so every symbol runs the same body and one more `code` drains it.
    const u32 n_words = (n + 3) >> 2;
    if (n > 0) nextw = src[0];
    for (u32 w = 0; w < n_words; ++w) {
        // four symbols per 32-bit load, one word ahead.  The load is unconditional with its index clamped into
        // the chunk and sits where nothing has to be merged with it: a load under a lane-dependent or periodic
        // condition is copied into the loop-carried register at once, i.e. waited for at once (~1 us each).
        u32 word = nextw;
        nextw = src[min(w + 1, last_word)];
        const u32 cnt = min(4u, n - 4 * w);
#pragma unroll 1
        for (u32 j = 0; j < cnt; ++j) {
            u32 s = word & 0xFFu;
            word >>= 8;
            if (s >= P.K) {
                st |= SCL_ST_SYMBOL;
                s = 0;
            }
            const u32 cc = c_nx, dd = d_nx, TT = T_nx;
            const double xx = x_nx;
            model_issue(s);        // this symbol: LDS reads in flight ...
            code(cc, dd, TT, xx);  // ... while the previous symbol is coded
            model_finish();
        }
    }
    code(c_nx, d_nx, T_nx, x_nx);
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

// ---- decoder (round 4) ---------------------------------------------------------------------------------------------
// Its own table layout: a context row is 16 u16 INCLUSIVE cumulative counts Y[j] = count[0] + .. + count[j] (entries past
// the alphabet hold the total, so Y[15] IS the row total: no separate total array, no second LDS update, no clamp of the
// searched symbol).  c = Y[s-1] (0 for s = 0), d = Y[s]; update_model is Y[j] += 1 for j >= s = eight v_pk_add_u16 with a
// mask row from the 512-byte LUT; the search max{s : c[s] <= target} is 16 - #{j : Y[j] > target}.
// Symbols leave through 64 bytes of LDS per lane (the 16 KiB the totals used to take) as whole 64-byte sectors -- four
// back-to-back 16-byte stores every 64 symbols -- instead of one 4-byte store per four symbols, which the memory system
// did not merge at 262 144 open lines: 12.6 GB of HBM writes for 1 GiB of symbols (profiles/traffic.json, round 3).
#define AD_ROW_BASE 16                             // rows start 16 bytes in: the address "two bytes before a row" is never negative
#define AD_OUT_BASE (AD_ROW_BASE + AF_TABLE_BYTES)  // [thread][64 bytes]
#define AD_OUT_BYTES (AF_THREADS * 64)
#define AD_LDS_BYTES (AD_OUT_BASE + AD_OUT_BYTES)

__device__ __forceinline__ void ad_setup_tables(char *lds, const AecFastDev &P, u32 tid) {
    // inclusive from the exclusive initX: Y[j] = X[j + 1], Y[15] = total
    u32 y[8];
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const u32 lo = P.initX[r] >> 16;                                   // X[2r + 1]
        const u32 hi = (r < 7) ? (P.initX[r + 1] & 0xFFFFu) : P.total0;    // X[2r + 2]
        y[r] = lo | (hi << 16);
    }
    const uint4 a = make_uint4(y[0], y[1], y[2], y[3]);
    const uint4 b = make_uint4(y[4], y[5], y[6], y[7]);
    for (u32 c = 0; c < P.nctx; ++c) {
        *reinterpret_cast<uint4_lds *>(lds + AD_ROW_BASE + c * AF_CTX_BYTES + tid * 32) = a;
        *reinterpret_cast<uint4_lds *>(lds + AD_ROW_BASE + c * AF_CTX_BYTES + tid * 32 + 16) = b;
    }
    __syncthreads();
}

template <bool ORDER1>
__global__ void __launch_bounds__(AF_THREADS)
    aec_fast_decode_kernel(AecFastDev P, const u8 *__restrict__ in, u64 in_size_bytes,
                           const u64 *__restrict__ bit_off, const u32 *__restrict__ in_nbits, u64 n_chunks,
                           u8 *__restrict__ out_sym, u64 out_stride, u32 out_cap, u32 *__restrict__ out_lens,
                           u32 *__restrict__ consumed, u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[AD_LDS_BYTES];
    const u32 tid = threadIdx.x;
    ad_setup_tables(lds, P, tid);
    const u64 chunk = (u64)blockIdx.x * AF_THREADS + tid;
    if (chunk >= n_chunks) return;
    const u32 nbits = in_nbits[chunk];
    u32 st = 0;
    AfReader rd;
    rd.init(in, in_size_bytes, bit_off[chunk], nbits);
    u32 n = rd.get(P.size_bits);
    if (nbits < P.size_bits) {
        st |= SCL_ST_TRUNCATED;
        n = 0;
    }
    out_lens[chunk] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    if (n == 0) {  // quirk Q5, as in scl_aec.hip
        consumed[chunk] = (st == 0) ? P.size_bits + 2 : 0;
        if (status) status[chunk] = st;
        return;
    }
    AfSymOut so;
    so.init(lds + AD_OUT_BASE, tid, out_sym + chunk * out_stride);
    u32 state = rd.get(32);
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 ctx = 0;
    AfRow R = af_row_load(lds, AD_ROW_BASE + tid * 32);
    const u32 lane_m2 = AD_ROW_BASE + tid * 32 - 2;  // two bytes before the lane's row of context 0
    u32 row_m2 = lane_m2;                            // the same for the current context: carried from symbol to symbol
    // One symbol: decode_step_core (:177-201), update_model, symbol out.  The loop runs it for all but the last symbol of the
    // chunk, the last one follows the loop without a renormalisation (the reference breaks before it, :242-243): a single
    // exit test per iteration.
    auto step = [&](u32 i) {
        // T = Y[15] < 2^15, as float (one SDWA conversion out of the packed pair) and as double (from the float): both exact
        float Tf;
        asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(Tf) : "v"(R.b.w));
        const double Td = (double)Tf;
        const double xr = af_recip((double)(hm - low) + 1.0);
        // target = ((state - low + 1) * T - 1) // rng  (see scl_aec.hip).  low <= state <= hm holds for ANY input bits (the
        // symbol chosen is the one whose interval holds the state), so target <= T - 1; the guard is on s below.
        const double num = __builtin_fma((double)(state - low) + 1.0, Td, -0.5);
        const u32 tgt = (u32)(num * xr);
        // s = #{j : Y[j] <= target}; Y[15] = T > target, so s <= 15, and entries past the alphabet hold T as well, so
        // s <= K - 1
        const u32 Y[8] = {R.a.x, R.a.y, R.a.z, R.a.w, R.b.x, R.b.y, R.b.z, R.b.w};
        u32 msk[8];
        const u32 s = min(af_pk_search16(Y, tgt, msk), P.K - 1);
        // c = Y[s - 1], d = Y[s]: two u16 reads, issued before the row is rewritten; s = 0 reads the two bytes in front of
        // the row, masked away below.  All addresses of the step hang off "row - 2": one add fewer than with "row" and "- 2".
        const u32 ea_m2 = row_m2 + 2 * s;
        u32 c_raw = *reinterpret_cast<const u16_lds *>(lds + ea_m2);
        u32 ea_d = ea_m2;
        asm("" : "+v"(ea_d));  // hides that the two reads are adjacent (the compiler would merge them into one unaligned read)
        u32 d_raw = *reinterpret_cast<const u16_lds *>(lds + ea_d + 2);
        // update_model: Y[j] += 1 for j >= s, i.e. minus the search's masks
        *reinterpret_cast<uint4_lds *>(lds + row_m2 + 2) =
            make_uint4(af_pk_sub(Y[0], msk[0]), af_pk_sub(Y[1], msk[1]), af_pk_sub(Y[2], msk[2]), af_pk_sub(Y[3], msk[3]));
        *reinterpret_cast<uint4_lds *>(lds + row_m2 + 18) =
            make_uint4(af_pk_sub(Y[4], msk[4]), af_pk_sub(Y[5], msk[5]), af_pk_sub(Y[6], msk[6]), af_pk_sub(Y[7], msk[7]));
        ctx = af_next_ctx<ORDER1>(P, ctx, s);
        // next symbol's row: issued now, needed only after the arithmetic below
        asm("v_lshl_add_u32 %0, %1, 13, %2" : "=v"(row_m2) : "v"(ctx), "v"(lane_m2));  // ctx * AF_CTX_BYTES + lane_m2, one op
        R = af_row_load(lds, row_m2 + 2);
        // (the empty asm statements around 1/T place its seven instructions after the search and before the wait for c, d)
        float T2 = Tf;
        asm volatile("" : "+v"(T2) : "v"(s));
        double xT = af_recip_fd(T2, Td);
        asm volatile("" : "+v"(xT));
        asm volatile("" : "+v"(c_raw), "+v"(d_raw));
        const u32 c = c_raw & ~msk[0];
        af_shrink2_d(low, hm, (double)c, (double)d_raw, xT);
        so.put(s, i);  // four to a word, sixteen words to a 64-byte sector (AfSymOut)
    };
    // ---- renormalisation, :245-275; UNCHECKED selects the reader's refill (AfReader::next_word) ----
    auto renorm = [&](auto unchecked) {
        constexpr bool UC = decltype(unchecked)::value;
        u32 k, m, nlow, nhm;
        const bool edge = af_renorm2_dec(low, hm, k, m, nlow, nhm);
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
                stt += rd.get<UC>(1);
            }
            while (lo > AF_QTR && hi < 3ull * AF_QTR) {
                lo = (lo - AF_QTR) << 1;
                hi = (hi - AF_QTR) << 1;
                stt = (stt - AF_QTR) << 1;
                stt += rd.get<UC>(1);
            }
            low = (u32)lo;
            hm = (u32)(hi - 1);
            state = (u32)stt;
        } else {
            const u32 kt = k + m;  // <= 31
            state = af_state_shift_in<UC>(rd, state, k, kt);
            low = nlow;
            hm = nhm;
        }
    };
    // The symbol index i is the same for every lane still at work (they start together and only drop out), so it lives in a
    // scalar register.  Stretches: as many symbols as EVERY working lane can decode without reaching the last word of its
    // stream (and has left to decode) run with the unchecked refill and a scalar trip count; the last symbols of the chunks
    // -- and everything, in a wave that holds a very short stream -- run in the checked loop below.
    u32 i = 0;
    for (;;) {
        const bool at_work = i + 1 < n;
        const u32 mine = at_work ? min(rd.safe_symbols(), n - 1 - i) : 0xFFFFFFFFu;
        const u32 cnt = __builtin_amdgcn_readfirstlane(af_wave_min(mine));
        if (cnt == 0xFFFFFFFFu || cnt < 16) break;
        if (at_work) {
            const u32 *before = rd.ptr;
            // four symbols per trip once the index is a multiple of four: which byte of the output word a symbol fills, and
            // when the word is complete, are then known at compile time (AfSymOut::put: no scalar compare-and-branch per
            // symbol), and the trip count is tested once per four
            u32 u = 0;
            for (; u < cnt && ((i + u) & 3u); ++u) {
                step(i + u);
                renorm(std::true_type{});
            }
            for (; u + 4 <= cnt; u += 4) {
                const u32 base = i + u;
                __builtin_assume((base & 3u) == 0);
#pragma unroll
                for (u32 q = 0; q < 4; ++q) {
                    step(base + q);
                    renorm(std::true_type{});
                }
            }
            for (; u < cnt; ++u) {
                step(i + u);
                renorm(std::true_type{});
            }
            rd.settle(before);
        }
        i += cnt;
    }
    for (; i + 1 < n; ++i) {
        step(i);
        renorm(std::false_type{});
    }
    step(n - 1);
    so.finish(n);
    // how many of the last PRECISION bits belonged to the encoder (:277-282)
    const u64 lo = low, hi = (u64)hm + 1;
    u32 e = 0;
    for (; e < 32; ++e) {
        const u64 slo = ((u64)state >> e) << e, shi = slo + (1ull << e);
        if (slo < lo || shi > hi) break;
    }
    if (e == 32) e = 31;
    consumed[chunk] = (u32)((i64)rd.position() - ((i64)e - 1));
    if (status) status[chunk] = st;
}

// ---- host side ----------------------------------------------------------------------------------------------
bool aec_fast_ok(const scl_aec_model *m, u64 max_symbols) {
    const AecDev &d = m->dev;
    if (d.kind != SCL_MODEL_IID && d.kind != SCL_MODEL_ORDERK) return false;
    if (d.K < 2 || d.K > 16 || d.ctx_mod > 16 || d.P != 32) return false;
    const u64 total_max = (u64)d.total0 + max_symbols;  // IID: total; ORDERK: bound on a row total and on any count
    if (total_max >= 32768 || total_max >= d.max_total) return false;
    return true;
}

static AecFastDev aec_fast_dev(const scl_aec_model *m) {
    AecFastDev f;
    f.K = m->dev.K;
    f.nctx = (u32)m->dev.ctx_mod;
    f.ctx_magic = (65536u + f.nctx - 1) / f.nctx;
    u32 X[16], acc = 0;
    for (u32 j = 0; j < 16; ++j) {
        X[j] = acc;  // exclusive; entries past the alphabet hold the total
        if (j < f.K) acc += m->h_freq[j];
    }
    f.total0 = acc;
    f.size_bits = m->dev.size_bits;
    for (u32 r = 0; r < 8; ++r) f.initX[r] = X[2 * r] | (X[2 * r + 1] << 16);
    return f;
}

void aec_fast_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                            u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                            u32 *d_status, hipStream_t st) {
    // round 3: the two-role encoder (scl_aec_split.hip) serves every batch; SCL_AEC_ENC=lane keeps the one-lane-per-
    // chunk kernel below for A/B timing and as a second implementation the tests compare against
    const char *enc_env = getenv("SCL_AEC_ENC");  // read at every call, like the other switches
    const bool lane_kernel = enc_env && (enc_env[0] == 'l' || enc_env[0] == 'L');
    if (!lane_kernel) {
        (void)aec_split_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride,
                                      d_out_bit_offset, d_out_nbits, d_status, st);
        return;
    }
    const u32 blocks = (u32)((n_chunks + AF_THREADS - 1) / AF_THREADS);
    if (m->dev.k == 1)
        hipLaunchKernelGGL(aec_fast_encode_kernel<true>, dim3(blocks), dim3(AF_THREADS), 0, st, aec_fast_dev(m),
                           d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                           d_out_nbits, d_status);
    else
        hipLaunchKernelGGL(aec_fast_encode_kernel<false>, dim3(blocks), dim3(AF_THREADS), 0, st, aec_fast_dev(m),
                           d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                           d_out_nbits, d_status);
}

void aec_fast_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                            const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                            u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AF_THREADS - 1) / AF_THREADS);
    if (m->dev.k == 1)
        hipLaunchKernelGGL(aec_fast_decode_kernel<true>, dim3(blocks), dim3(AF_THREADS), 0, st, aec_fast_dev(m), d_in,
                           in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
    else
        hipLaunchKernelGGL(aec_fast_decode_kernel<false>, dim3(blocks), dim3(AF_THREADS), 0, st, aec_fast_dev(m),
                           d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
}
