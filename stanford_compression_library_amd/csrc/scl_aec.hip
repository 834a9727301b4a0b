// scl_aec.hip -- batched finite-precision arithmetic coder with pluggable frequency models for
// gfx950, one wavefront lane per chunk (one coder + one private model per lane).
//
// Replaces reference scl/compressors/arithmetic_coding.py
//   AECParams :20-38, ArithmeticEncoder.shrink_range :58-78, encode_block :80-161,
//   ArithmeticDecoder.decode_step_core :177-201, decode_block :203-287
// and scl/compressors/probability_models.py
//   FixedFreqModel :57-67, AdaptiveIIDFreqModel :70-92, AdaptiveOrderKFreqModel :95-160.
// Stream layout per chunk: [n : DATA_BLOCK_SIZE_BITS][renormalisation bits ...][termination bits].
//
// Reference behaviours reproduced on purpose:
//   * strict comparisons `high < HALF`, `low > HALF`, `low > QTR and high < 3*QTR`, final `low <= QTR`;
//   * the model is updated after shrink_range and before renormalisation; the decoder stops before the
//     last renormalisation and then works out how many of the last PRECISION bits were its own;
//   * one chunk == one FRESH model (the reference keeps model state across encode_block calls of one
//     object; batched chunks correspond to one new encoder object per chunk).
// Deliberately not inherited (quirks Q3/Q5): the 2^(2^32)-bit assert; decoding an empty block never
// terminates in the reference -- here it returns the header plus the two termination bits.
//
// The decoder's vector search  max{s : low + (c[s]*rng)//T <= state}  (:195-200) is evaluated through the
// exact integer equivalence  c[s] <= ((state - low + 1)*T - 1) // rng  and a scan of the model row.
//
// Model state lives in caller-provided device scratch, one private region per chunk:
//   IID    : K   u32 counts, initialised by the lane from the initial frequencies
//   ORDERK : K^(k+1) u32 cells holding (count - 1); the host zero-fills the scratch (= all-ones counts,
//            probability_models.py:110) with one hipMemsetAsync before the launch.  Models with more than 256 cells
//            store every row in two levels -- 16 block totals, then the counts in blocks of 16, all as
//            (count - 1) -- so that a lookup is two 64-byte reads issued together (ONE memory round trip; these
//            tables live in HBM/L2 and the kernel is bound by that latency), an update two stores, and the
//            decoder's search two dependent 64-byte reads, instead of a scan of the whole row.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "scl_aec_internal.h"

// Per-lane frequency model (one of the three kinds); the coders only see counts through rd()/wr().
// LDS16 = true: the adaptive counts of this lane live in LDS as u16 (cell c of thread t at [c][t], so a wave's
// accesses spread over all banks whatever cells the lanes touch) -- "per-lane context tables in LDS",
// BASELINE.json configs[3].  Used when K^(k+1) <= 256 cells and the counts cannot exceed 16 bits.
// LDS16 = false: counts are u32 in caller-provided global scratch (any K, k).
#define AEC_LDS_CELLS 256
template <bool LDS16>
struct LaneModel {
    const AecDev *P;
    u32 *cnt;    // private global scratch (IID / ORDERK): ORDERK stores count - 1 (zero-filled by the host)
    u16 *lcnt;   // this thread's column of the LDS table: actual counts
    u64 ctx;     // ORDERK: flattened index of the last k symbol indices (starts at 0, :116)
    u32 bad;     // rescale branch of the order-k model reached (raises AxisError in the reference)

    __device__ __forceinline__ u32 rd(u64 cell) const {
        if (LDS16) return lcnt[cell * 256];
        return cnt[cell] + ((P->kind == SCL_MODEL_ORDERK) ? 1u : 0u);
    }
    __device__ __forceinline__ void wr(u64 cell, u32 v) {
        if (LDS16)
            lcnt[cell * 256] = (u16)v;
        else
            cnt[cell] = v - ((P->kind == SCL_MODEL_ORDERK) ? 1u : 0u);
    }
    // ctx_state != nullptr: coder `chunk` CONTINUES from the counts already in its scratch region and from
    // ctx_state[chunk] (the reference's freq_model object lives across encode_block calls, quirk Q4); the
    // kernels store the context back when the block is done.  Only with LDS16 = false.
    __device__ __forceinline__ void init(const AecDev *P_, u32 *scratch, u64 chunk, u16 *lds_col,
                                         const u64 *ctx_state = nullptr) {
        P = P_;
        ctx = ctx_state ? ctx_state[chunk] : 0;
        bad = 0;
        lcnt = lds_col;
        cnt = (!LDS16 && scratch) ? scratch + chunk * P_->cells : nullptr;
        if (!LDS16 && ctx_state) return;
        if (LDS16) {
            for (u64 j = 0; j < P->cells; ++j) lcnt[j * 256] = (u16)((P->kind == SCL_MODEL_IID) ? P->d_freq[j] : 1u);
        } else if (P->kind == SCL_MODEL_IID) {
            for (u32 j = 0; j < P->K; ++j) cnt[j] = P->d_freq[j];
        }
    }
    __device__ __forceinline__ u64 row_base() const {
        return (P->kind == SCL_MODEL_ORDERK) ? ctx * ((!LDS16 && P->fenwick) ? P->row_cells : P->K) : 0;
    }
    // cumulative count below s, frequency of s and total of the current distribution (freqs_current)
    __device__ __forceinline__ void lookup(u32 s, const u32 *s_f, const u32 *s_c, u64 &c, u64 &f, u64 &T) const {
        if (P->kind == SCL_MODEL_FIXED) {
            c = s_c[s];
            f = s_f[s];
            T = P->total0;
            return;
        }
        const u64 rb = row_base();
        if (!LDS16 && P->fenwick) {  // count[j] = 1 + extra[j]
            u32 bt[16], cb[16];
            load16(bt, rb);
            load16(cb, rb + 16 + (s & ~15u));
            const u32 b = s >> 4, w = s & 15u;
            u64 below = 0, tot = 0, fs = 0;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                tot += bt[j];
                below += (j < b) ? bt[j] : 0u;
                below += (j < w) ? cb[j] : 0u;
                fs = (j == w) ? cb[j] : fs;
            }
            c = s + below;
            f = 1 + fs;
            T = P->K + tot;
            return;
        }
        u64 acc = 0, cs = 0, fs = 0;
        for (u32 j = 0; j < P->K; ++j) {
            const u64 v = rd(rb + j);
            if (j == s) {
                cs = acc;
                fs = v;
            }
            acc += v;
        }
        c = cs;
        f = fs;
        T = acc;
    }
    // 16 consecutive cells (64 bytes, 64-byte aligned) as four 16-byte loads issued back to back
    __device__ __forceinline__ void load16(u32 (&v)[16], u64 cell) const {
        const uint4 *p = reinterpret_cast<const uint4 *>(cnt + cell);
        const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        v[0] = q0.x, v[1] = q0.y, v[2] = q0.z, v[3] = q0.w, v[4] = q1.x, v[5] = q1.y, v[6] = q1.z, v[7] = q1.w;
        v[8] = q2.x, v[9] = q2.y, v[10] = q2.z, v[11] = q2.w, v[12] = q3.x, v[13] = q3.y, v[14] = q3.z, v[15] = q3.w;
    }
    __device__ __forceinline__ u64 total(const u32 *) const {
        if (P->kind == SCL_MODEL_FIXED) return P->total0;
        if (!LDS16 && P->fenwick) {
            u32 bt[16];
            load16(bt, row_base());
            u64 tot = 0;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) tot += bt[j];
            return P->K + tot;
        }
        const u64 rb = row_base();
        u64 acc = 0;
        for (u32 j = 0; j < P->K; ++j) acc += rd(rb + j);
        return acc;
    }
    // largest s with cum[s] <= cmax; returns s and its (c, f)
    __device__ __forceinline__ u32 search(u64 cmax, const u32 *s_f, const u32 *s_c, u64 &c, u64 &f) const {
        if (P->kind == SCL_MODEL_FIXED) {
            u32 lo = 0, hi = P->K;
            while (hi - lo > 1) {
                const u32 mid = (lo + hi) >> 1;
                if ((u64)s_c[mid] <= cmax)
                    lo = mid;
                else
                    hi = mid;
            }
            c = s_c[lo];
            f = s_f[lo];
            return lo;
        }
        const u64 rb = row_base();
        if (!LDS16 && P->fenwick) {  // largest s <= K-1 with c[s] = s + extras below s <= cmax: block, then symbol
            u32 bt[16], cb[16];
            load16(bt, rb);
            const u32 nblk = (P->K + 15) >> 4;
            u32 b = 0;
            u64 g = 0, run = 0;  // run = c[16 * j] while scanning
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                if (j < nblk && run <= cmax) {
                    b = j;
                    g = run;
                }
                run += 16 + bt[j];
            }
            load16(cb, rb + 16 + 16 * b);
            const u32 wmax = min(15u, P->K - 1 - 16 * b);
            u32 w = 0;
            u64 gw = g, fw = cb[0];
            run = g;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                if (j <= wmax && run <= cmax) {
                    w = j;
                    gw = run;
                    fw = cb[j];
                }
                run += 1 + cb[j];
            }
            c = gw;
            f = 1 + fw;
            return 16 * b + w;
        }
        u64 acc = 0;
        u32 j = 0;
        for (; j + 1 < P->K; ++j) {
            const u64 v = rd(rb + j);
            if (acc + v > cmax) break;
            acc += v;
        }
        c = acc;
        f = rd(rb + j);
        return j;
    }
    __device__ __forceinline__ void update(u32 s, u64 f_before) {
        if (P->kind == SCL_MODEL_FIXED) return;
        if (!LDS16 && P->fenwick) {  // order-k, probability_models.py:143-160
            const u64 rb = row_base();
            cnt[rb + 16 + s] = (u32)f_before;        // extra[s] = count - 1 = f_before after the increment
            atomicAdd(&cnt[rb + (s >> 4)], 1u);        // block total (no return value: fire and forget)
            if (P->k > 0) ctx = (ctx * P->K + s) % P->ctx_mod;
            if (f_before + 1 >= P->max_total) bad = 1;
            return;
        }
        if (P->kind == SCL_MODEL_IID) {  // probability_models.py:83-92
            wr(s, rd(s) + 1);
            u64 tot = 0;
            for (u32 j = 0; j < P->K; ++j) tot += rd(j);
            if (tot >= P->max_total)
                for (u32 j = 0; j < P->K; ++j) {
                    const u32 h = rd(j) >> 1;
                    wr(j, h > 1 ? h : 1);
                }
            return;
        }
        // order-k, probability_models.py:143-160
        const u64 cell = ctx * P->K + s;
        const u32 v = rd(cell) + 1;
        wr(cell, v);
        if (P->k > 0) ctx = (ctx * P->K + s) % P->ctx_mod;  // past_k[1:] + [s]
        if ((u64)v >= P->max_total) bad = 1;  // np.max(scalar, 1) -> AxisError in the reference
    }
};

// ---- PRECISION above 32 (round 3) ---------------------------------------------------------------------------
// low / high reach 2^PRECISION and range * count 2^(2 PRECISION - 2): the any-parameter kernels run them in 128 bits
// (WIDE = true) for PRECISION 33..62; the reference uses Python integers for low / high (but numpy int64 for the
// counts: its own products wrap once PRECISION + bit_length(total) exceeds 63 -- there these kernels compute the exact
// value, the reference does not).  Totals are below 2^32 here (the cells are u32), which the division relies on.
typedef unsigned __int128 u128;
template <bool WIDE>
struct AecWord {
    typedef u64 T;
};
template <>
struct AecWord<true> {
    typedef u128 T;
};
// floor(rng * c / T)
__device__ __forceinline__ u64 aec_muldiv(u64 rng, u64 c, u64 T) { return (rng * c) / T; }
__device__ __forceinline__ u128 aec_muldiv(u128 rng, u64 c, u64 T) {  // T < 2^32: four base-2^32 digits
    const u128 n = rng * c;
    u64 rem = 0;
    u128 q = 0;
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        const u64 cur = (rem << 32) | (u64)((u32)(n >> (32 * d)));
        const u64 qd = cur / T;
        rem = cur - qd * T;
        q = (q << 32) | qd;
    }
    return q;
}
// floor(((state - low + 1) * T - 1) / rng): the decoder's search bound (see the file header); the quotient is < T
__device__ __forceinline__ u64 aec_cmax(u64 state, u64 low, u64 T, u64 rng) { return ((state - low + 1) * T - 1) / rng; }
__device__ __forceinline__ u64 aec_cmax(u128 state, u128 low, u64 T, u128 rng) {
    const u128 num = (state - low + 1) * T - 1;  // < 2^62 * 2^32
    const double nd = (double)(u64)(num >> 64) * 18446744073709551616.0 + (double)(u64)num;
    const double rd = (double)(u64)(rng >> 64) * 18446744073709551616.0 + (double)(u64)rng;
    u64 q = (u64)(nd / rd);  // within a few units of the quotient (< 2^32, relative error 2^-50)
    while ((u128)q * rng > num) --q;
    while ((u128)(q + 1) * rng <= num) ++q;
    return q;
}

// SYM = u8: alphabets up to 256 (static tables staged in LDS).  SYM = u16 (the *_u16 entry points): alphabets up to 65536,
// static tables read where they are in device memory, adaptive rows scanned linearly; strides count SYMBOLS in both.
template <bool LDS16, bool WIDE = false, typename SYM = u8>
__global__ void __launch_bounds__(256) aec_encode_kernel(AecDev P, const SYM *__restrict__ sym, u64 sym_stride,
                                                        const u32 *__restrict__ lens, u32 chunk_len, u64 n_chunks,
                                                        u8 *__restrict__ out, u64 out_stride,
                                                        u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits,
                                                        u32 *__restrict__ status, u32 *__restrict__ scratch,
                                                        u64 *__restrict__ ctx_state) {
    __shared__ u32 s_f_lds[sizeof(SYM) == 1 ? 256 : 1];
    __shared__ u32 s_c_lds[sizeof(SYM) == 1 ? 256 : 1];
    __shared__ u16 s_cnt[LDS16 ? AEC_LDS_CELLS * 256 : 2];
    const u32 *s_f = P.d_freq, *s_c = P.d_cum;
    if constexpr (sizeof(SYM) == 1) {
        if (P.kind == SCL_MODEL_FIXED) {
            scl_load_table(s_f_lds, P.d_freq, P.K);
            scl_load_table(s_c_lds, P.d_cum, P.K);
        }
        __syncthreads();
        s_f = s_f_lds;
        s_c = s_c_lds;
    }
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 n = lens ? lens[c] : chunk_len;
    const SYM *src = sym + c * sym_stride;
    typedef typename AecWord<WIDE>::T W;
    const W FULL = (W)1 << P.P, HALF = FULL >> 1, QTR = FULL >> 2;
    LaneModel<LDS16> mdl;
    mdl.init(&P, scratch, c, s_cnt + threadIdx.x, ctx_state);
    FwdBitWriter w;
    w.init(out + c * out_stride, out_stride);
    u32 st = 0;
    if (P.size_bits < 32 && (n >> P.size_bits)) st |= SCL_ST_SIZE;
    w.put(n, P.size_bits);
    W low = 0, high = FULL;
    u64 pending = 0;
    for (u32 i = 0; i < n; ++i) {
        u32 s = src[i];
        if (s >= P.K) {
            st |= SCL_ST_SYMBOL;
            s = 0;
        }
        u64 cs, fs, T;
        mdl.lookup(s, s_f, s_c, cs, fs, T);
        if (T >= QTR || (WIDE && (T >> 32))) {  // assert total_freq < MAX_ALLOWED_TOTAL_FREQ, arithmetic_coding.py:110-112
            st |= SCL_ST_TOTAL;
            break;
        }
        const W rng = high - low;  // shrink_range :70-77
        high = low + aec_muldiv(rng, cs + fs, T);
        low = low + aec_muldiv(rng, cs, T);
        mdl.update(s, fs);  // :118
        while (high < HALF || low > HALF) {  // E1 / E2, :126-143
            if (high < HALF) {
                w.put(0, 1);
                w.put_run(1, pending);
                low <<= 1;
                high <<= 1;
            } else {
                w.put(1, 1);
                w.put_run(0, pending);
                low = (low - HALF) << 1;
                high = (high - HALF) << 1;
            }
            pending = 0;
        }
        while (low > QTR && high < 3 * QTR) {  // E3, :146-150
            pending += 1;
            low = (low - QTR) << 1;
            high = (high - QTR) << 1;
        }
    }
    pending += 1;  // termination, :153-159
    if (low <= QTR) {
        w.put(0, 1);
        w.put_run(1, pending);
    } else {
        w.put(1, 1);
        w.put_run(0, pending);
    }
    if (mdl.bad) st |= SCL_ST_TOTAL;
    if (ctx_state) ctx_state[c] = mdl.ctx;
    const u64 total = w.finish();
    if (w.overflow) st |= SCL_ST_CAPACITY;
    out_bit_off[c] = c * out_stride * 8;
    out_nbits[c] = (u32)total;
    if (status) status[c] = st;
}

template <bool LDS16, bool WIDE = false, typename SYM = u8>
__global__ void __launch_bounds__(256) aec_decode_kernel(AecDev P, const u8 *__restrict__ in, u64 in_size_bytes,
                                                        const u64 *__restrict__ bit_off,
                                                        const u32 *__restrict__ in_nbits, u64 n_chunks,
                                                        SYM *__restrict__ out_sym, u64 out_stride, u32 out_cap,
                                                        u32 *__restrict__ out_lens, u32 *__restrict__ consumed,
                                                        u32 *__restrict__ status, u32 *__restrict__ scratch,
                                                        u64 *__restrict__ ctx_state) {
    __shared__ u32 s_f_lds[sizeof(SYM) == 1 ? 256 : 1];
    __shared__ u32 s_c_lds[sizeof(SYM) == 1 ? 256 : 1];
    __shared__ u16 s_cnt[LDS16 ? AEC_LDS_CELLS * 256 : 2];
    const u32 *s_f = P.d_freq, *s_c = P.d_cum;
    if constexpr (sizeof(SYM) == 1) {
        if (P.kind == SCL_MODEL_FIXED) {
            scl_load_table(s_f_lds, P.d_freq, P.K);
            scl_load_table(s_c_lds, P.d_cum, P.K);
        }
        __syncthreads();
        s_f = s_f_lds;
        s_c = s_c_lds;
    }
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    typedef typename AecWord<WIDE>::T W;
    const W FULL = (W)1 << P.P, HALF = FULL >> 1, QTR = FULL >> 2;
    BitReader r;
    r.init(in, in_size_bytes, bit_off[c], in_nbits[c]);
    u32 st = 0;
    u32 n = r.get(P.size_bits);
    if (r.truncated) {
        st |= SCL_ST_TRUNCATED;
        n = 0;
    }
    out_lens[c] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    if (n == 0) {  // quirk Q5: defined here as header + the encoder's two termination bits
        consumed[c] = (st == 0) ? P.size_bits + 2 : 0;
        if (status) status[c] = st;
        return;
    }
    LaneModel<LDS16> mdl;
    mdl.init(&P, scratch, c, s_cnt + threadIdx.x, ctx_state);
    SYM *dst = out_sym + c * out_stride;
    // bit positions relative to the first bit after the header; bits past the end read as 0 (:258-261)
    const u64 body = r.pos;
    u64 used = P.P;  // the state register is always filled with PRECISION bits (:222-229)
    W state = P.P > 32 ? ((W)r.peek_at(body, P.P - 32) << 32) | r.peek_at(body + P.P - 32, 32) : (W)r.peek_at(body, P.P);
    W low = 0, high = FULL;
    u32 ndec = 0;
    for (;;) {
        const u64 T = mdl.total(s_f);
        if (T >= QTR || (WIDE && (T >> 32))) {
            st |= SCL_ST_TOTAL;
            break;
        }
        const W rng = high - low;
        const u64 cmax = aec_cmax(state, low, T, rng);  // see file header
        u64 cs, fs;
        const u32 s = mdl.search(cmax, s_f, s_c, cs, fs);
        high = low + aec_muldiv(rng, cs + fs, T);
        low = low + aec_muldiv(rng, cs, T);
        dst[ndec++] = (SYM)s;
        mdl.update(s, fs);
        if (ndec == n) break;  // before the renormalisation, :242-243
        while (high < HALF || low > HALF) {
            if (high < HALF) {
                low <<= 1;
                high <<= 1;
                state <<= 1;
            } else {
                low = (low - HALF) << 1;
                high = (high - HALF) << 1;
                state = (state - HALF) << 1;
            }
            state += r.peek_at(body + used, 1);
            used++;
        }
        while (low > QTR && high < 3 * QTR) {
            low = (low - QTR) << 1;
            high = (high - QTR) << 1;
            state = (state - QTR) << 1;
            state += r.peek_at(body + used, 1);
            used++;
        }
    }
    // how many of the last PRECISION bits belonged to the encoder (:277-282)
    u32 e = 0;
    for (; e < P.P; ++e) {
        const W slo = (state >> e) << e, shi = slo + ((W)1 << e);
        if (slo < low || shi > high) break;
    }
    if (e == P.P) e = P.P - 1;  // Python's loop variable after an unbroken range(PRECISION)
    if (mdl.bad) st |= SCL_ST_TOTAL;
    if (ctx_state) ctx_state[c] = mdl.ctx;
    consumed[c] = (u32)((i64)(used + P.size_bits) - ((i64)e - 1));
    if (status) status[c] = st;
}

// ---- host API -------------------------------------------------------------------------------------------
extern "C" int scl_aec_model_create(int model_kind, const uint32_t *h_freq_init, uint32_t K, uint32_t order_k,
                                    uint64_t max_total, uint32_t precision, uint32_t size_bits, scl_aec_model **out) {
    SCL_REQUIRE(out, "aec_model_create: null output");
    *out = nullptr;
    SCL_REQUIRE(model_kind == SCL_MODEL_FIXED || model_kind == SCL_MODEL_IID || model_kind == SCL_MODEL_ORDERK,
                "aec_model_create: unknown model kind %d", model_kind);
    SCL_REQUIRE(K >= 1 && K <= SCL_MAX_ALPHABET, "aec_model_create: alphabet size %u outside 1..65536", K);
    SCL_REQUIRE(precision >= 8 && precision <= 62, "aec_model_create: PRECISION %u outside 8..62", precision);
    SCL_REQUIRE(size_bits >= 1 && size_bits <= 32, "aec_model_create: DATA_BLOCK_SIZE_BITS %u outside 1..32", size_bits);
    SCL_REQUIRE(max_total >= 2, "aec_model_create: max_allowed_total_freq too small");
    scl_aec_model *m = new scl_aec_model();
    m->device = scl_current_device();
    m->dev.kind = model_kind;
    m->dev.K = K;
    m->dev.k = 0;
    m->dev.P = precision;
    m->dev.size_bits = size_bits;
    m->dev.max_total = max_total;
    m->dev.ctx_mod = 1;
    m->dev.cells = 0;
    m->dev.fenwick = 0;
    m->dev.row_cells = 0;
    std::vector<u32> freq_v(K), cum_v(K);
    u32 *freq = freq_v.data(), *cum = cum_v.data();
    u64 tot = 0;
    if (model_kind == SCL_MODEL_ORDERK) {
        if (order_k > 3) {
            delete m;
            scl_set_error("aec_model_create: order k = %u > 3 not supported", order_k);
            return SCL_E_PARAM;
        }
        m->dev.k = order_k;
        u64 cells = K;
        for (u32 i = 0; i < order_k; ++i) {
            m->dev.ctx_mod *= K;
            cells *= K;
        }
        if (cells > (1ull << 26)) {
            delete m;
            scl_set_error("aec_model_create: K^(k+1) = %llu cells per chunk is too large", (unsigned long long)cells);
            return SCL_E_PARAM;
        }
        m->dev.cells = cells;
        if (K <= 256 && cells > AEC_LDS_CELLS) {  // two-level rows: 16 block totals + counts in blocks of 16
            m->dev.fenwick = 1;
            m->dev.row_cells = 16 + 16 * ((K + 15) / 16);
            m->dev.cells = m->dev.ctx_mod * m->dev.row_cells;
        }
        for (u32 i = 0; i < K; ++i) {
            freq[i] = 1;
            cum[i] = i;
        }
        tot = K;
    } else {
        if (!h_freq_init) {
            delete m;
            scl_set_error("aec_model_create: initial frequencies required");
            return SCL_E_PARAM;
        }
        for (u32 i = 0; i < K; ++i) {
            if (h_freq_init[i] == 0 || tot + h_freq_init[i] >= (1ull << 32)) {
                delete m;
                scl_set_error("aec_model_create: zero or too large initial frequency at symbol %u", i);
                return SCL_E_PARAM;
            }
            freq[i] = h_freq_init[i];
            cum[i] = (u32)tot;
            tot += h_freq_init[i];
        }
        m->dev.cells = (model_kind == SCL_MODEL_IID) ? K : 0;
    }
    m->dev.total0 = (u32)tot;
    ::memcpy(m->h_freq, freq, (K < 256 ? K : 256) * sizeof(u32));
    const u64 tab_entries = K > 256 ? K : 256;
    hipError_t e = hipMalloc((void **)&m->d_freq, tab_entries * sizeof(u32));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_cum, tab_entries * sizeof(u32));
    if (e == hipSuccess) e = hipMemcpy(m->d_freq, freq, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_cum, cum, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess && model_kind == SCL_MODEL_IID && K > 16 && K <= 256) {
        u32 init[136];
        aec_iid_build_init(freq, K, init);
        e = hipMalloc((void **)&m->d_iid_init, sizeof(init));
        if (e == hipSuccess) e = hipMemcpy(m->d_iid_init, init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        scl_set_error("aec_model_create: device table upload failed: %s", hipGetErrorString(e));
        scl_aec_model_destroy(m);
        return SCL_E_HIP;
    }
    m->dev.d_freq = m->d_freq;
    m->dev.d_cum = m->d_cum;
    *out = m;
    return SCL_OK;
}

extern "C" void scl_aec_model_destroy(scl_aec_model *m) {
    if (!m) return;
    if (m->d_freq) (void)hipFree(m->d_freq);
    if (m->d_cum) (void)hipFree(m->d_cum);
    if (m->d_iid_init) (void)hipFree(m->d_iid_init);
    delete m;
}

extern "C" uint64_t scl_aec_slot_bytes(const scl_aec_model *m, uint64_t n_symbols) {
    if (!m) return 0;
    // every symbol narrows the interval by at most a factor 1/QTR-ish: <= PRECISION bits per symbol, plus
    // header, termination and pending bits
    const u64 bits = (u64)m->dev.size_bits + n_symbols * (u64)m->dev.P + 2 * m->dev.P + 8;
    return scl_round_up((bits + 7) / 8 + 4, 128);
}

extern "C" uint64_t scl_aec_scratch_bytes(const scl_aec_model *m, uint64_t n_chunks) {
    if (!m) return 0;
    return scl_round_up(m->dev.cells * n_chunks * sizeof(u32), 256);
}

extern "C" int scl_aec_fast_path(const scl_aec_model *m, uint64_t max_symbols) {
    return (m && (aec_fast_ok(m, max_symbols) || aec_static_ok(m) || aec_iid_ok(m, max_symbols) ||
                  aec_wide_ok(m, max_symbols))) ? 1 : 0;
}

// per-lane context tables in LDS: at most 256 cells, counts (initial + one per symbol) must fit 16 bits
static bool aec_use_lds(const scl_aec_model *m, u64 max_symbols) {
    if (m->dev.kind == SCL_MODEL_FIXED || m->dev.cells == 0 || m->dev.cells > AEC_LDS_CELLS) return false;
    u64 max_init = 1;
    if (m->dev.kind == SCL_MODEL_IID) max_init = m->dev.total0;  // bound on any single initial count
    return max_init + max_symbols < 65535 && m->dev.max_total > max_init + max_symbols;
}

// SCL_AEC_WIDE=dense in the environment keeps the two-level-row kernels of scl_aec_wide.hip in charge of order-k models on
// large alphabets (tests run both; the default is scl_aec_sparse.hip)
static bool aec_wide_dense_forced() {
    const char *e = getenv("SCL_AEC_WIDE");
    return e && e[0] == 'd';
}

// zero_bytes != 0: the kernels about to run use (and need zero-filled) only that much of it (scl_aec_wide.hip: u16 cells)
static int aec_prepare_scratch(const scl_aec_model *m, u64 n_chunks, void *d_scratch, u64 scratch_bytes,
                               hipStream_t st, u64 zero_bytes = 0) {
    const u64 need = m->dev.cells * n_chunks * sizeof(u32);
    SCL_REQUIRE(need == 0 || (d_scratch && scratch_bytes >= need), "aec: scratch of %llu bytes required, got %llu",
                (unsigned long long)need, (unsigned long long)scratch_bytes);
    if (m->dev.kind == SCL_MODEL_ORDERK) SCL_HIP_TRY(hipMemsetAsync(d_scratch, 0, zero_bytes ? zero_bytes : need, st));
    return SCL_OK;
}

extern "C" int scl_aec_encode_batch(const scl_aec_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                    const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks, uint8_t *d_out,
                                    uint64_t out_stride, uint64_t *d_out_bit_offset, uint32_t *d_out_nbits,
                                    uint32_t *d_status, void *d_scratch, uint64_t scratch_bytes, void *stream) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits, "aec_encode_batch: null pointer argument");
    SCL_REQUIRE(m->dev.K <= 256, "aec_encode_batch: alphabet of %u symbols: use scl_aec_encode_batch_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "aec_encode_batch")) return rc_dev;
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride * 8 < (1ull << 32),
                "aec_encode_batch: bad out_stride %llu", (unsigned long long)out_stride);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0, "aec_encode_batch: d_out must be 16-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool tuned = !scl_force_generic();
    RowRelay relay;  // rows that do not start on 16-byte boundaries are re-laid for the tuned kernels
    if (tuned && (aec_fast_ok(m, chunk_len) || aec_iid_ok(m, chunk_len) || aec_static_ok(m) || aec_wide_ok(m, chunk_len)) &&
        out_stride >= scl_aec_slot_bytes(m, chunk_len))
        if (int rc_r = relay.in(d_sym, sym_stride, chunk_len, n_chunks, st)) return rc_r;
    // small-alphabet adaptive models: cumulative context rows in LDS, closed-form renormalisation (scl_aec_fast.hip)
    if (tuned && aec_fast_ok(m, chunk_len) && ((uintptr_t)d_sym & 15) == 0 && (sym_stride & 15) == 0 &&
        sym_stride >= scl_round_up(chunk_len, 16) && (out_stride & 63) == 0 &&
        out_stride >= scl_aec_slot_bytes(m, chunk_len)) {
        aec_fast_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                               d_out_nbits, d_status, st);
        SCL_HIP_TRY(hipGetLastError());
        return SCL_OK;
    }
    // adaptive i.i.d. model on a large alphabet: two-level cumulative table per lane in LDS (scl_aec_iid.hip)
    if (tuned && aec_iid_ok(m, chunk_len) && ((uintptr_t)d_sym & 3) == 0 && (sym_stride & 3) == 0 &&
        sym_stride >= scl_round_up(chunk_len, 4) && out_stride >= scl_aec_slot_bytes(m, chunk_len)) {
        aec_iid_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                              d_out_nbits, d_status, st);
        SCL_HIP_TRY(hipGetLastError());
        return SCL_OK;
    }
    // static model: shared table in LDS, line-granular I/O (scl_aec_static.hip)
    if (tuned && aec_static_ok(m) && ((uintptr_t)d_sym & 15) == 0 && (sym_stride & 15) == 0 && (out_stride & 15) == 0 &&
        out_stride >= scl_aec_slot_bytes(m, chunk_len)) {
        aec_static_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride,
                                 d_out_bit_offset, d_out_nbits, d_status, st);
        SCL_HIP_TRY(hipGetLastError());
        return SCL_OK;
    }
    // order-k on a large alphabet: the same two-level rows in device memory, tuned arithmetic, lookups issued ahead (scl_aec_wide.hip)
    const bool wide = tuned && aec_wide_ok(m, chunk_len) && ((uintptr_t)d_sym & 3) == 0 && (sym_stride & 3) == 0 &&
                      sym_stride >= scl_round_up(chunk_len, 4) && out_stride >= scl_aec_slot_bytes(m, chunk_len);
    if (!aec_use_lds(m, chunk_len)) {
        int rc = aec_prepare_scratch(m, n_chunks, d_scratch, scratch_bytes, st,
                                     wide ? (aec_wide_dense_forced() ? aec_wide_scratch_bytes(m, n_chunks)
                                                                     : aec_sparse_zero_bytes(m, n_chunks))
                                          : 0);
        if (rc) return rc;
    }
    if (wide && !aec_wide_dense_forced()) {  // one table line per symbol while a context is young (scl_aec_sparse.hip)
        aec_sparse_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                                 d_out_nbits, d_status, (u32 *)d_scratch, st);
        SCL_HIP_TRY(hipGetLastError());
        return SCL_OK;
    }
    if (wide) {
        aec_wide_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                               d_out_nbits, d_status, (u32 *)d_scratch, st);
        SCL_HIP_TRY(hipGetLastError());
        return SCL_OK;
    }
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    if (aec_use_lds(m, chunk_len))
        do {
        if (m->dev.P > 32)
            hipLaunchKernelGGL((aec_encode_kernel<true, true>), dim3(blocks), dim3(threads), 0, st, m->dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status,
                           (u32 *)d_scratch, (u64 *)nullptr);
        else
            hipLaunchKernelGGL((aec_encode_kernel<true, false>), dim3(blocks), dim3(threads), 0, st, m->dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status,
                           (u32 *)d_scratch, (u64 *)nullptr);
    } while (0);
    else
        do {
        if (m->dev.P > 32)
            hipLaunchKernelGGL((aec_encode_kernel<false, true>), dim3(blocks), dim3(threads), 0, st, m->dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status,
                           (u32 *)d_scratch, (u64 *)nullptr);
        else
            hipLaunchKernelGGL((aec_encode_kernel<false, false>), dim3(blocks), dim3(threads), 0, st, m->dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status,
                           (u32 *)d_scratch, (u64 *)nullptr);
    } while (0);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_aec_decode_batch(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                    const uint64_t *d_bit_offset, const uint32_t *d_in_nbits, uint64_t n_chunks,
                                    uint8_t *d_out_sym, uint64_t out_stride, uint32_t out_cap, uint32_t *d_out_lens,
                                    uint32_t *d_consumed, uint32_t *d_status, void *d_scratch, uint64_t scratch_bytes,
                                    void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "aec_decode_batch: null pointer argument");
    SCL_REQUIRE(m->dev.K <= 256, "aec_decode_batch: alphabet of %u symbols: use scl_aec_decode_batch_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "aec_decode_batch")) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0, "aec_decode_batch: d_in must be 4-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool tuned = !scl_force_generic() && in_size_bytes >= 4;  // the tuned readers load whole 32-bit words
    RowRelay relay;  // output rows the tuned kernels cannot store to go through aligned scratch and are copied back
    if (tuned && (aec_fast_ok(m, out_cap) || aec_iid_ok(m, out_cap) || aec_static_ok(m) || aec_wide_ok(m, out_cap)) &&
        ((uintptr_t)d_in & 15) == 0)
        if (int rc_r = relay.out_begin(d_out_sym, out_stride, out_cap, n_chunks, st)) return rc_r;
    if (tuned && aec_fast_ok(m, out_cap) && ((uintptr_t)d_in & 15) == 0 && ((uintptr_t)d_out_sym & 15) == 0 &&
        (out_stride & 15) == 0 && out_stride >= scl_round_up(out_cap, 16)) {
        aec_fast_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                               out_cap, d_out_lens, d_consumed, d_status, st);
        SCL_HIP_TRY(hipGetLastError());
        return relay.out_end(d_out_lens);
    }
    if (tuned && aec_iid_ok(m, out_cap) && ((uintptr_t)d_out_sym & 3) == 0 && (out_stride & 3) == 0 &&
        out_stride >= scl_round_up(out_cap, 4)) {
        aec_iid_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                              out_cap, d_out_lens, d_consumed, d_status, st);
        SCL_HIP_TRY(hipGetLastError());
        return relay.out_end(d_out_lens);
    }
    if (tuned && aec_static_ok(m) && ((uintptr_t)d_in & 15) == 0 && in_size_bytes < (1ull << 34) &&
        ((uintptr_t)d_out_sym & 15) == 0 && (out_stride & 15) == 0 && out_stride >= out_cap) {
        aec_static_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                                 out_cap, d_out_lens, d_consumed, d_status, st);
        SCL_HIP_TRY(hipGetLastError());
        return relay.out_end(d_out_lens);
    }
    const bool wide = tuned && aec_wide_ok(m, out_cap) && ((uintptr_t)d_out_sym & 3) == 0 && (out_stride & 3) == 0 &&
                      out_stride >= scl_round_up(out_cap, 4);
    if (!aec_use_lds(m, out_cap)) {
        int rc = aec_prepare_scratch(m, n_chunks, d_scratch, scratch_bytes, st,
                                     wide ? (aec_wide_dense_forced() ? aec_wide_scratch_bytes(m, n_chunks)
                                                                     : aec_sparse_zero_bytes(m, n_chunks))
                                          : 0);
        if (rc) return rc;
    }
    if (wide && !aec_wide_dense_forced()) {
        aec_sparse_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                                 d_out_lens, d_consumed, d_status, (u32 *)d_scratch, st);
        SCL_HIP_TRY(hipGetLastError());
        return relay.out_end(d_out_lens);
    }
    if (wide) {
        aec_wide_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                               d_out_lens, d_consumed, d_status, (u32 *)d_scratch, st);
        SCL_HIP_TRY(hipGetLastError());
        return relay.out_end(d_out_lens);
    }
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    if (aec_use_lds(m, out_cap))
        do {
        if (m->dev.P > 32)
            hipLaunchKernelGGL((aec_decode_kernel<true, true>), dim3(blocks), dim3(threads), 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                           d_consumed, d_status, (u32 *)d_scratch, (u64 *)nullptr);
        else
            hipLaunchKernelGGL((aec_decode_kernel<true, false>), dim3(blocks), dim3(threads), 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                           d_consumed, d_status, (u32 *)d_scratch, (u64 *)nullptr);
    } while (0);
    else
        do {
        if (m->dev.P > 32)
            hipLaunchKernelGGL((aec_decode_kernel<false, true>), dim3(blocks), dim3(threads), 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                           d_consumed, d_status, (u32 *)d_scratch, (u64 *)nullptr);
        else
            hipLaunchKernelGGL((aec_decode_kernel<false, false>), dim3(blocks), dim3(threads), 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                           d_consumed, d_status, (u32 *)d_scratch, (u64 *)nullptr);
    } while (0);
    SCL_HIP_TRY(hipGetLastError());
    return relay.out_end(d_out_lens);
}

// ---- coder state carried across blocks (quirk Q4) -----------------------------------------------------------
// The reference's ArithmeticEncoder / ArithmeticDecoder own a freq_model object and never reset it
// (arithmetic_coding.py:52-56,118): block i+1 of DataEncoder.encode (core/data_encoder_decoder.py:57-69) is coded
// with the counts AND the order-k context block i left behind.  Device state of n coders:
//   [cells * n u32: the same private model regions the batch kernels use as scratch][pad to 256 B][u64 ctx[n]]
// scl_aec_*_batch_resume run the any-parameter kernels on it without re-initialising anything.
static u64 aec_state_cells_bytes(const scl_aec_model *m, u64 n_coders) {
    return scl_round_up(m->dev.cells * n_coders * sizeof(u32), 256);
}

extern "C" uint64_t scl_aec_state_bytes(const scl_aec_model *m, uint64_t n_coders) {
    if (!m) return 0;
    return aec_state_cells_bytes(m, n_coders) + scl_round_up(n_coders * sizeof(u64), 256);
}

extern "C" uint64_t scl_aec_state_counts(const scl_aec_model *m) {
    if (!m || m->dev.kind == SCL_MODEL_FIXED) return 0;
    return (m->dev.kind == SCL_MODEL_ORDERK) ? m->dev.ctx_mod * m->dev.K : m->dev.K;
}

__global__ void aec_state_fill_iid(u32 *__restrict__ cnt, const u32 *__restrict__ freq, u32 K, u64 total) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) cnt[i] = freq[i % K];
}

extern "C" int scl_aec_state_reset(const scl_aec_model *m, void *d_state, uint64_t state_bytes, uint64_t n_coders,
                                   void *stream) {
    SCL_REQUIRE(m && d_state, "aec_state_reset: null pointer argument");
    SCL_REQUIRE(state_bytes >= scl_aec_state_bytes(m, n_coders), "aec_state_reset: state of %llu bytes required",
                (unsigned long long)scl_aec_state_bytes(m, n_coders));
    hipStream_t st = (hipStream_t)stream;
    SCL_HIP_TRY(hipMemsetAsync(d_state, 0, scl_aec_state_bytes(m, n_coders), st));  // ORDERK: count - 1 = 0; ctx = 0
    if (m->dev.kind == SCL_MODEL_IID && n_coders) {
        const u64 total = (u64)m->dev.K * n_coders;
        hipLaunchKernelGGL(aec_state_fill_iid, dim3((u32)((total + 255) / 256)), dim3(256), 0, st, (u32 *)d_state,
                           m->dev.d_freq, m->dev.K, total);
        SCL_HIP_TRY(hipGetLastError());
    }
    return SCL_OK;
}

// canonical host form <-> device form of ONE coder's state.  h_counts: scl_aec_state_counts() actual counts
// (IID: [K]; ORDERK: [K^(k+1)] row-major, last axis = next symbol, = freqs_kplus1_tuple.ravel()); h_past_k: the
// last k symbol indices, oldest first (= past_k, probability_models.py:116).
static int aec_state_to_device(const scl_aec_model *m, const u32 *h_counts, const u32 *h_past_k, std::vector<u32> &cells,
                               u64 &ctx) {
    const AecDev &P = m->dev;
    cells.assign(P.cells, 0);
    ctx = 0;
    if (P.kind == SCL_MODEL_FIXED) return SCL_OK;
    const u64 n = scl_aec_state_counts(m);
    for (u64 i = 0; i < n; ++i) SCL_REQUIRE(h_counts[i] >= 1, "aec_state: count %llu is zero", (unsigned long long)i);
    if (P.kind == SCL_MODEL_IID) {
        for (u32 j = 0; j < P.K; ++j) cells[j] = h_counts[j];
        return SCL_OK;
    }
    for (u32 i = 0; i < P.k; ++i) {
        SCL_REQUIRE(h_past_k[i] < P.K, "aec_state: past symbol index %u outside the alphabet", h_past_k[i]);
        ctx = ctx * P.K + h_past_k[i];
    }
    for (u64 row = 0; row < P.ctx_mod; ++row)
        for (u32 s = 0; s < P.K; ++s) {
            const u32 extra = h_counts[row * P.K + s] - 1;
            if (P.fenwick) {
                cells[row * P.row_cells + 16 + s] = extra;
                cells[row * P.row_cells + (s >> 4)] += extra;
            } else {
                cells[row * P.K + s] = extra;
            }
        }
    return SCL_OK;
}

static void aec_state_from_device(const scl_aec_model *m, const std::vector<u32> &cells, u64 ctx, u32 *h_counts,
                                  u32 *h_past_k) {
    const AecDev &P = m->dev;
    if (P.kind == SCL_MODEL_FIXED) return;
    if (P.kind == SCL_MODEL_IID) {
        for (u32 j = 0; j < P.K; ++j) h_counts[j] = cells[j];
        return;
    }
    for (u64 row = 0; row < P.ctx_mod; ++row)
        for (u32 s = 0; s < P.K; ++s)
            h_counts[row * P.K + s] = 1 + (P.fenwick ? cells[row * P.row_cells + 16 + s] : cells[row * P.K + s]);
    for (u32 i = P.k; i-- > 0;) {
        h_past_k[i] = (u32)(ctx % P.K);
        ctx /= P.K;
    }
}

extern "C" int scl_aec_state_upload(const scl_aec_model *m, void *d_state, uint64_t n_coders, uint64_t coder,
                                    const uint32_t *h_counts, const uint32_t *h_past_k, void *stream) {
    SCL_REQUIRE(m && d_state && coder < n_coders, "aec_state_upload: bad arguments");
    SCL_REQUIRE(m->dev.kind == SCL_MODEL_FIXED || (h_counts && (m->dev.k == 0 || h_past_k)),
                "aec_state_upload: null state arrays");
    std::vector<u32> cells;
    u64 ctx;
    int rc = aec_state_to_device(m, h_counts, h_past_k, cells, ctx);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    u8 *base = (u8 *)d_state;
    if (!cells.empty())
        SCL_HIP_TRY(hipMemcpyAsync(base + coder * m->dev.cells * sizeof(u32), cells.data(), cells.size() * sizeof(u32),
                                   hipMemcpyHostToDevice, st));
    SCL_HIP_TRY(hipMemcpyAsync(base + aec_state_cells_bytes(m, n_coders) + coder * sizeof(u64), &ctx, sizeof(u64),
                               hipMemcpyHostToDevice, st));
    SCL_HIP_TRY(hipStreamSynchronize(st));  // the staging vector dies with this call
    return SCL_OK;
}

extern "C" int scl_aec_state_download(const scl_aec_model *m, const void *d_state, uint64_t n_coders, uint64_t coder,
                                      uint32_t *h_counts, uint32_t *h_past_k, void *stream) {
    SCL_REQUIRE(m && d_state && coder < n_coders, "aec_state_download: bad arguments");
    SCL_REQUIRE(m->dev.kind == SCL_MODEL_FIXED || (h_counts && (m->dev.k == 0 || h_past_k)),
                "aec_state_download: null state arrays");
    std::vector<u32> cells(m->dev.cells);
    u64 ctx = 0;
    hipStream_t st = (hipStream_t)stream;
    const u8 *base = (const u8 *)d_state;
    if (!cells.empty())
        SCL_HIP_TRY(hipMemcpyAsync(cells.data(), base + coder * m->dev.cells * sizeof(u32), cells.size() * sizeof(u32),
                                   hipMemcpyDeviceToHost, st));
    SCL_HIP_TRY(hipMemcpyAsync(&ctx, base + aec_state_cells_bytes(m, n_coders) + coder * sizeof(u64), sizeof(u64),
                               hipMemcpyDeviceToHost, st));
    SCL_HIP_TRY(hipStreamSynchronize(st));
    aec_state_from_device(m, cells, ctx, h_counts, h_past_k);
    return SCL_OK;
}

extern "C" int scl_aec_encode_batch_resume(const scl_aec_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                           const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                           uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                           uint32_t *d_out_nbits, uint32_t *d_status, void *d_state,
                                           uint64_t state_bytes, uint64_t n_coders, void *stream) {
    SCL_REQUIRE(m, "aec_encode_batch_resume: null model");
    SCL_REQUIRE(m->dev.K <= 256, "aec_encode_batch_resume: alphabet of %u symbols: use scl_aec_encode_batch_resume_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "aec_encode_batch_resume")) return rc_dev;
    if (m->dev.kind == SCL_MODEL_FIXED)  // nothing to carry
        return scl_aec_encode_batch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride,
                                    d_out_bit_offset, d_out_nbits, d_status, nullptr, 0, stream);
    SCL_REQUIRE(d_sym && d_out && d_out_bit_offset && d_out_nbits && d_state,
                "aec_encode_batch_resume: null pointer argument");
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride * 8 < (1ull << 32),
                "aec_encode_batch_resume: bad out_stride %llu", (unsigned long long)out_stride);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_state & 255) == 0,
                "aec_encode_batch_resume: d_out must be 16-byte, d_state 256-byte aligned");
    // the layout of d_state is a function of the n_coders it was reset with (the context array follows the cells of
    // ALL coders): chunk c continues coder c, so a batch may be shorter than the state, never longer
    SCL_REQUIRE(n_chunks <= n_coders, "aec_encode_batch_resume: %llu chunks but the state holds %llu coders",
                (unsigned long long)n_chunks, (unsigned long long)n_coders);
    SCL_REQUIRE(state_bytes >= scl_aec_state_bytes(m, n_coders), "aec_encode_batch_resume: state of %llu bytes required",
                (unsigned long long)scl_aec_state_bytes(m, n_coders));
    if (n_chunks == 0) return SCL_OK;
    u64 *ctx_state = (u64 *)((u8 *)d_state + aec_state_cells_bytes(m, n_coders));
    do {
        if (m->dev.P > 32)
            hipLaunchKernelGGL((aec_encode_kernel<false, true>), dim3((u32)((n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       m->dev, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                       d_out_nbits, d_status, (u32 *)d_state, ctx_state);
        else
            hipLaunchKernelGGL((aec_encode_kernel<false, false>), dim3((u32)((n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       m->dev, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                       d_out_nbits, d_status, (u32 *)d_state, ctx_state);
    } while (0);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_aec_decode_batch_resume(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                           const uint64_t *d_bit_offset, const uint32_t *d_in_nbits, uint64_t n_chunks,
                                           uint8_t *d_out_sym, uint64_t out_stride, uint32_t out_cap,
                                           uint32_t *d_out_lens, uint32_t *d_consumed, uint32_t *d_status,
                                           void *d_state, uint64_t state_bytes, uint64_t n_coders, void *stream) {
    SCL_REQUIRE(m, "aec_decode_batch_resume: null model");
    SCL_REQUIRE(m->dev.K <= 256, "aec_decode_batch_resume: alphabet of %u symbols: use scl_aec_decode_batch_resume_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "aec_decode_batch_resume")) return rc_dev;
    if (m->dev.kind == SCL_MODEL_FIXED)
        return scl_aec_decode_batch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                                    out_cap, d_out_lens, d_consumed, d_status, nullptr, 0, stream);
    SCL_REQUIRE(d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed && d_state,
                "aec_decode_batch_resume: null pointer argument");
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0 && ((uintptr_t)d_state & 255) == 0,
                "aec_decode_batch_resume: d_in must be 4-byte, d_state 256-byte aligned");
    // the layout of d_state is a function of the n_coders it was reset with (the context array follows the cells of
    // ALL coders): chunk c continues coder c, so a batch may be shorter than the state, never longer
    SCL_REQUIRE(n_chunks <= n_coders, "aec_decode_batch_resume: %llu chunks but the state holds %llu coders",
                (unsigned long long)n_chunks, (unsigned long long)n_coders);
    SCL_REQUIRE(state_bytes >= scl_aec_state_bytes(m, n_coders), "aec_decode_batch_resume: state of %llu bytes required",
                (unsigned long long)scl_aec_state_bytes(m, n_coders));
    if (n_chunks == 0) return SCL_OK;
    u64 *ctx_state = (u64 *)((u8 *)d_state + aec_state_cells_bytes(m, n_coders));
    do {
        if (m->dev.P > 32)
            hipLaunchKernelGGL((aec_decode_kernel<false, true>), dim3((u32)((n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       m->dev, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                       d_out_lens, d_consumed, d_status, (u32 *)d_state, ctx_state);
        else
            hipLaunchKernelGGL((aec_decode_kernel<false, false>), dim3((u32)((n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       m->dev, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                       d_out_lens, d_consumed, d_status, (u32 *)d_state, ctx_state);
    } while (0);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}


// ---- uint16 symbol indices: alphabets up to 65536 (any model; the any-parameter kernels, counts in device memory) ----
// d_state != nullptr: the *_resume form (chunk c continues coder c of a state reset with n_coders coders)
static int aec_encode_u16(const char *what, const scl_aec_model *m, const u16 *d_sym, u64 sym_stride, const u32 *d_lens,
                          u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset,
                          u32 *d_out_nbits, u32 *d_status, void *d_scratch, u64 scratch_bytes, void *d_state,
                          u64 state_bytes, u64 n_coders, hipStream_t st) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits, "%s: null pointer argument", what);
    if (int rc_dev = scl_check_device(m->device, what)) return rc_dev;
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride * 8 < (1ull << 32), "%s: bad out_stride %llu", what,
                (unsigned long long)out_stride);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_sym & 1) == 0,
                "%s: d_out must be 16-byte aligned, d_sym 2-byte aligned", what);
    u32 *cells = (u32 *)d_scratch;
    u64 *ctx_state = nullptr;
    if (d_state && m->dev.kind != SCL_MODEL_FIXED) {
        SCL_REQUIRE(((uintptr_t)d_state & 255) == 0, "%s: d_state must be 256-byte aligned", what);
        SCL_REQUIRE(n_chunks <= n_coders, "%s: %llu chunks but the state holds %llu coders", what,
                    (unsigned long long)n_chunks, (unsigned long long)n_coders);
        SCL_REQUIRE(state_bytes >= scl_aec_state_bytes(m, n_coders), "%s: state of %llu bytes required", what,
                    (unsigned long long)scl_aec_state_bytes(m, n_coders));
        cells = (u32 *)d_state;
        ctx_state = (u64 *)((u8 *)d_state + aec_state_cells_bytes(m, n_coders));
    }
    if (n_chunks == 0) return SCL_OK;
    if (!ctx_state)
        if (int rc = aec_prepare_scratch(m, n_chunks, d_scratch, scratch_bytes, st)) return rc;
    const dim3 grid((u32)((n_chunks + 255) / 256)), block(256);
    if (m->dev.P > 32)
        hipLaunchKernelGGL((aec_encode_kernel<false, true, u16>), grid, block, 0, st, m->dev, d_sym, sym_stride, d_lens,
                           chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status, cells,
                           ctx_state);
    else
        hipLaunchKernelGGL((aec_encode_kernel<false, false, u16>), grid, block, 0, st, m->dev, d_sym, sym_stride, d_lens,
                           chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status, cells,
                           ctx_state);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

static int aec_decode_u16(const char *what, const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes,
                          const u64 *d_bit_offset, const u32 *d_in_nbits, u64 n_chunks, u16 *d_out_sym, u64 out_stride,
                          u32 out_cap, u32 *d_out_lens, u32 *d_consumed, u32 *d_status, void *d_scratch,
                          u64 scratch_bytes, void *d_state, u64 state_bytes, u64 n_coders, hipStream_t st) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "%s: null pointer argument", what);
    if (int rc_dev = scl_check_device(m->device, what)) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0 && ((uintptr_t)d_out_sym & 1) == 0,
                "%s: d_in must be 4-byte aligned, d_out_sym 2-byte aligned", what);
    u32 *cells = (u32 *)d_scratch;
    u64 *ctx_state = nullptr;
    if (d_state && m->dev.kind != SCL_MODEL_FIXED) {
        SCL_REQUIRE(((uintptr_t)d_state & 255) == 0, "%s: d_state must be 256-byte aligned", what);
        SCL_REQUIRE(n_chunks <= n_coders, "%s: %llu chunks but the state holds %llu coders", what,
                    (unsigned long long)n_chunks, (unsigned long long)n_coders);
        SCL_REQUIRE(state_bytes >= scl_aec_state_bytes(m, n_coders), "%s: state of %llu bytes required", what,
                    (unsigned long long)scl_aec_state_bytes(m, n_coders));
        cells = (u32 *)d_state;
        ctx_state = (u64 *)((u8 *)d_state + aec_state_cells_bytes(m, n_coders));
    }
    if (n_chunks == 0) return SCL_OK;
    if (!ctx_state)
        if (int rc = aec_prepare_scratch(m, n_chunks, d_scratch, scratch_bytes, st)) return rc;
    const dim3 grid((u32)((n_chunks + 255) / 256)), block(256);
    if (m->dev.P > 32)
        hipLaunchKernelGGL((aec_decode_kernel<false, true, u16>), grid, block, 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens, d_consumed,
                           d_status, cells, ctx_state);
    else
        hipLaunchKernelGGL((aec_decode_kernel<false, false, u16>), grid, block, 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens, d_consumed,
                           d_status, cells, ctx_state);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_aec_encode_batch_u16(const scl_aec_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                                        const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks, uint8_t *d_out,
                                        uint64_t out_stride, uint64_t *d_out_bit_offset, uint32_t *d_out_nbits,
                                        uint32_t *d_status, void *d_scratch, uint64_t scratch_bytes, void *stream) {
    return aec_encode_u16("aec_encode_batch_u16", m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride,
                          d_out_bit_offset, d_out_nbits, d_status, d_scratch, scratch_bytes, nullptr, 0, 0,
                          (hipStream_t)stream);
}

extern "C" int scl_aec_decode_batch_u16(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                        const uint64_t *d_bit_offset, const uint32_t *d_in_nbits, uint64_t n_chunks,
                                        uint16_t *d_out_sym, uint64_t out_stride, uint32_t out_cap,
                                        uint32_t *d_out_lens, uint32_t *d_consumed, uint32_t *d_status, void *d_scratch,
                                        uint64_t scratch_bytes, void *stream) {
    return aec_decode_u16("aec_decode_batch_u16", m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym,
                          out_stride, out_cap, d_out_lens, d_consumed, d_status, d_scratch, scratch_bytes, nullptr, 0, 0,
                          (hipStream_t)stream);
}

extern "C" int scl_aec_encode_batch_resume_u16(const scl_aec_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                                               const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                               uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                               uint32_t *d_out_nbits, uint32_t *d_status, void *d_state,
                                               uint64_t state_bytes, uint64_t n_coders, void *stream) {
    SCL_REQUIRE(m && (d_state || m->dev.kind == SCL_MODEL_FIXED), "aec_encode_batch_resume_u16: null model or state");
    return aec_encode_u16("aec_encode_batch_resume_u16", m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out,
                          out_stride, d_out_bit_offset, d_out_nbits, d_status, nullptr, 0, d_state, state_bytes, n_coders,
                          (hipStream_t)stream);
}

extern "C" int scl_aec_decode_batch_resume_u16(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                               const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                               uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                                               uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                               uint32_t *d_status, void *d_state, uint64_t state_bytes,
                                               uint64_t n_coders, void *stream) {
    SCL_REQUIRE(m && (d_state || m->dev.kind == SCL_MODEL_FIXED), "aec_decode_batch_resume_u16: null model or state");
    return aec_decode_u16("aec_decode_batch_resume_u16", m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks,
                          d_out_sym, out_stride, out_cap, d_out_lens, d_consumed, d_status, nullptr, 0, d_state,
                          state_bytes, n_coders, (hipStream_t)stream);
}

// ---- single-chunk host drivers --------------------------------------------------------------------------
static int aec_run_enc(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                       u32 *d_nbits, u32 *d_status, void *d_scratch, u64 scratch_bytes) {
    return scl_aec_encode_batch((const scl_aec_model *)model, d_sym, n, nullptr, n, 1, d_out, out_stride, d_bit_off,
                                d_nbits, d_status, d_scratch, scratch_bytes, nullptr);
}
static u64 aec_slot(const void *model, u64 n) { return scl_aec_slot_bytes((const scl_aec_model *)model, n); }
static u64 aec_scratch(const void *model) { return scl_aec_scratch_bytes((const scl_aec_model *)model, 1); }
static int aec_run_dec(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
                       u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *d_scratch,
                       u64 scratch_bytes) {
    return scl_aec_decode_batch((const scl_aec_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1, d_out_sym,
                                scl_round_up((u64)out_cap + 1, 16), out_cap, d_out_len, d_consumed, d_status,
                                d_scratch, scratch_bytes, nullptr);
}

extern "C" int scl_aec_encode_host(const scl_aec_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                                   uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {aec_run_enc, aec_slot, aec_scratch};
    return scl_host_encode_one(call, m, h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_aec_decode_host(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits, uint8_t *h_out_sym,
                                   uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {aec_run_dec, aec_scratch};
    return scl_host_decode_one(call, m, h_in, in_nbits, h_out_sym, out_cap, n_out, consumed);
}

// one block of a coder whose model state lives on the host between calls (what the drop-in classes do with the
// caller's freq_model object): upload -> code one block -> download
struct AecHostState {
    uint32_t *counts, *past_k;
};
static u64 aec_state1(const void *model) { return scl_aec_state_bytes((const scl_aec_model *)model, 1); }
static int aec_state_pre(const void *model, void *d_scratch, void *user) {
    const AecHostState *hs = (const AecHostState *)user;
    return scl_aec_state_upload((const scl_aec_model *)model, d_scratch, 1, 0, hs->counts, hs->past_k, nullptr);
}
static int aec_state_post(const void *model, const void *d_scratch, void *user) {
    const AecHostState *hs = (const AecHostState *)user;
    return scl_aec_state_download((const scl_aec_model *)model, d_scratch, 1, 0, hs->counts, hs->past_k, nullptr);
}
static int aec_run_enc_resume(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                              u32 *d_nbits, u32 *d_status, void *d_scratch, u64 scratch_bytes) {
    return scl_aec_encode_batch_resume((const scl_aec_model *)model, d_sym, n, nullptr, n, 1, d_out, out_stride,
                                       d_bit_off, d_nbits, d_status, d_scratch, scratch_bytes, 1, nullptr);
}
static int aec_run_dec_resume(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off,
                              const u32 *d_in_nbits, u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed,
                              u32 *d_status, void *d_scratch, u64 scratch_bytes) {
    return scl_aec_decode_batch_resume((const scl_aec_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1,
                                       d_out_sym, scl_round_up((u64)out_cap + 1, 16), out_cap, d_out_len, d_consumed,
                                       d_status, d_scratch, scratch_bytes, 1, nullptr);
}

extern "C" int scl_aec_encode_host_resume(const scl_aec_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                                          uint64_t out_cap_bytes, uint64_t *nbits, uint32_t *h_counts,
                                          uint32_t *h_past_k) {
    SCL_REQUIRE(m, "aec_encode_host_resume: null model");
    if (m->dev.kind == SCL_MODEL_FIXED) return scl_aec_encode_host(m, h_sym, n, h_out, out_cap_bytes, nbits);
    SCL_REQUIRE(h_counts && (m->dev.k == 0 || h_past_k), "aec_encode_host_resume: null state arrays");
    AecHostState hs = {h_counts, h_past_k};
    HostEncodeCall call = {aec_run_enc_resume, aec_slot, aec_state1, aec_state_pre, aec_state_post, &hs};
    return scl_host_encode_one(call, m, h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_aec_decode_host_resume(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                          uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed,
                                          uint32_t *h_counts, uint32_t *h_past_k) {
    SCL_REQUIRE(m, "aec_decode_host_resume: null model");
    if (m->dev.kind == SCL_MODEL_FIXED) return scl_aec_decode_host(m, h_in, in_nbits, h_out_sym, out_cap, n_out, consumed);
    SCL_REQUIRE(h_counts && (m->dev.k == 0 || h_past_k), "aec_decode_host_resume: null state arrays");
    AecHostState hs = {h_counts, h_past_k};
    HostDecodeCall call = {aec_run_dec_resume, aec_state1, aec_state_pre, aec_state_post, &hs};
    return scl_host_decode_one(call, m, h_in, in_nbits, h_out_sym, out_cap, n_out, consumed);
}

// ---- single-chunk host drivers, uint16 symbol indices --------------------------------------------------------
static int aec_run_enc16(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                         u32 *d_nbits, u32 *d_status, void *d_scratch, u64 scratch_bytes) {
    return scl_aec_encode_batch_u16((const scl_aec_model *)model, (const u16 *)d_sym, n, nullptr, n, 1, d_out, out_stride,
                                    d_bit_off, d_nbits, d_status, d_scratch, scratch_bytes, nullptr);
}
static int aec_run_dec16(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
                         u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *d_scratch,
                         u64 scratch_bytes) {
    return scl_aec_decode_batch_u16((const scl_aec_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1,
                                    (u16 *)d_out_sym, (u64)out_cap + 1, out_cap, d_out_len, d_consumed, d_status,
                                    d_scratch, scratch_bytes, nullptr);
}
static int aec_run_enc_resume16(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                                u32 *d_nbits, u32 *d_status, void *d_scratch, u64 scratch_bytes) {
    return scl_aec_encode_batch_resume_u16((const scl_aec_model *)model, (const u16 *)d_sym, n, nullptr, n, 1, d_out,
                                           out_stride, d_bit_off, d_nbits, d_status, d_scratch, scratch_bytes, 1,
                                           nullptr);
}
static int aec_run_dec_resume16(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off,
                                const u32 *d_in_nbits, u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed,
                                u32 *d_status, void *d_scratch, u64 scratch_bytes) {
    return scl_aec_decode_batch_resume_u16((const scl_aec_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1,
                                           (u16 *)d_out_sym, (u64)out_cap + 1, out_cap, d_out_len, d_consumed, d_status,
                                           d_scratch, scratch_bytes, 1, nullptr);
}

extern "C" int scl_aec_encode_host_u16(const scl_aec_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                                       uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {aec_run_enc16, aec_slot, aec_scratch};
    call.sym_bytes = 2;
    return scl_host_encode_one(call, m, (const u8 *)h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_aec_decode_host_u16(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                       uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {aec_run_dec16, aec_scratch};
    call.sym_bytes = 2;
    return scl_host_decode_one(call, m, h_in, in_nbits, (u8 *)h_out_sym, out_cap, n_out, consumed);
}

extern "C" int scl_aec_encode_host_resume_u16(const scl_aec_model *m, const uint16_t *h_sym, uint64_t n,
                                              uint8_t *h_out, uint64_t out_cap_bytes, uint64_t *nbits,
                                              uint32_t *h_counts, uint32_t *h_past_k) {
    SCL_REQUIRE(m, "aec_encode_host_resume_u16: null model");
    if (m->dev.kind == SCL_MODEL_FIXED) return scl_aec_encode_host_u16(m, h_sym, n, h_out, out_cap_bytes, nbits);
    SCL_REQUIRE(h_counts && (m->dev.k == 0 || h_past_k), "aec_encode_host_resume_u16: null state arrays");
    AecHostState hs = {h_counts, h_past_k};
    HostEncodeCall call = {aec_run_enc_resume16, aec_slot, aec_state1, aec_state_pre, aec_state_post, &hs};
    call.sym_bytes = 2;
    return scl_host_encode_one(call, m, (const u8 *)h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_aec_decode_host_resume_u16(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                              uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out,
                                              uint64_t *consumed, uint32_t *h_counts, uint32_t *h_past_k) {
    SCL_REQUIRE(m, "aec_decode_host_resume_u16: null model");
    if (m->dev.kind == SCL_MODEL_FIXED)
        return scl_aec_decode_host_u16(m, h_in, in_nbits, h_out_sym, out_cap, n_out, consumed);
    SCL_REQUIRE(h_counts && (m->dev.k == 0 || h_past_k), "aec_decode_host_resume_u16: null state arrays");
    AecHostState hs = {h_counts, h_past_k};
    HostDecodeCall call = {aec_run_dec_resume16, aec_state1, aec_state_pre, aec_state_post, &hs};
    call.sym_bytes = 2;
    return scl_host_decode_one(call, m, h_in, in_nbits, (u8 *)h_out_sym, out_cap, n_out, consumed);
}
