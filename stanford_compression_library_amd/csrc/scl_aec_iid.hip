// scl_aec_iid.hip -- adaptive i.i.d. arithmetic coding (AdaptiveIIDFreqModel) for alphabets up to 256 symbols
// with the lane's whole model in LDS, one wavefront lane per chunk.  Same streams, bit for bit, as scl_aec.hip
// and the reference:
//   ArithmeticEncoder / ArithmeticDecoder   scl/compressors/arithmetic_coding.py:58-161, :177-287
//   AdaptiveIIDFreqModel                     scl/compressors/probability_models.py:70-92
//
// Served models (aec_iid_ok): kind IID, alphabet 17..256 (smaller ones are a single row of scl_aec_fast.hip),
// PRECISION = 32, initial total + chunk length < 2^15 and below the model's rescale
// threshold (so the halving rule :86-92 cannot fire inside a chunk).
//
// Model layout: two levels of cumulative counts, all u16, 32-byte rows [row][thread] in LDS:
//   XB[16]      exclusive cumulative totals of the 16 blocks of 16 symbols   (XB[b] = count of all symbols < 16 b)
//   IC[b][16]   inclusive cumulative counts inside block b                   (IC[b][w] = sum of block b up to w)
// c[s] = XB[b] + IC[b][w-1], c[s] + f[s] = XB[b] + IC[b][w]  (b = s >> 4, w = s & 15): three 2-byte reads;
// count[s] += 1 is IC[b][j] += 1 for j >= w and XB[j] += 1 for j > b: sixteen v_pk_add_u16 with two mask rows;
// the decoder's search is two packed compare-and-counts (block, then symbol) with one dependent row read between.
// 17 rows x 32 B x 256 lanes = 136 KiB: one workgroup per CU, one wave per SIMD, like scl_aec_fast.hip, whose
// arithmetic (exact binary64 division, closed-form renormalisation) and per-lane I/O are reused unchanged.
// The generic kernel scans the 256 counts twice per symbol: 0.38 GB/s round trip on bytes.
#include <type_traits>
#include "scl_aec_internal.h"
#include "scl_aec_math.h"
#include "scl_aec_lane_io.h"

#define AI_THREADS 256
#define AI_ROW_BYTES (AI_THREADS * 32)          // one 32-byte row of every thread
#define AI_IC_BASE AI_ROW_BYTES                 // XB row first, then 16 block rows
#define AI_LUT_GE_BASE (17 * AI_ROW_BYTES)      // mask rows (j >= w)
#define AI_LUT_GT_BASE (AI_LUT_GE_BASE + 512)   // mask rows (j > b)
#define AI_OUT_BASE (AI_LUT_GT_BASE + 512)      // decoder: 64 bytes of symbol staging per lane (AfSymOut)
#define AI_LDS_BYTES (AI_LUT_GT_BASE + 512)
#define AI_DEC_LDS_BYTES (AI_OUT_BASE + AI_THREADS * 64)

struct AecIidDev {
    u32 K, total0;
    u32 size_bits;  // DATA_BLOCK_SIZE_BITS (1..32)
    const u32 *d_init;  // 17 rows x 8 packed u32: XB, then IC[0..15], built from the initial frequencies
};

struct AiRow {
    uint4 a, b;
};
__device__ __forceinline__ AiRow ai_row_add(const AiRow &R, const uint4 &ia, const uint4 &ib) {
    AiRow o;
    o.a = make_uint4(af_pk_add(R.a.x, ia.x), af_pk_add(R.a.y, ia.y), af_pk_add(R.a.z, ia.z), af_pk_add(R.a.w, ia.w));
    o.b = make_uint4(af_pk_add(R.b.x, ib.x), af_pk_add(R.b.y, ib.y), af_pk_add(R.b.z, ib.z), af_pk_add(R.b.w, ib.w));
    return o;
}
__device__ __forceinline__ void ai_setup_tables(char *lds, const AecIidDev &P, u32 tid) {
    if (tid < 128) {  // mask rows, register r of a row holds elements 2r and 2r+1
        const u32 s = tid >> 3, r = tid & 7;
        const u32 ge = ((2 * r >= s) ? 1u : 0u) | ((2 * r + 1 >= s) ? 0x10000u : 0u);
        const u32 gt = ((2 * r > s) ? 1u : 0u) | ((2 * r + 1 > s) ? 0x10000u : 0u);
        *reinterpret_cast<u32_lds *>(lds + AI_LUT_GE_BASE + s * 32 + r * 4) = ge;
        *reinterpret_cast<u32_lds *>(lds + AI_LUT_GT_BASE + s * 32 + r * 4) = gt;
    }
    const uint4 *init = reinterpret_cast<const uint4 *>(P.d_init);
    for (u32 row = 0; row < 17; ++row) {
        *reinterpret_cast<uint4_lds *>(lds + row * AI_ROW_BYTES + tid * 32) = init[2 * row];
        *reinterpret_cast<uint4_lds *>(lds + row * AI_ROW_BYTES + tid * 32 + 16) = init[2 * row + 1];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(AI_THREADS)
    aec_iid_encode_kernel(AecIidDev P, const u8 *__restrict__ sym, u64 sym_stride, const u32 *__restrict__ lens,
                          u32 chunk_len, u64 n_chunks, u8 *__restrict__ out, u64 out_stride,
                          u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits, u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[AI_DEC_LDS_BYTES];  // tables + 64 bytes of stream staging per lane
    const u32 tid = threadIdx.x;
    ai_setup_tables(lds, P, tid);
    const u64 chunk = (u64)blockIdx.x * AI_THREADS + tid;
    if (chunk >= n_chunks) return;
    const u32 n = lens ? lens[chunk] : chunk_len;
    const u32 *src = reinterpret_cast<const u32 *>(sym + chunk * sym_stride);
    AfWriterT<true> wr;
    wr.init(out + chunk * out_stride, lds + AI_OUT_BASE, tid);
    wr.put(P.size_bits < 32 ? (n & ((1u << P.size_bits) - 1u)) : n, P.size_bits);  // header, :92-99
    u32 st = (P.size_bits < 32 && (n >> P.size_bits)) ? SCL_ST_SIZE : 0u;
    u32 low = 0, hm = 0xFFFFFFFFu, pending = 0;
    u32 T = P.total0;  // the total of an i.i.d. model is its initial total plus the symbols seen
    u32 nextw = 0;
    const u32 last_word = n ? (n - 1) >> 2 : 0;
    u32 c_nx = 0, d_nx = 1, T_nx = 1;
    double x_nx = 1.0;
    AiRow XB;  // the block-level row also lives in registers (one context: it is never re-read)
    XB.a = *reinterpret_cast<const uint4_lds *>(lds + tid * 32);
    XB.b = *reinterpret_cast<const uint4_lds *>(lds + tid * 32 + 16);

    u32 m_rowaddr = 0, m_xb = 0, m_ic = 0, m_icm1 = 0, m_w = 0;
    AiRow m_IC;
    uint4 m_ge_a, m_ge_b, m_gt_a, m_gt_b;
    // model stage, split as in scl_aec_fast.hip: loads first, update and 1/T after the previous symbol is coded
    auto model_issue = [&](u32 s) {
        const u32 b = s >> 4, w = s & 15u;
        m_w = w;
        m_rowaddr = AI_IC_BASE + b * AI_ROW_BYTES + tid * 32;
        m_xb = *reinterpret_cast<const u16_lds *>(lds + tid * 32 + 2 * b);
        m_ic = *reinterpret_cast<const u16_lds *>(lds + m_rowaddr + 2 * w);
        m_icm1 = *reinterpret_cast<const u16_lds *>(lds + m_rowaddr + 2 * w - 2);  // unused when w == 0
        m_IC.a = *reinterpret_cast<const uint4_lds *>(lds + m_rowaddr);
        m_IC.b = *reinterpret_cast<const uint4_lds *>(lds + m_rowaddr + 16);
        m_ge_a = *reinterpret_cast<const uint4_lds *>(lds + AI_LUT_GE_BASE + w * 32);
        m_ge_b = *reinterpret_cast<const uint4_lds *>(lds + AI_LUT_GE_BASE + w * 32 + 16);
        m_gt_a = *reinterpret_cast<const uint4_lds *>(lds + AI_LUT_GT_BASE + b * 32);
        m_gt_b = *reinterpret_cast<const uint4_lds *>(lds + AI_LUT_GT_BASE + b * 32 + 16);
    };
    auto model_finish = [&]() {
        c_nx = m_xb + (m_w ? m_icm1 : 0u);
        d_nx = m_xb + m_ic;
        T_nx = T;
        x_nx = af_recip((double)T);
        // update_model (:83-85): count[s] += 1
        const AiRow IC2 = ai_row_add(m_IC, m_ge_a, m_ge_b);
        *reinterpret_cast<uint4_lds *>(lds + m_rowaddr) = IC2.a;
        *reinterpret_cast<uint4_lds *>(lds + m_rowaddr + 16) = IC2.b;
        XB = ai_row_add(XB, m_gt_a, m_gt_b);
        *reinterpret_cast<uint4_lds *>(lds + tid * 32) = XB.a;
        *reinterpret_cast<uint4_lds *>(lds + tid * 32 + 16) = XB.b;
        T += 1;
    };
    auto code = [&](u32 cc, u32 dd, u32 TT, double xx) {
        af_shrink2(low, hm, cc, dd, xx);
        u32 k, m, nlow, nhm;
        const bool edge = af_renorm2_dec(low, hm, k, m, nlow, nhm);  // the conservative corner test: two compares fewer
        const bool rare = edge | (k + pending > 32);  // one condition, one branch
        if (__builtin_expect(rare, 0)) {
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
            // b0, then `pending` copies of !b0, then the other k - 1 common bits -- without a branch on k: for k = 0 the
            // field is empty (v = 0, nb = 0) and the pending count just grows
            const bool any = k != 0;
            const u32 km1 = (k - 1u) & 31u;
            const u32 b0 = low >> 31;
            const u32 rest = __builtin_amdgcn_ubfe(low, (32u - k) & 31u, km1);  // bits 30 .. 32-k of low
            const u32 pat = (1u << pending) - (b0 ^ 1u);                       // pending <= 31 here
            u32 fv = (pat << km1) | rest, fn = k + pending;
            asm volatile("" : "+v"(fv), "+v"(fn));  // computed for every lane: the compiler would turn the selects into a branch
            wr.put_field(any ? fv : 0u, any ? fn : 0u);
            pending = (any ? 0u : pending) + m;
            low = nlow;
            hm = nhm;
        }
    };

    // (c, d, T, 1/T) = (0, 1, 1, 1.0) before the first symbol makes `code` a no-op, see scl_aec_fast.hip
    const u32 n_words = (n + 3) >> 2;
    if (n > 0) nextw = src[0];
    for (u32 w = 0; w < n_words; ++w) {
        u32 word = nextw;
        nextw = src[min(w + 1, last_word)];  // unconditional, one word ahead
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
            model_issue(s);
            code(cc, dd, TT, xx);
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

__global__ void __launch_bounds__(AI_THREADS)
    aec_iid_decode_kernel(AecIidDev P, const u8 *__restrict__ in, u64 in_size_bytes, const u64 *__restrict__ bit_off,
                          const u32 *__restrict__ in_nbits, u64 n_chunks, u8 *__restrict__ out_sym, u64 out_stride,
                          u32 out_cap, u32 *__restrict__ out_lens, u32 *__restrict__ consumed,
                          u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[AI_DEC_LDS_BYTES];
    const u32 tid = threadIdx.x;
    ai_setup_tables(lds, P, tid);
    const u64 chunk = (u64)blockIdx.x * AI_THREADS + tid;
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
    so.init(lds + AI_OUT_BASE, tid, out_sym + chunk * out_stride);
    u32 state = rd.get(32);
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 T = P.total0;
    AiRow XB;
    XB.a = *reinterpret_cast<const uint4_lds *>(lds + tid * 32);
    XB.b = *reinterpret_cast<const uint4_lds *>(lds + tid * 32 + 16);
    // One symbol: decode_step_core (:177-201), update_model, symbol out; the last symbol of the chunk follows the loop without
    // a renormalisation (the reference breaks before it, :242-243), so the loop has a single exit test.
    // low <= state <= hm holds for ANY input bits (the symbol chosen is the one whose interval holds the state), so
    // target <= T - 1, and with it b <= last block (XB of the blocks past the alphabet is T) and w <= the last symbol of the
    // block (its sums stay at the block total past the alphabet): the searches need no clamps.  b <= 15 structurally
    // (XB[0] = 0 <= target), which is what the addresses need; the guard on s is for the symbol written.
    auto step = [&](u32 i) {
        const double xr = af_recip((double)(hm - low) + 1.0);
        const double num = __builtin_fma((double)(state - low) + 1.0, (double)T, -0.5);
        const u32 tgt = (u32)(num * xr);  // ((state - low + 1) * T - 1) // rng, see scl_aec.hip
        // block: largest b with XB[b] <= target  (XB[0] = 0 always counts).  Both searches hand back their compare masks:
        // the rows are sorted, so "entry > target" is exactly the set update_model increments (af_pk_search16)
        const u32 XBv[8] = {XB.a.x, XB.a.y, XB.a.z, XB.a.w, XB.b.x, XB.b.y, XB.b.z, XB.b.w};
        u32 xbm[8], icm[8];
        const u32 b = (af_pk_search16(XBv, tgt, xbm) - 1u) & 15u;
        const u32 rowaddr = AI_IC_BASE + b * AI_ROW_BYTES + tid * 32;
        u32 xb = *reinterpret_cast<const u16_lds *>(lds + tid * 32 + 2 * b);
        AiRow IC;
        IC.a = *reinterpret_cast<const uint4_lds *>(lds + rowaddr);
        IC.b = *reinterpret_cast<const uint4_lds *>(lds + rowaddr + 16);
        // 1/T: its seven instructions go here, while the row is in flight (the empty asm statements pin it between the block
        // search and the first use of the row; they also keep the 16-bit reads 32 bits wide -- narrowed to 16-bit operations
        // each costs a v_and 0xffff)
        u32 T2 = T;
        asm volatile("" : "+v"(T2) : "v"(b));
        double xT = af_recip((double)T2);
        asm volatile("" : "+v"(xT));
        asm volatile("" : "+v"(xb));
        // symbol inside the block: number of inclusive sums <= target - XB[b]
        const u32 t2 = tgt - xb;
        const u32 ICv[8] = {IC.a.x, IC.a.y, IC.a.z, IC.a.w, IC.b.x, IC.b.y, IC.b.z, IC.b.w};
        const u32 w = af_pk_search16(ICv, t2, icm);
        const u32 s = min(16 * b + w, P.K - 1);
        u32 ic = *reinterpret_cast<const u16_lds *>(lds + rowaddr + 2 * w);
        u32 icm1 = *reinterpret_cast<const u16_lds *>(lds + rowaddr + 2 * w - 2);
        // update_model
        *reinterpret_cast<uint4_lds *>(lds + rowaddr) =
            make_uint4(af_pk_sub(ICv[0], icm[0]), af_pk_sub(ICv[1], icm[1]), af_pk_sub(ICv[2], icm[2]), af_pk_sub(ICv[3], icm[3]));
        *reinterpret_cast<uint4_lds *>(lds + rowaddr + 16) =
            make_uint4(af_pk_sub(ICv[4], icm[4]), af_pk_sub(ICv[5], icm[5]), af_pk_sub(ICv[6], icm[6]), af_pk_sub(ICv[7], icm[7]));
        XB.a = make_uint4(af_pk_sub(XBv[0], xbm[0]), af_pk_sub(XBv[1], xbm[1]), af_pk_sub(XBv[2], xbm[2]), af_pk_sub(XBv[3], xbm[3]));
        XB.b = make_uint4(af_pk_sub(XBv[4], xbm[4]), af_pk_sub(XBv[5], xbm[5]), af_pk_sub(XBv[6], xbm[6]), af_pk_sub(XBv[7], xbm[7]));
        *reinterpret_cast<uint4_lds *>(lds + tid * 32) = XB.a;
        *reinterpret_cast<uint4_lds *>(lds + tid * 32 + 16) = XB.b;
        asm volatile("" : "+v"(ic), "+v"(icm1));
        // w = 0 <=> IC[0] > target - XB[b] <=> the low half of icm[0] is all ones
        const u32 c = xb + (icm1 & ~icm[0]), d = xb + ic;
        af_shrink2(low, hm, c, d, xT);
        T += 1;
        so.put(s, i);  // symbol out: whole 64-byte sectors (AfSymOut)
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
    // Stretches of symbols with the unchecked refill and a scalar trip count, then the checked loop for the chunks' last
    // symbols: as in scl_aec_fast.hip (the symbol index i is the same for every lane still at work).
    u32 i = 0;
    for (;;) {
        const bool at_work = i + 1 < n;
        const u32 mine = at_work ? min(rd.safe_symbols(), n - 1 - i) : 0xFFFFFFFFu;
        const u32 cnt = __builtin_amdgcn_readfirstlane(af_wave_min(mine));
        if (cnt == 0xFFFFFFFFu || cnt < 16) break;
        if (at_work) {
            const u32 *before = rd.ptr;
            // four symbols per trip once the index is a multiple of four (see scl_aec_fast.hip): the byte a symbol fills in
            // the output word and the word's completion are compile-time facts, the trip count is tested once per four
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
bool aec_iid_ok(const scl_aec_model *m, u64 max_symbols) {
    const AecDev &d = m->dev;
    if (d.kind != SCL_MODEL_IID || d.K < 17 || d.K > 256 || d.P != 32 || !m->d_iid_init)
        return false;
    const u64 total_max = (u64)d.total0 + max_symbols;
    return total_max < 32768 && total_max < d.max_total;
}

// 17 rows of 16 u16 (packed in u32 pairs) from the initial frequencies: XB, then IC[0..15]
void aec_iid_build_init(const u32 *h_freq, u32 K, u32 *out136) {
    u32 rows[17][16];
    u32 acc = 0;
    for (u32 b = 0; b < 16; ++b) {
        rows[0][b] = acc;
        u32 in_block = 0;
        for (u32 w = 0; w < 16; ++w) {
            const u32 s = 16 * b + w;
            if (s < K) in_block += h_freq[s];
            rows[1 + b][w] = in_block;
        }
        acc += in_block;
    }
    for (u32 r = 0; r < 17; ++r)
        for (u32 j = 0; j < 8; ++j) out136[r * 8 + j] = rows[r][2 * j] | (rows[r][2 * j + 1] << 16);
}

static AecIidDev aec_iid_dev(const scl_aec_model *m) {
    AecIidDev f;
    f.K = m->dev.K;
    f.total0 = m->dev.total0;
    f.size_bits = m->dev.size_bits;
    f.d_init = m->d_iid_init;
    return f;
}

void aec_iid_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                           u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                           u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AI_THREADS - 1) / AI_THREADS);
    hipLaunchKernelGGL(aec_iid_encode_kernel, dim3(blocks), dim3(AI_THREADS), 0, st, aec_iid_dev(m), d_sym, sym_stride,
                       d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status);
}

void aec_iid_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                           const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                           u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AI_THREADS - 1) / AI_THREADS);
    hipLaunchKernelGGL(aec_iid_decode_kernel, dim3(blocks), dim3(AI_THREADS), 0, st, aec_iid_dev(m), d_in,
                       in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                       d_consumed, d_status);
}
