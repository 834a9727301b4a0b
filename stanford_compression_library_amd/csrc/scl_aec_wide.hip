// scl_aec_wide.hip -- adaptive order-k arithmetic coding on LARGE alphabets (BASELINE.json configs[3], second point:
// order-1 on bytes, K = 256), one wavefront lane per chunk, per-lane two-level count rows in device memory.  Round 3:
// the tuned arithmetic of scl_aec_fast.hip over the tables of scl_aec.hip.  Same streams, bit for bit, as scl_aec.hip and
//   ArithmeticEncoder.shrink_range / encode_block   scl/compressors/arithmetic_coding.py:58-78, :80-161
//   ArithmeticDecoder.decode_step_core / decode_block                               :177-201, :203-287
//   AdaptiveOrderKFreqModel                          scl/compressors/probability_models.py:95-160
//
// Served models (aec_wide_ok): AdaptiveOrderKFreqModel in the two-level row layout of scl_aec.hip (alphabet 17..256, more
// than 256 cells per chunk: 16 block totals + counts in blocks of 16, all stored as count - 1 in zero-filled scratch),
// PRECISION = 32, row totals that stay below 2^15 and below the model's rescale threshold for the whole chunk.
//
// Why it exists.  profiles/r03_aec_k256_pmc_summary.txt: the any-parameter kernels spend 830 (encode) / 1116 (decode)
// vector instructions per symbol on this model -- 64-bit divisions, literal renormalisation loops, a bit reader / writer
// that handles any width -- on top of one (encode) or two (decode) dependent round trips to a table that cannot be
// cached (272 KiB per lane).  Here the arithmetic is that of scl_aec_fast.hip (exact binary64 quotients with one
// reciprocal, closed-form renormalisation with the literal loops on the strict-comparison corners, 32-bit word I/O,
// ~150 instructions per symbol), and the ENCODER issues the two row reads of symbol i + 1 before it codes symbol i:
// its lookups do not depend on the coder state, only on the counts, so what symbol i adds to the rows that are already
// in flight is patched into the results (same context: total + 1; same context and a smaller / the same symbol:
// cumulative + 1 / frequency + 1).  The decoder learns its context from the symbol it has just decoded: two dependent
// round trips per symbol remain (block totals, then the block of counts the search lands in).
#include "scl_aec_internal.h"
#include "scl_aec_math.h"
#include "scl_aec_lane_io.h"

#define AW_THREADS 256

struct AecWideDev {
    u32 K;          // alphabet size 17..256
    u32 k;          // order 0..3
    u32 ctx_mod;    // K^k
    u32 row_cells;  // 16 + 16 * ceil(K / 16)
    u32 size_bits;  // DATA_BLOCK_SIZE_BITS (1..32)
    u64 cells;      // per chunk: ctx_mod * row_cells
};

// Cells are u16 here (the any-parameter kernels keep u32 cells in the same scratch: these kernels own their zero-filled
// scratch for one launch and use the first half of it): every count and block total stays below 2^15 (aec_wide_ok), and a
// group of 16 cells is 32 bytes -- the kernels are bound by the number of bytes their scattered row accesses move.
typedef u16 aw_cell;
struct AwRow {  // 16 consecutive cells
    u32 v[16];
};
// 16 cells, aligned to their size, as 16-byte loads issued back to back
__device__ __forceinline__ AwRow aw_load16(const aw_cell *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    AwRow r;
    const uint4 a = q[0], b = q[1];
    const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (u32 j = 0; j < 8; ++j) {
        r.v[2 * j] = w[j] & 0xFFFFu;
        r.v[2 * j + 1] = w[j] >> 16;
    }
    return r;
}
__device__ __forceinline__ u32 aw_next_ctx(const AecWideDev &P, u32 ctx, u32 s) {  // past_k[1:] + [s], :146-151
    if (P.k == 0) return 0;
    if (P.k == 1) return s;
    return (u32)(((u64)ctx * P.K + s) % P.ctx_mod);
}

__global__ void __launch_bounds__(AW_THREADS)
    aec_wide_encode_kernel(AecWideDev P, const u8 *__restrict__ sym, u64 sym_stride, const u32 *__restrict__ lens,
                           u32 chunk_len, u64 n_chunks, u8 *__restrict__ out, u64 out_stride,
                           u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits, u32 *__restrict__ status,
                           u32 *__restrict__ scratch) {
    const u64 chunk = (u64)blockIdx.x * AW_THREADS + threadIdx.x;
    if (chunk >= n_chunks) return;
    const u32 n = lens ? lens[chunk] : chunk_len;
    const u32 *src = reinterpret_cast<const u32 *>(sym + chunk * sym_stride);
    aw_cell *cnt = reinterpret_cast<aw_cell *>(scratch) + chunk * P.cells;
    AfWriter wr;
    wr.init(out + chunk * out_stride);
    wr.put(P.size_bits < 32 ? (n & ((1u << P.size_bits) - 1u)) : n, P.size_bits);  // header, :92-99
    u32 st = (P.size_bits < 32 && (n >> P.size_bits)) ? SCL_ST_SIZE : 0u;
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 pending = 0;  // E3 steps not yet resolved

    // arithmetic stage: shrink_range (:58-78) and the renormalisation loops (:126-150) of one symbol
    auto code = [&](u32 cc, u32 dd, u32 TT, double xx) {
        af_shrink2(low, hm, cc, dd, xx);
        u32 k, m, nlow, nhm;
        const bool edge = af_renorm2(low, hm, k, m, nlow, nhm);
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
                const u32 top = low >> (32 - k);
                const u32 b0 = top >> (k - 1);
                const u32 rest = top & ((1u << (k - 1)) - 1u);
                const u32 pat = (1u << pending) - (b0 ^ 1u);  // pending <= 31 here
                wr.put((pat << (k - 1)) | rest, k + pending);
                pending = 0;
            }
            pending += m;
            low = nlow;
            hm = nhm;
        }
    };

    // symbol i's rows are in flight since the previous iteration; what symbol i - 1 added to them (it was counted
    // after they were issued) is patched into the results.  Symbols arrive four per 32-bit load, two words ahead.
    const u32 last_word = n ? (n - 1) >> 2 : 0;
    u32 word_a = src[0], word_b = src[min(1u, last_word)];
    u32 ctx = 0;
    u32 s_cur = word_a & 0xFFu;
    if (n > 0 && s_cur >= P.K) st |= SCL_ST_SYMBOL;
    s_cur = (s_cur >= P.K) ? 0u : s_cur;
    AwRow bt = aw_load16(cnt);
    AwRow cb = aw_load16(cnt + 16 + (s_cur & ~15u));
    u32 p_ctx = 0xFFFFFFFFu, p_s = 0;  // the symbol counted while these rows were in flight (none yet)
    u32 c_pv = 0, d_pv = 1, T_pv = 1;  // (c, d, T) = (0, 1, 1): the arithmetic stage is a no-op before the first symbol
    double x_pv = 1.0;
    for (u32 i = 0; i < n; ++i) {
        const u32 s = s_cur, b = s >> 4, w = s & 15u;
        const u64 rb = (u64)ctx * P.row_cells;
        // next symbol: its rows are issued now, before this symbol is counted
        const u32 inx = i + 1;
        if ((inx & 3u) == 0) {
            word_a = word_b;
            word_b = src[min((inx >> 2) + 1, last_word)];
        }
        u32 s_nx = (word_a >> (8 * (inx & 3u))) & 0xFFu;
        if (inx < n && s_nx >= P.K) st |= SCL_ST_SYMBOL;
        s_nx = (s_nx >= P.K) ? 0u : s_nx;  // also past the end of the chunk: the rows read for it are never used
        const u32 ctx_nx = aw_next_ctx(P, ctx, s);
        const AwRow bt_nx = aw_load16(cnt + (u64)ctx_nx * P.row_cells);
        const AwRow cb_nx = aw_load16(cnt + (u64)ctx_nx * P.row_cells + 16 + (s_nx & ~15u));
        // meanwhile: the arithmetic of the previous symbol
        code(c_pv, d_pv, T_pv, x_pv);
        // freqs_current of this symbol (:118) from its rows (count[j] = 1 + cell)
        u32 below = 0, tot = 0, fs = 0, fb = 0;
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            tot += bt.v[j];
            below += (j < b) ? bt.v[j] : 0u;
            below += (j < w) ? cb.v[j] : 0u;
            fs = (j == w) ? cb.v[j] : fs;
            fb = (j == b) ? bt.v[j] : fb;
        }
        if (p_ctx == ctx) {  // symbol i - 1 had the same context: its count is missing from the rows read above
            const u32 pb = p_s >> 4;
            tot += 1;
            below += (p_s < s) ? 1u : 0u;
            fs += (p_s == s) ? 1u : 0u;
            fb += (pb == b) ? 1u : 0u;
        }
        c_pv = s + below;
        d_pv = c_pv + 1 + fs;
        T_pv = P.K + tot;
        x_pv = af_recip((double)T_pv);
        // update_model (:143-160): count[s] += 1, block total += 1 (cells hold count - 1)
        cnt[rb + 16 + s] = (aw_cell)(fs + 1);
        cnt[rb + b] = (aw_cell)(fb + 1);
        p_ctx = ctx;
        p_s = s;
        ctx = ctx_nx;
        s_cur = s_nx;
        bt = bt_nx;
        cb = cb_nx;
    }
    code(c_pv, d_pv, T_pv, x_pv);
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

__global__ void __launch_bounds__(AW_THREADS)
    aec_wide_decode_kernel(AecWideDev P, const u8 *__restrict__ in, u64 in_size_bytes, const u64 *__restrict__ bit_off,
                           const u32 *__restrict__ in_nbits, u64 n_chunks, u8 *__restrict__ out_sym, u64 out_stride,
                           u32 out_cap, u32 *__restrict__ out_lens, u32 *__restrict__ consumed,
                           u32 *__restrict__ status, u32 *__restrict__ scratch) {
    const u64 chunk = (u64)blockIdx.x * AW_THREADS + threadIdx.x;
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
    aw_cell *cnt = reinterpret_cast<aw_cell *>(scratch) + chunk * P.cells;
    u32 *dst = reinterpret_cast<u32 *>(out_sym + chunk * out_stride);
    u64 used = 32;
    u32 state = rd.get(32);
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 ctx = 0;
    u32 oword = 0;
    const u32 nblk = (P.K + 15) >> 4;
    AwRow bt = aw_load16(cnt);
    for (u32 i = 0;; ++i) {
        const u64 rb = (u64)ctx * P.row_cells;
        // ---- decode_step_core, :177-201 ----
        const double xr = af_recip((double)(hm - low) + 1.0);  // issued before the row arrives
        u32 tot = 0;
#pragma unroll
        for (u32 j = 0; j < 16; ++j) tot += bt.v[j];
        const u32 T = P.K + tot;
        const double xT = af_recip((double)T);
        // target = ((state - low + 1) * T - 1) // rng  (see scl_aec.hip), clamped for corrupt streams
        const double num = __builtin_fma((double)(state - low) + 1.0, (double)T, -0.5);
        u32 tgt = (u32)(num * xr);
        tgt = min(tgt, T - 1);
        // largest s with c[s] = s + extras below s <= target: the block first (c[16 j] = 16 j + totals below), ...
        u32 b = 0, g = 0, run = 0, fb = bt.v[0];
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            const bool take = j < nblk && run <= tgt;
            b = take ? j : b;
            g = take ? run : g;
            fb = take ? bt.v[j] : fb;
            run += 16 + bt.v[j];
        }
        // ... then the symbol inside it
        const AwRow cb = aw_load16(cnt + rb + 16 + 16 * b);
        const u32 wmax = min(15u, P.K - 1 - 16 * b);
        u32 w = 0, c = g, fs = cb.v[0];
        run = g;
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            const bool take = j <= wmax && run <= tgt;
            w = take ? j : w;
            c = take ? run : c;
            fs = take ? cb.v[j] : fs;
            run += 1 + cb.v[j];
        }
        const u32 s = 16 * b + w;
        const u32 d = c + 1 + fs;
        // update_model, then the next symbol's block totals: issued now, needed after the arithmetic below
        cnt[rb + 16 + s] = (aw_cell)(fs + 1);
        cnt[rb + b] = (aw_cell)(fb + 1);
        ctx = aw_next_ctx(P, ctx, s);
        bt = aw_load16(cnt + (u64)ctx * P.row_cells);
        af_shrink2(low, hm, c, d, xT);
        // ---- symbol out ----
        oword |= s << (8 * (i & 3));
        if ((i & 3) == 3) {
            dst[i >> 2] = oword;
            oword = 0;
        }
        if (i + 1 == n) break;  // before the renormalisation, :242-243
        // ---- renormalisation, :245-275 ----
        u32 k, m, nlow, nhm;
        const bool edge = af_renorm2(low, hm, k, m, nlow, nhm);
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
            state = af_state_shift_in(rd, state, k, kt);
            low = nlow;
            hm = nhm;
            used += kt;
        }
    }
    if ((n & 3) != 0) dst[(n - 1) >> 2] = oword;  // last, partial word (zero-padded inside the row)
    // how many of the last PRECISION bits belonged to the encoder (:277-282)
    const u64 lo = low, hi = (u64)hm + 1;
    u32 e = 0;
    for (; e < 32; ++e) {
        const u64 slo = ((u64)state >> e) << e, shi = slo + (1ull << e);
        if (slo < lo || shi > hi) break;
    }
    if (e == 32) e = 31;
    consumed[chunk] = (u32)((i64)(used + P.size_bits) - ((i64)e - 1));
    if (status) status[chunk] = st;
}

// ---- host side ----------------------------------------------------------------------------------------------
bool aec_wide_ok(const scl_aec_model *m, u64 max_symbols) {
    const AecDev &d = m->dev;
    if (d.kind != SCL_MODEL_ORDERK || !d.fenwick || d.P != 32) return false;
    const u64 total_max = (u64)d.K + max_symbols;  // bound on a row total and on any count
    return total_max < 32768 && total_max < d.max_total;
}

static AecWideDev aec_wide_dev(const scl_aec_model *m) {
    AecWideDev f;
    f.K = m->dev.K;
    f.k = m->dev.k;
    f.ctx_mod = (u32)m->dev.ctx_mod;
    f.row_cells = m->dev.row_cells;
    f.size_bits = m->dev.size_bits;
    f.cells = m->dev.cells;
    return f;
}

// bytes of the scratch these kernels use (and need zero-filled) for n_chunks chunks
u64 aec_wide_scratch_bytes(const scl_aec_model *m, u64 n_chunks) { return m->dev.cells * n_chunks * sizeof(aw_cell); }

void aec_wide_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                            u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                            u32 *d_status, u32 *d_scratch, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AW_THREADS - 1) / AW_THREADS);
    hipLaunchKernelGGL(aec_wide_encode_kernel, dim3(blocks), dim3(AW_THREADS), 0, st, aec_wide_dev(m), d_sym, sym_stride,
                       d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status, d_scratch);
}

void aec_wide_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                            const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                            u32 *d_out_lens, u32 *d_consumed, u32 *d_status, u32 *d_scratch, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AW_THREADS - 1) / AW_THREADS);
    hipLaunchKernelGGL(aec_wide_decode_kernel, dim3(blocks), dim3(AW_THREADS), 0, st, aec_wide_dev(m), d_in, in_size_bytes,
                       d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens, d_consumed, d_status,
                       d_scratch);
}
