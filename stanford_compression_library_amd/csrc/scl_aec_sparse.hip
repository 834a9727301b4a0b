// scl_aec_sparse.hip -- adaptive order-k arithmetic coding on LARGE alphabets with ONE table line per symbol
// (BASELINE.json configs[3], second point: order-1 on bytes, K = 256; round 3).  Same streams, bit for bit, as
// scl_aec_wide.hip / scl_aec.hip and
//   ArithmeticEncoder.shrink_range / encode_block   scl/compressors/arithmetic_coding.py:58-78, :80-161
//   ArithmeticDecoder.decode_step_core / decode_block                               :177-201, :203-287
//   AdaptiveOrderKFreqModel                          scl/compressors/probability_models.py:95-160
// Served models: those of scl_aec_wide.hip (aec_wide_ok).
//
// Why.  scl_aec_wide.hip is bound by its table accesses: two 64-byte pieces read and two written back per symbol, at the
// ~2 TB/s this chip delivers for lone 64-byte pieces at random addresses (profiles/r03_aec_k256_wide_pmc_summary.txt); a
// timing run that skipped one of the two pieces took 15.4 instead of 27.2 ms.  A two-level row cannot be one piece (256
// counts), but a context that has been seen a few times does not need a row: its counts ARE the list of symbols seen in
// it.  A 4 KiB chunk of order-1 byte data visits a context 16 times on average.
//
// Layout (inside the scratch of scl_aec_scratch_bytes, which is sized for u32 dense rows):
//   [n_chunks x ctx_mod lines of 64 bytes, zero-filled before the launch]
//   [n_chunks x ctx_mod dense rows of row_cells u16 cells, the layout of scl_aec_wide.hip; a row is written whole when its
//    context moves into it, so the launch zero-fills 16 KiB per chunk instead of 136]
// A line is {u32 n, up to 30 x u16 entry}: the first n entries hold symbol + 1 in arrival order, 0 = empty (the
// encoder uses all 30, the decoder 28: AP_WORDS_ENC / AP_WORDS_DEC below).
//   count of symbols < s   = n - #{entries > s}
//   count of symbol s      = #{entries > s} - #{entries > s + 1}          (+ 1 each: the model starts from all ones)
//   row total              = K + n
// (#{entries > v} for all of them at once: three packed-u16 instructions per word).  Counting a symbol appends one entry: one
// 2-byte store and the 4-byte header, same line.  The symbol that no longer fits moves the context to its dense row for good (the row
// is built in LDS and written out whole, header 0xFFFF); from then on it costs what scl_aec_wide.hip costs -- and a wave
// pays that whenever one of its 64 chunks is in such a context.
// The decoder has no order in the list to search by: it bisects.  With u = target - s, "s + (symbols below s) <= target"
// reads (symbols below target - u) <= u, u in 0..n: five steps of one packed count each.
#include "scl_aec_internal.h"
#include "scl_aec_math.h"
#include "scl_aec_lane_io.h"

#define AP_THREADS 64       // one wave per workgroup: the staging area below is 17 KiB, nine workgroups fit a CU
// words of entries per line (two u16 entries each; 15 = the whole line but its header).  Measured on one box: the encoder
// gains 2 % from the 15th word (fewer contexts outgrow their line), the decoder loses 6 % to it -- each kernel owns its
// lines, so each takes its own value.
#define AP_WORDS_ENC 15
#define AP_WORDS_DEC 14
#define AP_DENSE 0xFFFFu     // header of a context that lives in its dense row

struct AecSparseDev {
    u32 K;          // alphabet size 17..256
    u32 k;          // order 0..3
    u32 ctx_mod;    // K^k
    u32 row_cells;  // 16 + 16 * ceil(K / 16)
    u32 size_bits;  // DATA_BLOCK_SIZE_BITS (1..32)
    u64 dense_base; // u32 words in front of the dense rows: 16 * ctx_mod * n_chunks
};

struct ApLine {
    u32 w[16];  // w[0] = n or AP_DENSE; w[1..15] = entries, two per word
};
__device__ __forceinline__ ApLine ap_load_line(const u32 *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    ApLine L;
    L.w[0] = a.x, L.w[1] = a.y, L.w[2] = a.z, L.w[3] = a.w, L.w[4] = b.x, L.w[5] = b.y, L.w[6] = b.z, L.w[7] = b.w;
    L.w[8] = c.x, L.w[9] = c.y, L.w[10] = c.z, L.w[11] = c.w, L.w[12] = d.x, L.w[13] = d.y, L.w[14] = d.z, L.w[15] = d.w;
    return L;
}
// #{entries > v} over all 2 WORDS entries (empty entries are 0 and never count); v < 2^15
template <u32 WORDS>
__device__ __forceinline__ u32 ap_count_gt(const ApLine &L, u32 v) {
    const u32 vp = v | (v << 16);
    u32 a0 = 0, a1 = 0;
#pragma unroll
    for (u32 j = 1; j <= 14; j += 2) {
        a0 = af_pk_count_gt(a0, vp, L.w[j]);
        a1 = af_pk_count_gt(a1, vp, L.w[j + 1]);
    }
    if (WORDS == 15) a0 = af_pk_count_gt(a0, vp, L.w[15]);
    const u32 a = af_pk_add(a0, a1);  // minus the count, per half
    return (u32)(-((int32_t)(a << 16) >> 16) - ((int32_t)a >> 16));
}
__device__ __forceinline__ u32 ap_next_ctx(const AecSparseDev &P, u32 ctx, u32 s) {  // past_k[1:] + [s], :146-151
    if (P.k == 0) return 0;
    if (P.k == 1) return s;
    return (u32)(((u64)ctx * P.K + s) % P.ctx_mod);
}

// ---- the dense rows of contexts that have outgrown their line (layout of scl_aec_wide.hip, u16 cells) ---------------
struct ApRow16 {
    u32 v[16];
};
__device__ __forceinline__ ApRow16 ap_load16(const u16 *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1];
    const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    ApRow16 r;
#pragma unroll
    for (u32 j = 0; j < 8; ++j) {
        r.v[2 * j] = w[j] & 0xFFFFu;
        r.v[2 * j + 1] = w[j] >> 16;
    }
    return r;
}
// The context leaves its line: its dense row (u16 cells: 16 block totals, then the counts) is built in LDS -- a byte per
// cell, every count is at most 32 here -- and written out whole with plain 16-byte stores: no atomics on device memory, so
// this lane's later plain loads of the row see it without any fence (the row has never been read in this launch, and
// plain stores followed by plain loads of one lane are ordered).  `extra0` / `extra1` (symbol + 1, 0 = none) are counted
// on top of what the line holds.  LDS image: piece p (cells 16 p .. 16 p + 15) of lane t at [p][t].
#define AP_STAGE_PIECES 17
__device__ __forceinline__ void ap_stage_count(char *stage, u32 lane, u32 cell) {
    u32 *wp = reinterpret_cast<u32 *>(stage + (cell >> 4) * (AP_THREADS * 16) + lane * 16 + (cell & 12u));
    __hip_atomic_fetch_add(wp, 1u << (8 * (cell & 3u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <u32 WORDS>
__device__ __forceinline__ void ap_move_to_dense(const ApLine &L, u32 *line, u16 *row, u32 row_cells, u32 extra0, u32 extra1,
                                                 char *stage, u32 lane) {
    const u32 pieces = row_cells >> 4;
    for (u32 p = 0; p < pieces; ++p)
        *reinterpret_cast<uint4_lds *>(stage + p * (AP_THREADS * 16) + lane * 16) = make_uint4(0, 0, 0, 0);
    auto count = [&](u32 e) {  // e = symbol + 1, 0 = empty
        if (e) {
            ap_stage_count(stage, lane, 16 + (e - 1));
            ap_stage_count(stage, lane, (e - 1) >> 4);
        }
    };
#pragma unroll
    for (u32 j = 1; j <= WORDS; ++j) {
        count(L.w[j] & 0xFFFFu);
        count(L.w[j] >> 16);
    }
    count(extra0);
    count(extra1);
    uint4 *dst = reinterpret_cast<uint4 *>(row);
    for (u32 p = 0; p < pieces; ++p) {
        const uint4 b = *reinterpret_cast<const uint4_lds *>(stage + p * (AP_THREADS * 16) + lane * 16);
        const u32 w[4] = {b.x, b.y, b.z, b.w};
        u32 o[8];
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {  // four u8 cells -> four u16 cells
            o[2 * q] = (w[q] & 0xFFu) | ((w[q] & 0xFF00u) << 8);
            o[2 * q + 1] = ((w[q] >> 16) & 0xFFu) | ((w[q] >> 24) << 16);
        }
        dst[2 * p] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[2 * p + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
    line[0] = AP_DENSE;
}
// (c, f, T) of symbol s from a dense row: one round trip for the two pieces
struct ApCft {
    u32 c, f, T, fb;  // fb = the symbol's block total (count - 1 units), for the update
};
__device__ __forceinline__ ApCft ap_dense_lookup(const u16 *row, u32 K, u32 s) {
    const ApRow16 bt = ap_load16(row);
    const ApRow16 cb = ap_load16(row + 16 + (s & ~15u));
    const u32 b = s >> 4, w = s & 15u;
    u32 below = 0, tot = 0, fs = 0, fb = 0;
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
        tot += bt.v[j];
        below += (j < b) ? bt.v[j] : 0u;
        below += (j < w) ? cb.v[j] : 0u;
        fs = (j == w) ? cb.v[j] : fs;
        fb = (j == b) ? bt.v[j] : fb;
    }
    ApCft r;
    r.c = s + below;
    r.f = 1 + fs;
    r.T = K + tot;
    r.fb = fb;
    return r;
}

// ---- encoder ----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(AP_THREADS)
    aec_sparse_encode_kernel(AecSparseDev P, const u8 *__restrict__ sym, u64 sym_stride, const u32 *__restrict__ lens,
                             u32 chunk_len, u64 n_chunks, u8 *__restrict__ out, u64 out_stride,
                             u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits, u32 *__restrict__ status,
                             u32 *__restrict__ scratch) {
    __shared__ __attribute__((aligned(16))) char stage[AP_STAGE_PIECES * AP_THREADS * 16];
    const u64 chunk = (u64)blockIdx.x * AP_THREADS + threadIdx.x;
    if (chunk >= n_chunks) return;
    const u32 n = lens ? lens[chunk] : chunk_len;
    const u32 *src = reinterpret_cast<const u32 *>(sym + chunk * sym_stride);
    u32 *lines = scratch + chunk * (16ull * P.ctx_mod);
    u16 *dense = reinterpret_cast<u16 *>(scratch + P.dense_base) + chunk * ((u64)P.ctx_mod * P.row_cells);
    AfWriter wr;
    wr.init(out + chunk * out_stride);
    wr.put(P.size_bits < 32 ? (n & ((1u << P.size_bits) - 1u)) : n, P.size_bits);  // header, :92-99
    u32 st = (P.size_bits < 32 && (n >> P.size_bits)) ? SCL_ST_SIZE : 0u;
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 pending = 0;  // E3 steps not yet resolved

    // arithmetic stage: shrink_range (:58-78) and the renormalisation loops (:126-150) of one symbol
    auto code = [&](u32 cc, u32 dd, double xx) {
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

    // symbol i's line is in flight since the previous iteration; what symbol i - 1 added to it (it was counted after the
    // line was issued) is patched into the results.  Symbols arrive four per 32-bit load, two words ahead.
    const u32 last_word = n ? (n - 1) >> 2 : 0;
    u32 word_a = src[0], word_b = src[min(1u, last_word)];
    u32 ctx = 0;
    u32 s_cur = word_a & 0xFFu;
    if (n > 0 && s_cur >= P.K) st |= SCL_ST_SYMBOL;
    s_cur = (s_cur >= P.K) ? 0u : s_cur;
    ApLine L = ap_load_line(lines);
    u32 p_ctx = 0xFFFFFFFFu, p_s = 0;  // the symbol counted while this line was in flight (none yet)
    u32 c_pv = 0, d_pv = 1;            // (c, d, T) = (0, 1, 1): the arithmetic stage is a no-op before the first symbol
    double x_pv = 1.0;
    for (u32 i = 0; i < n; ++i) {
        const u32 s = s_cur;
        // next symbol: its line is issued now, before this symbol is counted
        const u32 inx = i + 1;
        if ((inx & 3u) == 0) {
            word_a = word_b;
            word_b = src[min((inx >> 2) + 1, last_word)];
        }
        u32 s_nx = (word_a >> (8 * (inx & 3u))) & 0xFFu;
        if (inx < n && s_nx >= P.K) st |= SCL_ST_SYMBOL;
        s_nx = (s_nx >= P.K) ? 0u : s_nx;  // also past the end of the chunk: the line read for it is never used
        const u32 ctx_nx = ap_next_ctx(P, ctx, s);
        const ApLine L_nx = ap_load_line(lines + 16ull * ctx_nx);
        // meanwhile: the arithmetic of the previous symbol
        code(c_pv, d_pv, x_pv);
        // freqs_current of this symbol (:118), then update_model (:143-160)
        u32 *line = lines + 16ull * ctx;
        u16 *row = dense + (u64)ctx * P.row_cells;
        const bool same = p_ctx == ctx;  // symbol i - 1 had this context: its entry is missing from L
        const u32 cnt = L.w[0] + (same ? 1u : 0u);  // entries the context has (line not dense), this symbol not among them
        u32 c, f, T;
        if (__builtin_expect(L.w[0] == AP_DENSE || cnt > (2 * AP_WORDS_ENC), 0)) {
            // dense -- when the line was read, or since the previous symbol (whose move the line in hand has not seen):
            // everything counted so far is in the row, read now
            const ApCft r = ap_dense_lookup(row, P.K, s);
            c = r.c, f = r.f, T = r.T;
            row[16 + s] = (u16)r.f;  // plain stores (visible to this lane's later loads), cells hold count - 1
            row[s >> 4] = (u16)(r.fb + 1);
        } else {
            const u32 g0 = ap_count_gt<AP_WORDS_ENC>(L, s) + ((same && p_s + 1 > s) ? 1u : 0u);
            const u32 g1 = ap_count_gt<AP_WORDS_ENC>(L, s + 1) + ((same && p_s + 1 > s + 1) ? 1u : 0u);
            c = s + cnt - g0;
            f = 1 + g0 - g1;
            T = P.K + cnt;
            if (cnt < (2 * AP_WORDS_ENC)) {
                reinterpret_cast<u16 *>(line)[2 + cnt] = (u16)(s + 1);
                line[0] = cnt + 1;
            } else {  // the line is full: it (and what is missing from it) goes into the dense row with this symbol
                ap_move_to_dense<AP_WORDS_ENC>(L, line, row, P.row_cells, same ? p_s + 1 : 0u, s + 1, stage, threadIdx.x);
            }
        }
        c_pv = c;
        d_pv = c + f;
        x_pv = af_recip((double)T);
        p_ctx = ctx;
        p_s = s;
        ctx = ctx_nx;
        s_cur = s_nx;
        L = L_nx;
    }
    code(c_pv, d_pv, x_pv);
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

// ---- decoder ----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(AP_THREADS)
    aec_sparse_decode_kernel(AecSparseDev P, const u8 *__restrict__ in, u64 in_size_bytes, const u64 *__restrict__ bit_off,
                             const u32 *__restrict__ in_nbits, u64 n_chunks, u8 *__restrict__ out_sym, u64 out_stride,
                             u32 out_cap, u32 *__restrict__ out_lens, u32 *__restrict__ consumed,
                             u32 *__restrict__ status, u32 *__restrict__ scratch) {
    __shared__ __attribute__((aligned(16))) char stage[AP_STAGE_PIECES * AP_THREADS * 16];
    const u64 chunk = (u64)blockIdx.x * AP_THREADS + threadIdx.x;
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
    u32 *lines = scratch + chunk * (16ull * P.ctx_mod);
    u16 *dense = reinterpret_cast<u16 *>(scratch + P.dense_base) + chunk * ((u64)P.ctx_mod * P.row_cells);
    u32 *dst = reinterpret_cast<u32 *>(out_sym + chunk * out_stride);
    u64 used = 32;
    u32 state = rd.get(32);
    u32 low = 0, hm = 0xFFFFFFFFu;
    u32 ctx = 0;
    u32 oword = 0;
    const u32 nblk = (P.K + 15) >> 4;
    ApLine L = ap_load_line(lines);
    for (u32 i = 0;; ++i) {
        u32 *line = lines + 16ull * ctx;
        u16 *row = dense + (u64)ctx * P.row_cells;
        // ---- decode_step_core, :177-201 ----
        const double xr = af_recip((double)(hm - low) + 1.0);  // issued before the line arrives
        u32 s, c, f, T;
        if (__builtin_expect(L.w[0] == AP_DENSE, 0)) {
            // the two-level search of scl_aec_wide.hip: block totals, then the block the target lands in
            const ApRow16 bt = ap_load16(row);
            u32 tot = 0;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) tot += bt.v[j];
            T = P.K + tot;
            const double num = __builtin_fma((double)(state - low) + 1.0, (double)T, -0.5);
            u32 tgt = (u32)(num * xr);
            tgt = min(tgt, T - 1);
            u32 b = 0, g = 0, run = 0, fb = bt.v[0];
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                const bool take = j < nblk && run <= tgt;
                b = take ? j : b;
                g = take ? run : g;
                fb = take ? bt.v[j] : fb;
                run += 16 + bt.v[j];
            }
            const ApRow16 cb = ap_load16(row + 16 + 16 * b);
            const u32 wmax = min(15u, P.K - 1 - 16 * b);
            u32 w = 0, fs = cb.v[0];
            c = g;
            run = g;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                const bool take = j <= wmax && run <= tgt;
                w = take ? j : w;
                c = take ? run : c;
                fs = take ? cb.v[j] : fs;
                run += 1 + cb.v[j];
            }
            s = 16 * b + w;
            f = 1 + fs;
            row[16 + s] = (u16)(fs + 1);  // plain stores, as in the encoder
            row[b] = (u16)(fb + 1);
        } else {
            const u32 cnt = L.w[0];
            T = P.K + cnt;
            // target = ((state - low + 1) * T - 1) // rng  (see scl_aec.hip), clamped for corrupt streams
            const double num = __builtin_fma((double)(state - low) + 1.0, (double)T, -0.5);
            u32 tgt = (u32)(num * xr);
            tgt = min(tgt, T - 1);
            // largest s <= K - 1 with s + (symbols below s) <= tgt.  With u = tgt - s: (symbols below tgt - u) <= u, the
            // smallest such u in 0..cnt; symbols below v = cnt - #{entries > v}.  u >= tgt - (K - 1) keeps s in the alphabet.
            const u32 u_min = tgt > P.K - 1 ? tgt - (P.K - 1) : 0u;
            u32 u = u_min;  // the answer lies in [u_min, u_min + 31]: cnt <= 28 entries can push it up by at most cnt
#pragma unroll
            for (u32 bit = 16; bit > 0; bit >>= 1) {
                // is u + bit - 1 still too small?  then the answer is at least u + bit
                const u32 t = u + bit - 1;
                const u32 v = tgt - min(t, tgt);  // candidate symbol (0 when t runs past tgt: then the test passes)
                const bool ok = (cnt - ap_count_gt<AP_WORDS_DEC>(L, v)) <= t;
                u = ok ? u : u + bit;
            }
            s = tgt - u;
            if (cnt < (2 * AP_WORDS_DEC)) {
                reinterpret_cast<u16 *>(line)[2 + cnt] = (u16)(s + 1);
                line[0] = cnt + 1;
            } else {  // the line is full: it goes into the dense row with this symbol
                ap_move_to_dense<AP_WORDS_DEC>(L, line, row, P.row_cells, 0u, s + 1, stage, threadIdx.x);
            }
        }
        // the symbol is known and counted: the next symbol's line is issued NOW (it is the one access this lane waits for),
        // (c, f) of this symbol and the arithmetic come out of the line in hand while it travels
        const ApLine Lc = L;
        const bool was_dense = Lc.w[0] == AP_DENSE;
        ctx = ap_next_ctx(P, ctx, s);
        L = ap_load_line(lines + 16ull * ctx);
        if (!was_dense) {
            const u32 cnt = Lc.w[0];
            const u32 g0 = ap_count_gt<AP_WORDS_DEC>(Lc, s), g1 = ap_count_gt<AP_WORDS_DEC>(Lc, s + 1);
            c = s + cnt - g0;
            f = 1 + g0 - g1;
        }
        const double xT = af_recip((double)T);
        af_shrink2(low, hm, c, c + f, xT);
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
static AecSparseDev aec_sparse_dev(const scl_aec_model *m, u64 n_chunks) {
    AecSparseDev f;
    f.K = m->dev.K;
    f.k = m->dev.k;
    f.ctx_mod = (u32)m->dev.ctx_mod;
    f.row_cells = m->dev.row_cells;
    f.size_bits = m->dev.size_bits;
    f.dense_base = 16ull * f.ctx_mod * n_chunks;
    return f;
}

// bytes at the front of the scratch that must be zero before a launch (the lines); lines + dense rows never need more than
// the u32 dense rows scl_aec_scratch_bytes is sized for (64 + 2 row_cells <= 4 row_cells for row_cells >= 32, i.e. every
// alphabet these kernels serve)
u64 aec_sparse_zero_bytes(const scl_aec_model *m, u64 n_chunks) { return 64ull * m->dev.ctx_mod * n_chunks; }

void aec_sparse_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                              u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                              u32 *d_status, u32 *d_scratch, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AP_THREADS - 1) / AP_THREADS);
    hipLaunchKernelGGL(aec_sparse_encode_kernel, dim3(blocks), dim3(AP_THREADS), 0, st, aec_sparse_dev(m, n_chunks), d_sym,
                       sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status,
                       d_scratch);
}

void aec_sparse_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, u32 *d_scratch, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AP_THREADS - 1) / AP_THREADS);
    hipLaunchKernelGGL(aec_sparse_decode_kernel, dim3(blocks), dim3(AP_THREADS), 0, st, aec_sparse_dev(m, n_chunks), d_in,
                       in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                       d_consumed, d_status, d_scratch);
}
