// scl_aec_split.hip -- adaptive arithmetic ENCODER with the work of one chunk split over three wavefront lanes in three
// different waves (BASELINE.json configs[3]; round 3).  Same streams, bit for bit, as scl_aec_fast.hip / scl_aec.hip and
//   ArithmeticEncoder.shrink_range / encode_block   scl/compressors/arithmetic_coding.py:58-78, :80-161
//   AdaptiveIIDFreqModel / AdaptiveOrderKFreqModel   scl/compressors/probability_models.py:70-92, :95-160
// Served models: those of scl_aec_fast.hip (aec_fast_ok).
//
// Why.  A lane's 16 context rows take 512 B of LDS, so a CU holds 256 chunks = ONE wave per SIMD with one lane per
// chunk, and a lone wave issues one instruction per ~2.3 ns whatever it is (profiles/r02_ubench_valu_rate.txt).  The
// rows cannot be halved (256 counters of 13 bits), but the per-symbol work of a chunk is a feed-forward pipeline:
//   model  : look up (c, d, T) for the symbol, count it            -- independent of the coder state, can run ahead
//   coder  : shrink_range, count the renormalisation steps, new (low, high)   -- the only serial state
//   writer : turn (top bits of low, k, pending) into stream bits, store words -- consumes, never feeds back
// So a workgroup of 3 AS_LANES lanes serves AS_LANES chunks: lane L of the model wave(s) owns the tables of chunk L, lane
// L of the coder wave(s) its (low, high, pending), lane L of the writer wave(s) its output window -- three waves per SIMD
// over the same tables, each running a third of the instruction stream.  AS_LANES = 64 (one wave per role, four 40 KiB
// workgroups per CU whose barriers are independent) measured 2 % faster than one 768-lane workgroup per CU; 128 is much
// slower (8.1 vs 5.1 ms: six-wave workgroups land their roles unevenly on the four SIMDs).  The roles meet in two LDS FIFOs, AS_TILE
// symbols per lane and round: in round r the model fills tile r, the coder drains tile r - 1, the writer tile r - 2.  With
// one buffer per FIFO (AS_NBUF = 1, shipped: 40 KiB hold tiles of four symbols) a round has two barriers: consumers take
// their tile into registers, barrier, producers overwrite it, barrier; AS_NBUF = 2 needs one barrier and twice the FIFO
// space (tiles of two at four workgroups per CU: 5.10 ms, tiles of four at three workgroups: 5.59, shipped form: 4.97).
// Measured (profiles/r03_aec_split_note.txt): 7.9 -> 5.3 ms per GiB of order-1 K = 16 data; the roles alone take
// 0.86 (model) / ~0.9 / ~0.9 ms per 256 MiB, all three together 1.38 -- the VALU is ~70 % busy, the rest is LDS time.
//
// Model side.  Context rows are 16 u16 EXCLUSIVE cumulative counts as in scl_aec_fast.hip, but laid out in planes:
// word w (two counts) of context `ctx` of lane t at (ctx * 8 + w) * AS_PLANE + 4 t -- every 4-byte access of a wave is
// conflict free whatever the lanes' contexts and symbols.  `count[s] += 1` is X[j] += 1 for j > s: eight ds_add_u32
// with an addend row from a 512-byte LUT (no read-modify-write through registers, so all LDS traffic of a tile is
// issued back to back and waited for once); c = X[s], d = X[s + 1] come out of ONE two-word read (ds_read2st64) and
// one v_alignbit.  Row totals are u16 [ctx][lane], counted by a ds_add_u32 on the pair's word.
// Coder side.  hm' = low + ((rng * d) // T - 1) is evaluated as trunc(fma(fma(rng, d, .5), 1/T, -1)): no special case
// for d == T (where the quotient may be 2^32), so FIFO 1 carries just c | d << 16 and 1/T as binary64 = 12 B.  The
// closed-form step counts k, m and the corner test are those of scl_aec_math.h, the test rewritten on the shifted
// values (a power-of-two boundary inside the shifted-out prefix <=> the new low is 0 / the new high - 1 all ones).
// Writer side.  64-bit window of 32-bit halves, one field of <= 31 bits per symbol (b0, pending x !b0, k - 1 more
// bits); k + pending > 31 and the termination take the bit-run path.
#include "scl_aec_internal.h"
#include "scl_aec_math.h"
#include "scl_aec_lane_io.h"

#define AS_LANES 64                    // chunks per workgroup: one wave per role, four workgroups per CU (40 KiB of LDS each)
#define AS_THREADS (3 * AS_LANES)      // wave 0: model role, wave 1: coder role, wave 2: writer role
#define AS_PLANE (AS_LANES * 4)        // one u32 of every chunk
#define AS_ROW_BYTES (8 * AS_PLANE)    // 8 KiB per context
#define AS_TABLE_BYTES (16 * AS_ROW_BYTES)
// The row TOTAL lives in the row itself (round 6): X[0], the exclusive cumulative count of symbol 0, is always 0, so its
// u16 carries the total instead -- the addend rows count it up with every symbol (one LDS atomic per symbol less than the
// separate totals array of rounds 3-5 took), the lookup masks it out for s = 0.
#define AS_LUT_BASE AS_TABLE_BYTES
#define AS_LUT_BYTES 512
#define AS_TILE 4                      // symbols per lane and barrier: 2 or 4 (a 32-bit word of symbols is 4 / AS_TILE tiles)
#define AS_RPW (4 / AS_TILE)            // rounds per word of symbols
#define AS_NBUF 1                      // 2: double-buffered FIFOs, one barrier per round; 1: one buffer, consumers take their
                                       // tile into registers first and a second barrier per round releases the buffer
// FIFO 1 (model -> coder), two buffers of AS_TILE symbols: 1/T as binary64 [buf][j][lane], c | d << 16 [buf][j][lane]
#define AS_F1X_BASE (AS_LUT_BASE + AS_LUT_BYTES)
#define AS_F1X_SLOT (AS_LANES * 8)
#define AS_F1C_BASE (AS_F1X_BASE + AS_NBUF * AS_TILE * AS_F1X_SLOT)
#define AS_F1C_SLOT (AS_LANES * 4)
// FIFO 2 (coder -> writer): {top k bits of low, k | pending << 8} [buf][j][lane]
#define AS_F2_BASE (AS_F1C_BASE + AS_NBUF * AS_TILE * AS_F1C_SLOT)
#define AS_F2_SLOT (AS_LANES * 8)
#define AS_RED_BASE (AS_F2_BASE + AS_NBUF * AS_TILE * AS_F2_SLOT)
#define AS_LDS_BYTES (AS_RED_BASE + 16)
static_assert(AS_LDS_BYTES <= 160 * 1024, "LDS of one CU");

struct AecSplitDev {
    u32 K;          // alphabet size 2..16
    u32 nctx;       // K^k <= 16
    u32 ctx_magic;  // ceil(2^16 / nctx): (v * magic) >> 16 == v / nctx for v < 272
    u32 total0;     // initial total of a row
    u32 size_bits;  // DATA_BLOCK_SIZE_BITS (1..32)
    u32 initX[8];   // 16 packed u16: EXCLUSIVE cumulative initial counts, padded with the total
};

template <bool ORDER1>
__device__ __forceinline__ u32 as_next_ctx(const AecSplitDev &P, u32 ctx, u32 s) {  // past_k[1:] + [s], :146-151
    if (ORDER1) return s;
    // (24-bit multiplies: v < 272, magic <= 2^15, nctx <= 16 -- the 32-bit v_mul_lo_u32 is a quarter-rate instruction)
    const u32 v = __umul24(ctx, P.K) + s;
    return v - __umul24(__umul24(v, P.ctx_magic) >> 16, P.nctx);
}

template <bool ORDER1>
__global__ void __launch_bounds__(AS_THREADS)
    aec_split_encode_kernel(AecSplitDev P, const u8 *__restrict__ sym, u64 sym_stride, const u32 *__restrict__ lens,
                            u32 chunk_len, u64 n_chunks, u8 *__restrict__ out, u64 out_stride,
                            u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits, u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[AS_LDS_BYTES];
    const u32 tid = threadIdx.x;
    const u32 lane = tid & (AS_LANES - 1);
    const u32 role = tid / AS_LANES;  // 0 model, 1 coder, 2 writer (wave-uniform)
    const u64 chunk = (u64)blockIdx.x * AS_LANES + lane;
    const bool live = chunk < n_chunks;
    const u32 n = live ? (lens ? lens[chunk] : chunk_len) : 0u;

    // ---- tables, LUT, the longest chunk of the workgroup ----------------------------------------------------------
    if (tid < 128) {
        const u32 s = tid >> 3, r = tid & 7;  // addend of word r (counts 2r, 2r + 1) for symbol s: [j > s]
        // (count 0 of a row is its total: + 1 for every symbol)
        const u32 v = ((2 * r > s || r == 0) ? 1u : 0u) | ((2 * r + 1 > s) ? 0x10000u : 0u);
        *reinterpret_cast<u32_lds *>(lds + AS_LUT_BASE + s * 32 + r * 4) = v;
    }
    if (tid == 0) *reinterpret_cast<u32_lds *>(lds + AS_RED_BASE) = 0;
    if (role == 0) {
        for (u32 c = 0; c < P.nctx; ++c) {
#pragma unroll
            for (u32 w = 0; w < 8; ++w)
                *reinterpret_cast<u32_lds *>(lds + c * AS_ROW_BYTES + w * AS_PLANE + lane * 4) =
                    w ? P.initX[w] : ((P.initX[0] & 0xFFFF0000u) | (P.total0 & 0xFFFFu));
        }
    }
    __syncthreads();
    __hip_atomic_fetch_max(reinterpret_cast<u32 *>(lds + AS_RED_BASE), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const u32 nmax = *reinterpret_cast<const u32_lds *>(lds + AS_RED_BASE);
    // round r: the model fills tile r (symbols 2r, 2r + 1), the coder drains tile r - 1, the writer tile r - 2
    // AS_RPW rounds per 32-bit word of symbols; two more rounds drain the pipeline
    const u32 n_words = ((nmax + AS_TILE - 1) / AS_TILE + 2 + AS_RPW - 1) / AS_RPW;
    const u32 n_rounds = AS_RPW * n_words;

    if (role == 0) {
        // =============================== model role ===================================================================
        // Straight-line code per tile, for every lane whatever its length (past its end a lane counts symbols nobody
        // reads): all LDS traffic of a tile is issued back to back -- the addend rows of the word's four symbols first,
        // then per symbol one two-word read, the total, and nine adds that need no read result -- and waited for once.
        const u32 *src = reinterpret_cast<const u32 *>(sym + (live ? chunk : 0) * sym_stride);
        const u32 last_word = n ? (n - 1) >> 2 : 0;
        u32 st = 0;
        u32 ctx = 0;
        u32 nextw = src[0];
        const u32 lane4 = lane * 4;
        for (u32 w = 0; w < n_words; ++w) {
            const u32 word = nextw;
            nextw = src[min(w + 1, last_word)];  // one word ahead, never conditional
            u32 s[4];
            uint4 la[4], lb[4];
#pragma unroll
            for (u32 q = 0; q < 4; ++q) {
                s[q] = (word >> (8 * q)) & 0xFFu;
                const bool bad = s[q] >= P.K;
                if (bad && 4 * w + q < n) st |= SCL_ST_SYMBOL;
                s[q] = bad ? 0u : s[q];
                la[q] = *reinterpret_cast<const uint4_lds *>(lds + AS_LUT_BASE + s[q] * 32);
                lb[q] = *reinterpret_cast<const uint4_lds *>(lds + AS_LUT_BASE + s[q] * 32 + 16);
            }
#pragma unroll
            for (u32 half = 0; half < AS_RPW; ++half) {
                const u32 buf = (AS_NBUF == 2) ? ((w * AS_RPW + half) & 1) : 0;  // round w * AS_RPW + half fills this FIFO 1 buffer
                if (AS_NBUF == 1) __syncthreads();  // the coder has taken the previous tile out of FIFO 1
                u32 w0[AS_TILE], w1[AS_TILE], T[AS_TILE];
#pragma unroll
                for (u32 j = 0; j < AS_TILE; ++j) {
                    const u32 q = AS_TILE * half + j;
                    const u32 rowaddr = ctx * AS_ROW_BYTES + lane4;
                    // freqs_current lookup (:118): X[s], X[s + 1] out of words s / 2 and s / 2 + 1, the row total
                    const u32 wa = rowaddr + (s[q] >> 1) * AS_PLANE;
                    w0[j] = *reinterpret_cast<const u32_lds *>(lds + wa);
                    w1[j] = *reinterpret_cast<const u32_lds *>(lds + wa + AS_PLANE);
                    T[j] = *reinterpret_cast<const u16_lds *>(lds + rowaddr);  // count 0 of the row = its total
                    // update_model: X[j] += 1 for j > s, total += 1 -- no register round trip
                    {
                        u32 *row = reinterpret_cast<u32 *>(lds + rowaddr);
                        const u32 add[8] = {la[q].x, la[q].y, la[q].z, la[q].w, lb[q].x, lb[q].y, lb[q].z, lb[q].w};
#pragma unroll
                        for (u32 p = 0; p < 8; ++p)
                            __hip_atomic_fetch_add(row + p * (AS_PLANE / 4), add[p], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    ctx = as_next_ctx<ORDER1>(P, ctx, s[q]);
                }
#pragma unroll
                for (u32 j = 0; j < AS_TILE; ++j) {
                    const u32 q = AS_TILE * half + j;
                    u32 cd = __builtin_amdgcn_alignbit(w1[j], w0[j], 16 * (s[q] & 1));
                    if (s[q] == 0) cd &= 0xFFFF0000u;                   // X[0] = 0 (its cell holds the total)
                    if (s[q] == 15) cd = (cd & 0xFFFFu) | (T[j] << 16);  // X[16] is the total
                    const double x = af_recip((double)T[j]);
                    *reinterpret_cast<double *>(lds + AS_F1X_BASE + (buf * AS_TILE + j) * AS_F1X_SLOT + lane * 8) = x;
                    *reinterpret_cast<u32_lds *>(lds + AS_F1C_BASE + (buf * AS_TILE + j) * AS_F1C_SLOT + lane4) = cd;
                }
                __syncthreads();
            }
        }
        // the chunk's status word is written by the writer lane: hand this side's bits over (FIFO 1 is idle now)
        *reinterpret_cast<u32_lds *>(lds + AS_F1C_BASE + lane4) = st;
        __syncthreads();
        return;
    }

    if (role == 1) {
        // =============================== coder role ===================================================================
        // owns (low, high - 1, pending): shrink_range (:58-78), then how many E1/E2 steps (k) and E3 steps (m) the
        // renormalisation loops (:126-150) take -- closed form, literal loops on the strict-comparison corners -- and
        // hands the writer what those steps emit: the top k bits of low and the pending count they meet
        u32 low = 0, hm = 0xFFFFFFFFu;
        u32 pending = 0;
        for (u32 r = 0; r < n_rounds; ++r) {
            const u32 buf1 = (AS_NBUF == 2) ? ((r + 1) & 1) : 0, buf2 = buf1;  // tile r - 1
            double xt[AS_TILE];
            u32 cdt[AS_TILE];
            if (AS_NBUF == 1) {  // the whole tile into registers, then the buffers are free for the next one
#pragma unroll
                for (u32 j = 0; j < AS_TILE; ++j) {
                    xt[j] = *reinterpret_cast<const double *>(lds + AS_F1X_BASE + j * AS_F1X_SLOT + lane * 8);
                    cdt[j] = *reinterpret_cast<const u32_lds *>(lds + AS_F1C_BASE + j * AS_F1C_SLOT + lane * 4);
                }
                __syncthreads();
            }
#pragma unroll
            for (u32 j = 0; j < AS_TILE; ++j) {
                const u32 i = (r - 1) * AS_TILE + j;
                if (r > 0 && i < n) {
                    const double x = (AS_NBUF == 1) ? xt[j] : *reinterpret_cast<const double *>(lds + AS_F1X_BASE + (buf1 * AS_TILE + j) * AS_F1X_SLOT + lane * 8);
                    const u32 cd = (AS_NBUF == 1) ? cdt[j] : *reinterpret_cast<const u32_lds *>(lds + AS_F1C_BASE + (buf1 * AS_TILE + j) * AS_F1C_SLOT + lane * 4);
                    const double rd = (double)(hm - low) + 1.0;
                    const u32 q1 = (u32)(__builtin_fma(rd, (double)(cd & 0xFFFFu), 0.5) * x);
                    const u32 q2m1 = (u32)__builtin_fma(__builtin_fma(rd, (double)(cd >> 16), 0.5), x, -1.0);
                    hm = low + q2m1;
                    low = low + q1;
                    // k = clz(low ^ hm) E1/E2 steps, then m = leading (1,0) pairs of (low, hm) below them E3 steps.  The
                    // reference's STRICT comparisons (quirk Q1) differ from this only when low or high hits a power-of-two
                    // boundary inside the shifted-out prefix: low != 0 but all its bits leave (the new low is 0), or
                    // high != 2^32 but high << (k + m + 1) == 0 (the new high - 1 is all ones) -- see scl_aec_math.h
                    u32 k = (u32)__builtin_clz(low ^ hm);  // low != hm: the interval holds more than one value
                    const u32 z = ((low & ~hm) << k) << 1;
                    u32 m = (u32)__builtin_clz(~z);
                    const u32 kt = k + m;  // <= 31
                    const u32 nlow = (low << kt) & 0x7FFFFFFFu;
                    const u32 nhm = ~(~hm << kt) | AF_HALF;  // (hm << kt) | ones(kt) | HALF
                    // the conservative form of the corner test (af_renorm2_dec): the first symbols of a chunk, coded while
                    // low is still 0 or high still 2^32, take the literal loops as well -- two compares fewer on all others
                    const bool edge = nlow == 0 || nhm == 0xFFFFFFFFu;
                    u32 top;
                    if (__builtin_expect(edge, 0)) {
                        u64 lo = low, hi = (u64)hm + 1;
                        k = 0;
                        m = 0;
                        while (hi < AF_HALF || lo > AF_HALF) {
                            if (hi < AF_HALF) {
                                lo <<= 1;
                                hi <<= 1;
                            } else {
                                lo = (lo - AF_HALF) << 1;
                                hi = (hi - AF_HALF) << 1;
                            }
                            k += 1;
                        }
                        while (lo > AF_QTR && hi < 3ull * AF_QTR) {
                            m += 1;
                            lo = (lo - AF_QTR) << 1;
                            hi = (hi - AF_QTR) << 1;
                        }
                        top = k ? low >> (32 - k) : 0u;  // every step emitted the leading bit of lo
                        low = (u32)lo;
                        hm = (u32)(hi - 1);
                    } else {
                        top = low >> ((32 - k) & 31);  // k = 0: never looked at
                        low = nlow;
                        hm = nhm;
                    }
                    *reinterpret_cast<uint2 *>(lds + AS_F2_BASE + (buf2 * AS_TILE + j) * AS_F2_SLOT + lane * 8) =
                        make_uint2(top, k | (pending << 8));
                    pending = (k ? 0u : pending) + m;
                }
            }
            __syncthreads();
        }
        // termination (:153-159) is the writer's: it needs the final low and pending count
        *reinterpret_cast<uint2 *>(lds + AS_F2_BASE + lane * 8) = make_uint2(low, pending);
        __syncthreads();
        return;
    }

    // =================================== writer role =====================================================================
    // 64-bit window hi:lo, `cnt` pending bits bottom-aligned (cnt < 32 between symbols); a field is at most 31 bits, so
    // the window never holds more than 62.  32-bit shifts only (see AnsFwdWriter::put, scl_ans_fast_io.h).  A completed
    // big-endian word goes straight to the slot (4-byte stores; the stream is a third of the input and L2 merges them).
    // Round 6 staged the words in LDS (the 2 KiB the totals vacated) and stored whole 32-byte sectors: the same streams,
    // but the writer role -- one of three that pace each other -- grew by a tenth and the kernel went from 4.7 to 5.05 ms
    // (5.4 with the sector's store deferred by a round): not kept.  The 3 x HBM traffic of these word stores (sectors
    // written back partly filled) is 0.94 TB/s and bounds nothing here.
    u32 *dst = reinterpret_cast<u32 *>(out + (live ? chunk : 0) * out_stride);
    u32 whi = 0, wlo = 0, cnt = 0, nwords = 0;
    auto push = [&](u32 v, u32 nb) {  // v < 2^nb, nb <= 31; nb = 0 pushes nothing
        whi = (whi << nb) | ((wlo >> 1) >> (31 - nb));
        wlo = (wlo << nb) | v;
        cnt += nb;
        if (cnt >= 32) {
            cnt -= 32;
            const u32 word = cnt ? __builtin_amdgcn_alignbit(whi, wlo, cnt) : wlo;
            dst[nwords++] = __builtin_bswap32(word);
        }
    };
    auto push_run = [&](u32 bit, u32 count) {
        while (count >= 31) {
            push(bit ? 0x7FFFFFFFu : 0u, 31);
            count -= 31;
        }
        if (count) push(bit ? ((1u << count) - 1u) : 0u, count);
    };
    u32 st = (P.size_bits < 32 && (n >> P.size_bits)) ? SCL_ST_SIZE : 0u;
    if (live) {  // header, :92-99
        const u32 hv = P.size_bits < 32 ? (n & ((1u << P.size_bits) - 1u)) : n;
        if (P.size_bits == 32) {
            push(hv >> 16, 16);
            push(hv & 0xFFFFu, 16);
        } else {
            push(hv, P.size_bits);
        }
    }
    for (u32 r = 0; r < n_rounds; ++r) {
        const u32 buf2 = (AS_NBUF == 2) ? (r & 1) : 0;  // tile r - 2
        uint2 et[AS_TILE];
        if (AS_NBUF == 1) {
#pragma unroll
            for (u32 j = 0; j < AS_TILE; ++j)
                et[j] = *reinterpret_cast<const uint2 *>(lds + AS_F2_BASE + j * AS_F2_SLOT + lane * 8);
            __syncthreads();
        }
#pragma unroll
        for (u32 j = 0; j < AS_TILE; ++j) {
            const u32 i = (r - 2) * AS_TILE + j;
            if (r > 1 && i < n) {
                const uint2 e = (AS_NBUF == 1) ? et[j] : *reinterpret_cast<const uint2 *>(lds + AS_F2_BASE + (buf2 * AS_TILE + j) * AS_F2_SLOT + lane * 8);
                const u32 k = e.y & 0xFFu, pend = e.y >> 8, top = e.x;
                // the k E1/E2 steps emit b0, then `pend` copies of !b0, then the other k - 1 bits of top
                const u32 km1 = (k - 1) & 31;
                const u32 b0 = top >> km1;
                const u32 rest = top & ((1u << km1) - 1u);
                if (__builtin_expect(k != 0 && k + pend > 31, 0)) {
                    push(b0, 1);
                    push_run(b0 ^ 1u, pend);
                    push(rest, k - 1);
                } else {
                    const u32 pat = (1u << (pend & 31)) - (b0 ^ 1u);
                    const u32 v = (pat << km1) | rest;
                    push(k ? v : 0u, k ? k + pend : 0u);
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();  // the model's status bits and the coder's final state are in place (both FIFOs are idle now)
    st |= *reinterpret_cast<const u32_lds *>(lds + AS_F1C_BASE + lane * 4);
    const uint2 fin = *reinterpret_cast<const uint2 *>(lds + AS_F2_BASE + lane * 8);
    if (!live) return;
    const u32 pending = fin.y + 1;  // termination, :153-159
    if (fin.x <= AF_QTR) {
        push(0, 1);
        push_run(1, pending);
    } else {
        push(1, 1);
        push_run(0, pending);
    }
    const u64 total = (u64)nwords * 32 + cnt;
    if (cnt) dst[nwords] = __builtin_bswap32(wlo << (32 - cnt));
    out_bit_off[chunk] = chunk * out_stride * 8;
    out_nbits[chunk] = (u32)total;
    if (status) status[chunk] = st;
}

// ---- host side ----------------------------------------------------------------------------------------------
static AecSplitDev aec_split_dev(const scl_aec_model *m) {
    AecSplitDev f;
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

int aec_split_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                            u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                            u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AS_LANES - 1) / AS_LANES);
    if (m->dev.k == 1)
        hipLaunchKernelGGL(aec_split_encode_kernel<true>, dim3(blocks), dim3(AS_THREADS), 0, st, aec_split_dev(m), d_sym,
                           sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits,
                           d_status);
    else
        hipLaunchKernelGGL(aec_split_encode_kernel<false>, dim3(blocks), dim3(AS_THREADS), 0, st, aec_split_dev(m), d_sym,
                           sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits,
                           d_status);
    return SCL_OK;
}
