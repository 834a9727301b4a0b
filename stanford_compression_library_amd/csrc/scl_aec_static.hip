// scl_aec_static.hip -- arithmetic coding with a STATIC frequency model (FixedFreqModel) for gfx950, one wavefront
// lane per chunk.  Same streams, bit for bit, as scl_aec.hip and the reference:
//   ArithmeticEncoder.shrink_range / encode_block   scl/compressors/arithmetic_coding.py:58-78, :80-161
//   ArithmeticDecoder.decode_step_core / decode_block                               :177-201, :203-287
//   FixedFreqModel                                   scl/compressors/probability_models.py:57-67
//
// Served models (aec_static_ok): PRECISION = 32, alphabet 2..256, total <= 2^16.
// The model is read-only, so nothing is private to a lane but its interval: the {c, c + f} table (and, for totals
// up to 4096, a slot -> symbol table for the decoder) sits in LDS once per workgroup, and the kernels run at the
// occupancy of the rANS fast kernels with their line-granular I/O (scl_ans_fast_io.h): symbols arrive as whole
// 128-byte lines, stream words pass through a per-lane LDS ring and leave / arrive as whole lines.
// Arithmetic as in scl_aec_fast.hip (scl_aec_math.h): (rng*c)//T in exact binary64 with ONE reciprocal of T per
// launch, closed-form renormalisation, literal loops of the reference on its strict-comparison corners.
#include "scl_aec_internal.h"
#include "scl_aec_math.h"
#include "scl_ans_fast_io.h"

#define AS_THREADS 256
#define AS_RING_BYTES (32 * AS_THREADS * 4)
#define AS_TAB_BASE AS_RING_BYTES         // uint2 {c, c + f} per symbol
#define AS_LUT_BASE (AS_TAB_BASE + 2048)  // u8 symbol per slot (totals <= 4096)
#define AS_LDS_BYTES (AS_LUT_BASE + 4096)

struct AecStaticDev {
    u32 K, T;
    u32 t;          // log2(T) when T is a power of two (the POW2 kernels), else unused
    u32 size_bits;  // DATA_BLOCK_SIZE_BITS (1..32)
    const u32 *d_freq, *d_cum;
};

__device__ __forceinline__ void as_setup_tables(char *lds, const AecStaticDev &P, u32 tid, bool with_lut) {
    if (tid < P.K) {
        const u32 c = P.d_cum[tid], f = P.d_freq[tid];
        *reinterpret_cast<uint2 *>(lds + AS_TAB_BASE + tid * 8) = make_uint2(c, c + f);
        if (with_lut)
            for (u32 j = 0; j < f; ++j) *reinterpret_cast<u8 *>(lds + AS_LUT_BASE + c + j) = (u8)tid;
    } else if (tid < 256) {
        *reinterpret_cast<uint2 *>(lds + AS_TAB_BASE + tid * 8) = make_uint2(0, 1);  // never selected
    }
    __syncthreads();
}

// The ENCODER's table for power-of-two totals: 16-byte entries {c, 0, d, 0} -- both counts arrive zero-extended to 64 bits in
// aligned register pairs, which is how v_mad_u64_u32 (af_shrink_pow2_wide) takes its addend: four register copies per symbol
// otherwise.  (36 KiB of LDS per workgroup: still four per CU.  The decoder has no room for it next to its slot table.)
__device__ __forceinline__ void as_setup_table16(char *lds, const AecStaticDev &P, u32 tid) {
    if (tid < P.K) {
        const u32 c = P.d_cum[tid], f = P.d_freq[tid];
        *reinterpret_cast<uint4 *>(lds + AS_TAB_BASE + tid * 16) = make_uint4(c, 0u, c + f, 0u);
    } else if (tid < 256) {
        *reinterpret_cast<uint4 *>(lds + AS_TAB_BASE + tid * 16) = make_uint4(0u, 0u, 1u, 0u);  // never selected
    }
    __syncthreads();
}
typedef AnsFwdWriter<AS_THREADS> AsOut;
typedef AnsBitReader<AS_THREADS, true> AsIn;

// one symbol of the encoder: shrink_range, then the renormalisation loops (:126-150)
template <bool POW2>
__device__ __forceinline__ void as_encode_symbol(u32 &low, u32 &hm, u32 &pending, u64 c, u64 d, u32 t, double xT,
                                                 AsOut &wr, char *lds) {
    if (POW2)
        af_shrink_pow2_wide(low, hm, c, d, t);
    else
        af_shrink2(low, hm, (u32)c, (u32)d, xT);
    u32 k, m, nlow, nhm;
    const bool edge = af_renorm2_dec(low, hm, k, m, nlow, nhm);  // the conservative corner test: two compares fewer
    const bool rare = edge | (k + pending > 32);                 // one condition, one branch
    if (__builtin_expect(rare, 0)) {
        u64 lo = low, hi = (u64)hm + 1;
        while (hi < AF_HALF || lo > AF_HALF) {
            if (hi < AF_HALF) {
                wr.put(lds, 0, 1);
                wr.put_run(lds, 1, pending);
                lo <<= 1;
                hi <<= 1;
            } else {
                wr.put(lds, 1, 1);
                wr.put_run(lds, 0, pending);
                lo = (lo - AF_HALF) << 1;
                hi = (hi - AF_HALF) << 1;
            }
            wr.maybe_flush(lds);
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
        // b0, then `pending` copies of !b0, then the other k - 1 common bits -- without a branch on k: for k = 0 the field is
        // empty (v = 0, nb = 0) and the pending count just grows (as in scl_aec_iid.hip)
        const bool any = k != 0;
        const u32 km1 = (k - 1u) & 31u;
        const u32 b0 = low >> 31;
        const u32 rest = __builtin_amdgcn_ubfe(low, (32u - k) & 31u, km1);  // bits 30 .. 32-k of low
        const u32 pat = (1u << pending) - (b0 ^ 1u);                       // pending <= 31 here
        u32 fv = (pat << km1) | rest, fn = k + pending;
        asm volatile("" : "+v"(fv), "+v"(fn));  // computed for every lane: the compiler would turn the selects into a branch
        wr.put_field(lds, any ? fv : 0u, any ? fn : 0u);
        pending = (any ? 0u : pending) + m;
        low = nlow;
        hm = nhm;
    }
}

// POW2: the total is a power of two, shrink_range in integers (af_shrink_pow2)
template <bool POW2>
__global__ void __launch_bounds__(AS_THREADS, 4)
    aec_static_encode_kernel(AecStaticDev P, const u8 *__restrict__ sym, u64 sym_stride, const u32 *__restrict__ lens,
                             u32 chunk_len, u64 n_chunks, u8 *__restrict__ out, u64 out_stride,
                             u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits, u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[AS_TAB_BASE + (POW2 ? 4096 : 2048)];
    const u32 tid = threadIdx.x;
    if (POW2)
        as_setup_table16(lds, P, tid);
    else
        as_setup_tables(lds, P, tid, false);
    const u64 chunk = (u64)blockIdx.x * AS_THREADS + tid;
    if (chunk >= n_chunks) return;
    const u32 n = lens ? lens[chunk] : chunk_len;
    const u8 *src = sym + chunk * sym_stride;
    const double xT = af_recip((double)P.T);
    AsOut wr;
    wr.init(tid, out + chunk * out_stride);
    wr.put(lds, P.size_bits < 32 ? (n & ((1u << P.size_bits) - 1u)) : n, P.size_bits);  // header, :92-99
    u32 low = 0, hm = 0xFFFFFFFFu, pending = 0, bad = 0;

    auto code_word = [&](u32 w, u32 cnt) {  // up to four symbols, first symbol in the low byte
#pragma unroll 1
        for (u32 j = 0; j < cnt; ++j) {
            u32 s = w & 0xFFu;
            w >>= 8;
            bad = max(bad, s);
            s = (s < P.K) ? s : 0u;
            if (POW2) {
                const uint4 e = *reinterpret_cast<const uint4 *>(lds + AS_TAB_BASE + s * 16);  // {c, 0, d, 0}
                as_encode_symbol<POW2>(low, hm, pending, e.x | ((u64)e.y << 32), e.z | ((u64)e.w << 32), P.t, xT, wr, lds);
            } else {
                const uint2 e = *reinterpret_cast<const uint2 *>(lds + AS_TAB_BASE + s * 8);
                as_encode_symbol<POW2>(low, hm, pending, e.x, e.y, P.t, xT, wr, lds);
            }
        }
        wr.maybe_flush(lds);  // <= 4 new words per fast-path call on top of <= 15 pending (ring of 32)
    };

    const u32 n_lines = n >> 7;
    const uint4 *src16 = reinterpret_cast<const uint4 *>(src);
    Line128 cur;  // no second buffer: with it the kernel needs more than its 128 registers (4 waves per SIMD hide the load)
#pragma nounroll
    for (u32 t = 0; t < n_lines; ++t) {
        cur.load(src16 + 8 * t);
#pragma unroll 1
        for (u32 q = 0; q < 8; ++q) {
            const uint4 v = cur.v[0];
#pragma unroll
            for (int i = 0; i < 7; ++i) cur.v[i] = cur.v[i + 1];
            code_word(v.x, 4);
            code_word(v.y, 4);
            code_word(v.z, 4);
            code_word(v.w, 4);
        }
    }
    u32 i = n_lines << 7;
    for (; i + 4 <= n; i += 4) code_word(*reinterpret_cast<const u32 *>(src + i), 4);  // ragged tail
    if (i < n) {
        u32 w = 0;
        for (u32 j = 0; i + j < n; ++j) w |= (u32)src[i + j] << (8 * j);
        code_word(w, n - i);
    }
    pending += 1;  // termination, :153-159
    if (low <= AF_QTR) {
        wr.put(lds, 0, 1);
        wr.put_run(lds, 1, pending);
    } else {
        wr.put(lds, 1, 1);
        wr.put_run(lds, 0, pending);
    }
    const u64 total = wr.finish(lds);
    out_bit_off[chunk] = chunk * out_stride * 8;
    out_nbits[chunk] = (u32)total;
    if (status)
        status[chunk] = ((bad >= P.K) ? SCL_ST_SYMBOL : 0u) | ((P.size_bits < 32 && (n >> P.size_bits)) ? SCL_ST_SIZE : 0u);
}

// LUT = true: total <= 4096, the decoder's search is one byte read by target slot; else a binary search on c
template <bool LUT, bool POW2>
__global__ void __launch_bounds__(AS_THREADS, 4)
    aec_static_decode_kernel(AecStaticDev P, const u8 *__restrict__ in, u64 in_size_bytes,
                             const u64 *__restrict__ bit_off, const u32 *__restrict__ in_nbits, u64 n_chunks,
                             u8 *__restrict__ out_sym, u64 out_stride, u32 out_cap, u32 *__restrict__ out_lens,
                             u32 *__restrict__ consumed, u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char lds[LUT ? AS_LDS_BYTES : AS_TAB_BASE + 2048];
    const u32 tid = threadIdx.x;
    as_setup_tables(lds, P, tid, LUT);
    const u64 chunk = (u64)blockIdx.x * AS_THREADS + tid;
    if (chunk >= n_chunks) return;
    const u32 nbits = in_nbits[chunk];
    u32 st = 0;
    if (nbits < P.size_bits) {
        out_lens[chunk] = 0;
        consumed[chunk] = 0;
        if (status) status[chunk] = SCL_ST_TRUNCATED;
        return;
    }
    AsIn rd;
    rd.init(in, in_size_bytes, bit_off[chunk], lds, tid, nbits);
    u32 n = rd.get(lds, P.size_bits);
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
    rd.maybe_refill(lds);
    const double xT = af_recip((double)P.T), Td = (double)P.T;
    u8 *dst = out_sym + chunk * out_stride;
    u32 used = 32;
    u32 state = rd.get(lds, 32);
    u32 low = 0, hm = 0xFFFFFFFFu;

    // one symbol: decode_step_core (:177-201) and, unless it is the last one, the renormalisation (:245-275)
    auto decode_symbol = [&](bool last) -> u32 {
        const double xr = af_recip((double)(hm - low) + 1.0);
        const double num = __builtin_fma((double)(state - low) + 1.0, Td, -0.5);
        // ((state - low + 1) * T - 1) // rng, see scl_aec.hip.  low <= state <= hm holds for ANY input bits (the symbol chosen
        // is the one whose interval holds the state), so target <= T - 1: no clamp
        const u32 tgt = (u32)(num * xr);
        u32 s;
        if (LUT) {
            s = *reinterpret_cast<const u8 *>(lds + AS_LUT_BASE + tgt);
        } else {  // largest s with c[s] <= tgt
            s = 0;
#pragma unroll
            for (u32 b = 128; b > 0; b >>= 1) {
                const u32 t = s + b;
                const u32 ct = *reinterpret_cast<const u32 *>(lds + AS_TAB_BASE + min(t, 255u) * 8);
                s = (t < P.K && ct <= tgt) ? t : s;
            }
        }
        const uint2 e = *reinterpret_cast<const uint2 *>(lds + AS_TAB_BASE + s * 8);
        if (POW2)
            af_shrink_pow2(low, hm, e.x, e.y, P.t);
        else
            af_shrink2(low, hm, e.x, e.y, xT);
        if (last) return s;
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
                stt += rd.get(lds, 1);
                used++;
            }
            while (lo > AF_QTR && hi < 3ull * AF_QTR) {
                lo = (lo - AF_QTR) << 1;
                hi = (hi - AF_QTR) << 1;
                stt = (stt - AF_QTR) << 1;
                stt += rd.get(lds, 1);
                used++;
            }
            rd.maybe_refill(lds);
            low = (u32)lo;
            hm = (u32)(hi - 1);
            state = (u32)stt;
        } else {
            const u32 kt = k + m;  // <= 31
            // kt bits come in from the stream: {state, look} << kt in one 64-bit shift (kt may be 0)
            const u64 both = (((u64)state << 32) | rd.look()) << kt;
            const u32 keep = (state << k) & AF_HALF;
            rd.advance(lds, kt);
            state = ((u32)(both >> 32) & 0x7FFFFFFFu) | keep;
            low = nlow;
            hm = nhm;
            used += kt;
        }
        return s;
    };

    // whole 128-byte lines of output (eight registers, one burst), then the ragged tail byte by byte
    u32 i = 0;
    const u32 n_full = n & ~127u;
#pragma nounroll
    for (; i < n_full; i += 128) {
        uint4 a[8];
#pragma unroll 1
        for (u32 q = 0; q < 8; ++q) {
            u32 w4[4];
#pragma unroll 1
            for (u32 g = 0; g < 4; ++g) {
                u32 w = 0;
#pragma unroll 1
                for (u32 j = 0; j < 4; ++j) {
                    const bool last = (i + 16 * q + 4 * g + j + 1 == n);
                    w |= decode_symbol(last) << (8 * j);
                }
                rd.maybe_refill(lds);  // <= 4 words consumed by four fast-path symbols
                // rotate instead of indexing: w4[g] with a run-time g would go through scratch
                w4[0] = w4[1], w4[1] = w4[2], w4[2] = w4[3], w4[3] = w;
            }
#pragma unroll
            for (int r = 0; r < 7; ++r) a[r] = a[r + 1];
            a[7] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        uint4 *p = reinterpret_cast<uint4 *>(dst + i);
#pragma unroll
        for (int b = 0; b < 8; ++b) p[b] = a[b];
    }
    for (; i < n; ++i) {
        dst[i] = (u8)decode_symbol(i + 1 == n);
        if ((i & 3u) == 3u) rd.maybe_refill(lds);
    }
    // how many of the last PRECISION bits belonged to the encoder (:277-282)
    const u64 lo = low, hi = (u64)hm + 1;
    u32 e = 0;
    for (; e < 32; ++e) {
        const u64 slo = ((u64)state >> e) << e, shi = slo + (1ull << e);
        if (slo < lo || shi > hi) break;
    }
    if (e == 32) e = 31;
    consumed[chunk] = (u32)((i64)((u64)used + P.size_bits) - ((i64)e - 1));
    if (status) status[chunk] = st;
}

// ---- host side ----------------------------------------------------------------------------------------------
bool aec_static_ok(const scl_aec_model *m) {
    const AecDev &d = m->dev;
    return d.kind == SCL_MODEL_FIXED && d.K >= 2 && d.K <= 256 && d.P == 32 &&
           d.total0 <= 65536 && (u64)d.total0 < d.max_total;
}

static AecStaticDev aec_static_dev(const scl_aec_model *m) {
    AecStaticDev f;
    f.K = m->dev.K;
    f.T = m->dev.total0;
    f.t = (f.T & (f.T - 1)) == 0 ? scl_bit_width_u64(f.T) - 1 : 0xFFFFFFFFu;
    f.size_bits = m->dev.size_bits;
    f.d_freq = m->dev.d_freq;
    f.d_cum = m->dev.d_cum;
    return f;
}

void aec_static_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                              u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset,
                              u32 *d_out_nbits, u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AS_THREADS - 1) / AS_THREADS);
    const AecStaticDev dev = aec_static_dev(m);
    if (dev.t != 0xFFFFFFFFu)
        hipLaunchKernelGGL(aec_static_encode_kernel<true>, dim3(blocks), dim3(AS_THREADS), 0, st, dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status);
    else
        hipLaunchKernelGGL(aec_static_encode_kernel<false>, dim3(blocks), dim3(AS_THREADS), 0, st, dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status);
}

void aec_static_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + AS_THREADS - 1) / AS_THREADS);
    const AecStaticDev dev = aec_static_dev(m);
    const bool pow2 = dev.t != 0xFFFFFFFFu;
#define AS_LAUNCH_DEC(LUT, POW2)                                                                                         \
    hipLaunchKernelGGL((aec_static_decode_kernel<LUT, POW2>), dim3(blocks), dim3(AS_THREADS), 0, st, dev, d_in,          \
                       in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,   \
                       d_consumed, d_status)
    if (m->dev.total0 <= 4096) {
        if (pow2)
            AS_LAUNCH_DEC(true, true);
        else
            AS_LAUNCH_DEC(true, false);
    } else {
        if (pow2)
            AS_LAUNCH_DEC(false, true);
        else
            AS_LAUNCH_DEC(false, false);
    }
#undef AS_LAUNCH_DEC
}
