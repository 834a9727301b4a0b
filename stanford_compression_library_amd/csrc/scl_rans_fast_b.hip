// scl_rans_fast_b.hip -- gfx950 fast path of batched rANS with NUM_BITS_OUT = b > 1 (the reference's own parameter
// sweep uses NUM_BITS_OUT = 8, rANS.py:366-379; SURVEY 8d names b = 8, RANGE_FACTOR = 2^8 as the second point of
// configs[1]).  Same bit stream as the generic kernels in scl_rans.hip (reference rANS.py:186-210 / :270-297), same
// line-granular I/O as scl_rans_fast.hip (AnsBackWriter / AnsBitReader / CoopLineStore, scl_ans_fast_io.h).
// Served: b in {2, 4, 8, 16}, total M a power of two <= 4096, RANGE_FACTOR a power of two, H = L 2^b - 1 < 2^31,
// at most 31 bits per symbol.  Everything else with b > 1 stays on the generic kernels.
//
//  encode, per symbol (shrink_state :149-161 releases groups of b bits until x <= max_shrunk_state = RF f 2^b - 1):
//    x in [L, L 2^b) needs k_lo groups at x = L and at most k_lo + 1 at x = H, the switch being at
//    thresh = (RF f 2^b) << (b k_lo):   s = b k1 - b [x < thresh],  k1 = k_lo + 1   (bits released)
//    field = low s bits of x;  xs = x >> s;  q = xs // f = trunc((xs + 0.5) * (1/f)) in binary64 (xs < 2^31: the
//    product is off by < 2^-21, (xs + 0.5)/f is >= 2^-13 away from an integer);  x = xs + c + q (M - f)  (:138-147)
//    -- the 32-bit multiply-high of the b = 1 kernels needs NUM_STATE_BITS + 2b < 31 and cannot serve b = 8.
//  decode, per symbol: x = (x >> m) f + (slot - c)  (:234-249); expand_state (:251-260) reads groups of b bits while
//    x < L = 2^(r+m):  d = bit_width(L) - bit_width(x) missing bits -> ceil(d / b) groups, i.e.
//    sh = (max(d, 0) + b - 1) & ~(b - 1) bits in one 64-bit shift.
#include <vector>

#include "scl_ans_fast_io.h"
#include "scl_rans_internal.h"

#define RB_THREADS 256
#define RB_RING_BYTES (32 * RB_THREADS * 4)
typedef AnsBackWriter<RB_THREADS> EncOutB;
#define RBD_THREADS 1024      // batches that fill the chip with one 1024-lane workgroup per CU
#define RBD_THREADS_SMALL 256  // smaller ones spread over all CUs instead (see scl_rans_fast.hip)

// ---------------------------------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------------------------------
// LDS: [0, 32 KiB) word ring, then 256 entries {1/f (binary64), thresh, c}, then 256 words (M - f) | b k1 << 24
__device__ __forceinline__ void rb_encode_symbol(u32 &x, u32 sym, EncOutB &o, char *lds, const char *tabA,
                                                 const char *tabB, u32 b) {
    const uint4 e = *reinterpret_cast<const uint4 *>(tabA + sym * 16);
    const u32 a = *reinterpret_cast<const u32 *>(tabB + sym * 4);
    const u32 neg = (x - e.z) >> 31;           // 1 iff x < thresh (both < 2^31)
    const u32 s = (a >> 24) - (b & (0u - neg));
    const u32 bits = __builtin_amdgcn_ubfe(x, 0, s);
    const u32 xs = x >> s;
    const double inv_f = __hiloint2double((int)e.y, (int)e.x);
    const u32 q = (u32)(((double)xs + 0.5) * inv_f);
    x = __umul24(q, a) + xs + e.w;  // v_mad_u32_u24 reads only the low 24 bits of a
    o.put(lds, bits, s);            // s <= 31 (rans_fastb_build_tables)
}

template <bool CHECK_SYM>
__device__ __forceinline__ void rb_encode16(const uint4 v, u32 &x, EncOutB &o, u32 &smax, char *lds, const char *tabA,
                                            const char *tabB, u32 b) {
    const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 s = (wv[d] >> (8 * j)) & 0xFFu;
            if (CHECK_SYM) smax = max(smax, s);
            rb_encode_symbol(x, s, o, lds, tabA, tabB, b);
        }
    }
    o.maybe_flush(lds);  // every 16 symbols: <= 16 new words on top of <= 15 pending, ring of 32
}

template <bool CHECK_SYM>
__global__ void __launch_bounds__(RB_THREADS, 4) rans_encode_fastb_kernel(RansFastBDev P, const u8 *__restrict__ sym,
                                                                         u64 sym_stride,
                                                                         const u32 *__restrict__ lens, u32 chunk_len,
                                                                         u64 n_chunks, u8 *__restrict__ out,
                                                                         u64 out_stride, u64 *__restrict__ out_bit_off,
                                                                         u32 *__restrict__ out_nbits,
                                                                         u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char s_lds[RB_RING_BYTES + 256 * 16 + 256 * 4];
    char *lds = s_lds;
    const char *tabA = s_lds + RB_RING_BYTES;
    const char *tabB = s_lds + RB_RING_BYTES + 256 * 16;
    reinterpret_cast<uint4 *>(s_lds + RB_RING_BYTES)[threadIdx.x & 255] = P.d_enc[threadIdx.x & 255];
    reinterpret_cast<u32 *>(s_lds + RB_RING_BYTES + 256 * 16)[threadIdx.x & 255] = P.d_aux[threadIdx.x & 255];
    __syncthreads();
    const u64 c = (u64)blockIdx.x * RB_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 n = lens ? lens[c] : chunk_len;
    const u8 *src = sym + c * sym_stride;
    const u32 b = P.b;
    EncOutB o;
    o.init(threadIdx.x, out + (c + 1) * out_stride);
    u32 x = P.L;
    u32 smax = 0;

    const u32 n_lines = n >> 7;
    const uint4 *src16 = reinterpret_cast<const uint4 *>(src);
    Line128 cur;  // one line in registers; 4 waves per SIMD hide the load
#pragma nounroll
    for (u32 t = 0; t < n_lines; ++t) {
        cur.load(src16 + 8 * t);
#pragma nounroll
        for (int q = 0; q < 4; ++q) {
            rb_encode16<CHECK_SYM>(cur.v[0], x, o, smax, lds, tabA, tabB, b);
            rb_encode16<CHECK_SYM>(cur.v[1], x, o, smax, lds, tabA, tabB, b);
#pragma unroll
            for (int i = 0; i < 6; ++i) cur.v[i] = cur.v[i + 2];
        }
    }
    u32 i = n_lines << 7;
    for (; i + 16 <= n; i += 16)  // ragged tail: whole 16-byte blocks, then single symbols
        rb_encode16<CHECK_SYM>(*reinterpret_cast<const uint4 *>(src + i), x, o, smax, lds, tabA, tabB, b);
    for (; i < n; ++i) {
        const u32 s = src[i];
        if (CHECK_SYM) smax = max(smax, s);
        rb_encode_symbol(x, s, o, lds, tabA, tabB, b);
    }
    o.maybe_flush(lds);
    o.put32(lds, x, P.nsb);
    u32 st = (CHECK_SYM && smax >= P.K) ? SCL_ST_SYMBOL : 0u;
    if (P.size_bits < 32 && (n >> P.size_bits)) st |= SCL_ST_SIZE;
    o.put32(lds, n, P.size_bits);
    const u64 total = o.finish(lds);
    out_bit_off[c] = (c + 1) * out_stride * 8 - total;
    out_nbits[c] = (u32)total;
    if (status) status[c] = st;
}

// ---------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------
// One workgroup of 1024 lanes per CU: slot table {f | sym << 24, slot - c} at LDS offset 0 (32 KiB), word ring behind it.
template <typename DecInB>
__device__ __forceinline__ u32 rb_decode_symbol(u32 &x, DecInB &r, char *lds, const char *tab, u32 m_log2, u32 cbl,
                                                u32 bm1) {
    const uint2 e = *reinterpret_cast<const uint2 *>(tab + ((x << 3) & (((1u << m_log2) - 1u) << 3)));
    const u32 xn = __umul24(x >> m_log2, e.x) + e.y;  // v_mad_u32_u24 reads the low 24 bits (f) of e.x
    // missing bits d = bit_width(L) - bit_width(xn) = clz(xn) - cbl; groups of b bits: sh = b ceil(max(d, 0) / b)
    const int d = (int)__builtin_clz(xn) - (int)cbl;
    const u32 sh = ((u32)max(d, 0) + bm1) & ~bm1;
    x = (u32)(((((u64)xn) << 32) | r.look()) << sh >> 32);  // sh <= 27: one 64-bit shift, sh = 0 included
    r.advance(lds, sh);
    return e.x;
}

template <typename DecInB>
__device__ __forceinline__ uint4 rb_decode16(u32 &x, DecInB &r, char *lds, const char *tab, u32 m_log2, u32 cbl,
                                             u32 bm1) {
    u32 ow[4];
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        u32 o = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const u32 e = rb_decode_symbol(x, r, lds, tab, m_log2, cbl, bm1);
            o = __builtin_amdgcn_perm(o, e, 0x06050403u);  // o = (o << 8) | (e >> 24)
        }
        r.maybe_refill(lds);  // <= 4 words consumed by four symbols
        asm volatile("" : "+v"(o) : : "memory");
        ow[d] = o;
    }
    return make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) rans_decode_fastb_kernel(RansFastBDev P, const u8 *__restrict__ in,
                                                                       u64 in_size_bytes,
                                                                       const u64 *__restrict__ bit_off,
                                                                       const u32 *__restrict__ in_nbits, u64 n_chunks,
                                                                       u8 *__restrict__ out_sym, u64 out_stride,
                                                                       u32 out_cap, u32 *__restrict__ out_lens,
                                                                       u32 *__restrict__ consumed,
                                                                       u32 *__restrict__ status) {
    typedef AnsBitReader<THREADS> DecInB;
    __shared__ __attribute__((aligned(16))) char s_lds[4096 * 8 + DecInB::RING_BYTES];
    char *lds = s_lds + 4096 * 8;
    const char *tab = s_lds;
    for (u32 i = threadIdx.x; i < P.M; i += THREADS) reinterpret_cast<uint2 *>(s_lds)[i] = P.d_dec[i];
    __syncthreads();
    const u64 c = (u64)blockIdx.x * THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 avail = in_nbits[c];
    u32 st = 0;
    if (avail < P.size_bits + P.nsb) {  // header does not fit
        out_lens[c] = 0;
        consumed[c] = P.size_bits + P.nsb;
        if (status) status[c] = SCL_ST_TRUNCATED;
        return;
    }
    DecInB r;
    r.init(in, in_size_bytes, bit_off[c], lds, threadIdx.x);
    u32 n = r.get(lds, P.size_bits);
    u32 x = r.get(lds, P.nsb);
    out_lens[c] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    const u32 st_header = st;
    const u32 m_log2 = P.m_log2, cbl = P.cbl, bm1 = P.b - 1;
    u8 *dst = out_sym + c * out_stride;

    // symbols come out last-first (rANS.py:291): ragged head, 16-byte blocks up to a line boundary, then whole lines
    u32 i = n;
    while (i & 15u) {
        const u32 e = rb_decode_symbol(x, r, lds, tab, m_log2, cbl, bm1);
        dst[--i] = (u8)(e >> 24);
        if ((i & 3u) == 0) r.maybe_refill(lds);
    }
    while (i & 127u) {
        const uint4 v = rb_decode16(x, r, lds, tab, m_log2, cbl, bm1);
        i -= 16;
        *reinterpret_cast<uint4 *>(dst + i) = v;
    }
    CoopLineStore cs;  // whole waves of equally long chunks store cooperatively (scl_ans_fast_io.h)
    cs.init(out_sym, c, out_stride, i);
#pragma nounroll
    while (i) {
        uint4 a[8];
#pragma unroll
        for (int b = 7; b >= 0; --b) a[b] = rb_decode16(x, r, lds, tab, m_log2, cbl, bm1);
        i -= 128;
        if (cs.on) {
            cs.store(a, i);
        } else {
            uint4 *p = reinterpret_cast<uint4 *>(dst + i);
#pragma unroll
            for (int b = 0; b < 8; ++b) p[b] = a[b];
        }
    }
    const u32 used_bits = r.consumed();
    if (used_bits > avail) st |= SCL_ST_TRUNCATED;
    else if (st_header == 0 && x != P.L) st |= SCL_ST_STATE;  // assert state == INITIAL_STATE (rANS.py:295)
    consumed[c] = used_bits;
    if (status) status[c] = st;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// Decides whether the model qualifies and uploads the tables.  Returns SCL_OK also when the model simply does not
// qualify (m->fastb stays 0).
int rans_fastb_build_tables(scl_rans_model *m, const u32 *h_freq, const u32 *h_cum) {
    const RansDev &D = m->dev;
    m->fastb = 0;
    const u32 b = D.b;
    if (b != 2 && b != 4 && b != 8 && b != 16) return SCL_OK;
    if (D.m_log2 == 0xFFFFFFFFu || D.M < 2 || D.M > 4096 || D.K < 2) return SCL_OK;
    if ((D.RF & (D.RF - 1)) != 0 || (m->H >> 31) != 0 || m->max_bits_per_symbol > 31) return SCL_OK;
    if ((D.RF << b) > (1ull << 24)) return SCL_OK;  // x >> m and the quotient xs // f stay below 2^24 (v_mad_u32_u24)
    const u32 M = (u32)D.M;
    std::vector<uint4> enc(256);
    std::vector<u32> aux(256);
    std::vector<uint2> dec(M);
    for (u32 s = 0; s < 256; ++s) {
        const u32 src = s < D.K ? s : 0;  // out-of-alphabet symbols are flagged, entry 0 keeps the lane sane
        const u32 f = h_freq[src], c = h_cum[src];
        const u64 a1 = ((u64)D.RF * f) << b;  // max_shrunk_state + 1 (rANS.py:112)
        u32 k_lo = 0, k_hi = 0;
        while ((D.L >> (k_lo * b)) >= a1) ++k_lo;
        while ((m->H >> (k_hi * b)) >= a1) ++k_hi;
        if (k_hi > k_lo + 1 || (k_lo + 1) * b > 31) return SCL_OK;  // the first cannot happen (H < L 2^b)
        u64 thresh = a1 << (k_lo * b);
        if (thresh > 0x7FFFFFFFull) thresh = 0x7FFFFFFFull;  // never reached by x <= H < 2^31: always "below"
        const double inv_f = 1.0 / (double)f;
        union {
            double d;
            u64 u;
        } pun;
        pun.d = inv_f;
        const u64 bits64 = pun.u;
        enc[s] = make_uint4((u32)bits64, (u32)(bits64 >> 32), (u32)thresh, c);
        aux[s] = (M - f) | (((k_lo + 1) * b) << 24);
    }
    for (u32 s = 0; s < D.K; ++s)
        for (u32 j = 0; j < h_freq[s]; ++j) dec[h_cum[s] + j] = make_uint2(h_freq[s] | (s << 24), j);
    hipError_t e = hipMalloc((void **)&m->d_encb_tab, 256 * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_encb_aux, 256 * sizeof(u32));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_decb_tab, M * sizeof(uint2));
    if (e == hipSuccess) e = hipMemcpy(m->d_encb_tab, enc.data(), 256 * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_encb_aux, aux.data(), 256 * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_decb_tab, dec.data(), M * sizeof(uint2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("rans_model_create: fast-path (NUM_BITS_OUT > 1) table upload failed: %s", hipGetErrorString(e));
        return SCL_E_HIP;
    }
    RansFastBDev &F = m->fbdev;
    F.K = D.K;
    F.nsb = D.nsb;
    F.size_bits = D.size_bits;
    F.m_log2 = D.m_log2;
    F.L = (u32)D.L;
    F.M = M;
    F.b = b;
    F.cbl = 32 - scl_bit_width_u64(D.L);  // clz of any state that needs no refill
    F.d_enc = m->d_encb_tab;
    F.d_aux = m->d_encb_aux;
    F.d_dec = m->d_decb_tab;
    m->fastb = 1;
    return SCL_OK;
}

void rans_fastb_encode_launch(const scl_rans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                              u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                              u32 *d_status, hipStream_t st) {
    const u32 blocks = (u32)((n_chunks + RB_THREADS - 1) / RB_THREADS);
    if (m->fbdev.K < 256)
        hipLaunchKernelGGL((rans_encode_fastb_kernel<true>), dim3(blocks), dim3(RB_THREADS), 0, st, m->fbdev, d_sym,
                           sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_bit_off, d_nbits, d_status);
    else
        hipLaunchKernelGGL((rans_encode_fastb_kernel<false>), dim3(blocks), dim3(RB_THREADS), 0, st, m->fbdev, d_sym,
                           sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_bit_off, d_nbits, d_status);
}

void rans_fastb_decode_launch(const scl_rans_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st) {
    if (n_chunks > 2ull * 256 * RBD_THREADS_SMALL)  // more than two 256-lane workgroups per CU: the 1024-lane form
        hipLaunchKernelGGL((rans_decode_fastb_kernel<RBD_THREADS>), dim3((u32)((n_chunks + RBD_THREADS - 1) / RBD_THREADS)),
                           dim3(RBD_THREADS), 0, st, m->fbdev, d_in, in_size_bytes, d_bit_off, d_in_nbits, n_chunks,
                           d_out_sym, out_stride, out_cap, d_out_lens, d_consumed, d_status);
    else
        hipLaunchKernelGGL((rans_decode_fastb_kernel<RBD_THREADS_SMALL>),
                           dim3((u32)((n_chunks + RBD_THREADS_SMALL - 1) / RBD_THREADS_SMALL)), dim3(RBD_THREADS_SMALL), 0,
                           st, m->fbdev, d_in, in_size_bytes, d_bit_off, d_in_nbits, n_chunks, d_out_sym, out_stride,
                           out_cap, d_out_lens, d_consumed, d_status);
}
