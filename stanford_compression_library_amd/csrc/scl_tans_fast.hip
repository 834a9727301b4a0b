// scl_tans_fast.hip -- gfx950 fast path of batched tANS (cached rANS, reference scl/compressors/tANS.py).
//
// Same bit stream and the same lookup tables as scl_tans.hip (tANSEncoder.encode_symbol :126-157,
// tANSDecoder.decode_symbol :239-250), served when RANGE_FACTOR*M <= 8192 so that the tables sit in LDS next to
// the per-lane stream rings.  One workgroup of 1024 lanes per CU; stream I/O is the line-granular scheme of
// scl_ans_fast_io.h (shared with the rANS fast kernels: the stream layout is identical).
//   encode, per symbol: one 8-byte read {thresh | (nbits_base+1) << 16, row address} by symbol, then one 2-byte read
//                       of base_encode_step_table[(s, x >> nb)] by state -- no multiply, no divide.  (16-byte entries
//                       until round 3: the encoder was LDS bound, 82 % busy with random 16-byte gathers at half the
//                       LDS rate, profiles/r03_instruction_mix.txt; thresh <= 2L <= 2^14 and nbits <= 14 share a word
//                       at no cost, the two subtractions that use them select their half by SDWA);
//   decode, per symbol: one 4-byte read base_decode_step_table[x] = (x_shrunk << 8 | s) by state;
//                       expand_state_num_bits_table is clz.
#include <vector>

#include "scl_ans_fast_io.h"
#include "scl_tans_internal.h"

// Workgroup size: 1024 lanes (one workgroup per CU, tables staged once) for batches that fill the chip that way;
// batches of up to 131 072 chunks take 256-lane workgroups so that they spread over all CUs (see scl_rans_fast.hip).
#define TF_THREADS 1024
#define TF_THREADS_SMALL 256

struct TfSym {
    u32 bits, k;
};

__device__ __forceinline__ TfSym tf_encode_symbol(u32 &x, u32 addr, const char *sym_tab, const char *lds) {
    const uint2 e = *reinterpret_cast<const uint2 *>(sym_tab + addr);
    const u32 neg = (x - (e.x & 0xFFFFu)) >> 31;  // 1 iff x < shrink_state_thresh_table[s]
    const u32 nb = (e.x >> 16) - neg;             // shrink_state_num_out_bits_base_table[s] (+1 above the threshold)
    TfSym r;
    r.bits = __builtin_amdgcn_ubfe(x, 0, nb);
    r.k = nb;
    const u32 xs = x >> nb;
    x = *reinterpret_cast<const u16 *>(lds + e.y + xs + xs);  // base_encode_step_table[(s, xs)]
    return r;
}

template <typename TfOut>
__device__ __forceinline__ void tf_encode16(const uint4 v, u32 &x, TfOut &o, u32 &bad, char *lds, const char *sym_tab) {
    const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const u32 w = wv[d];
        const u32 a0 = (w << 3) & 0x7F8u, a1 = (w >> 5) & 0x7F8u, a2 = (w >> 13) & 0x7F8u, a3 = (w >> 21) & 0x7F8u;
        bad = max(max(bad, max(a0, a1)), max(a2, a3));
        const TfSym s0 = tf_encode_symbol(x, a0, sym_tab, lds);
        const TfSym s1 = tf_encode_symbol(x, a1, sym_tab, lds);
        o.put(lds, (s1.bits << s0.k) | s0.bits, s0.k + s1.k);  // later symbol in front (tANS.py:183 prepends)
        const TfSym s2 = tf_encode_symbol(x, a2, sym_tab, lds);
        const TfSym s3 = tf_encode_symbol(x, a3, sym_tab, lds);
        o.put(lds, (s3.bits << s2.k) | s2.bits, s2.k + s3.k);
    }
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) tans_encode_fast_kernel(TansFastDev P, const u8 *__restrict__ sym,
                                                                     u64 sym_stride, const u32 *__restrict__ lens,
                                                                     u32 chunk_len, u64 n_chunks,
                                                                     u8 *__restrict__ out, u64 out_stride,
                                                                     u64 *__restrict__ out_bit_off,
                                                                     u32 *__restrict__ out_nbits,
                                                                     u32 *__restrict__ status) {
    // [0,128K) word ring | [128K,132K) per-symbol table | [132K,148K) encode table (u16)
    // the 1024-lane form leaves 128 VGPRs per lane: no room to hold half a line there
    typedef AnsBackWriter<THREADS, (THREADS <= 256)> TfOut;
    constexpr u32 TF_RING_BYTES = TfOut::RING_BYTES;
    __shared__ __attribute__((aligned(16))) char s_lds[TF_RING_BYTES + 4096 + 8192 * 2];
    char *lds = s_lds;
    const char *sym_tab = s_lds + TF_RING_BYTES;
    if (threadIdx.x < 256) {  // row offsets are stored relative to the step table: make them LDS addresses
        const uint4 e = P.d_enc_sym[threadIdx.x];
        reinterpret_cast<uint2 *>(s_lds + TF_RING_BYTES)[threadIdx.x] =
            make_uint2(e.x | (e.z << 16), e.y + TF_RING_BYTES + 4096);
    }
    for (u32 i = threadIdx.x; i < P.L; i += THREADS)
        reinterpret_cast<u16 *>(s_lds + TF_RING_BYTES + 4096)[i] = P.d_enc_tab[i];
    __syncthreads();
    const u64 c = (u64)blockIdx.x * THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 n = lens ? lens[c] : chunk_len;
    const u8 *src = sym + c * sym_stride;
    TfOut o;
    o.init(threadIdx.x, out + (c + 1) * out_stride);
    u32 x = P.L, bad = 0;

    // 1024 lanes per workgroup leave 128 VGPRs per lane, so the input is staged as 64-byte half lines (two
    // 16-register buffers) instead of the whole lines of the rANS encoder
    const u32 n_lines = n >> 6;  // 64-byte units
    const uint4 *src16 = reinterpret_cast<const uint4 *>(src);
    uint4 cur[4], nxt[4];
    if (n_lines) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = src16[i];
    }
#pragma nounroll
    for (u32 t = 0; t < n_lines; ++t) {
        if (t + 1 < n_lines) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nxt[i] = src16[4 * (t + 1) + i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tf_encode16(cur[i], x, o, bad, lds, sym_tab);
            if (i & 1) o.maybe_flush(lds);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
    u32 i = n_lines << 6;
    for (; i + 16 <= n; i += 16) {
        tf_encode16(*reinterpret_cast<const uint4 *>(src + i), x, o, bad, lds, sym_tab);
        o.maybe_flush(lds);
    }
    for (; i < n; ++i) {
        const u32 a = (u32)src[i] << 3;
        bad = max(bad, a);
        const TfSym s = tf_encode_symbol(x, a, sym_tab, lds);
        o.put(lds, s.bits, s.k);
        if ((i & 15u) == 15u) o.maybe_flush(lds);
    }
    o.maybe_flush(lds);
    o.put32(lds, x, P.nsb);
    u32 st = (bad >= (P.K << 3)) ? SCL_ST_SYMBOL : 0u;
    if (P.size_bits < 32 && (n >> P.size_bits)) st |= SCL_ST_SIZE;
    o.put32(lds, n, P.size_bits);
    const u64 total = o.finish(lds);
    out_bit_off[c] = (c + 1) * out_stride * 8 - total;
    out_nbits[c] = (u32)total;
    if (status) status[c] = st;
}

// decode one symbol from the 32-bit lookahead (bits consumed from its top); returns the table word
__device__ __forceinline__ u32 tf_decode_symbol(u32 &x, u32 lk, u32 &used, const char *tab, u32 idx_mask, u32 cb) {
    const u32 e = *reinterpret_cast<const u32 *>(tab + ((x << 2) & idx_mask));  // base_decode_step_table[x]
    const u32 xs = e >> 8;
    const u32 cl = (u32)__builtin_clz(xs);                     // xs >= RANGE_FACTOR > 0
    const u32 y = __builtin_amdgcn_alignbit(xs, lk, 32 - cl);  // (xs << cl) | (lk >> (32 - cl))
    x = y >> cb;                                               // nb = NUM_STATE_BITS - bit_width(xs) new bits
    used = cl - cb;
    return e;
}

template <typename TfIn>
__device__ __forceinline__ uint4 tf_decode16(u32 &x, TfIn &r, char *lds, const char *tab, u32 idx_mask, u32 cb) {
    u32 ow[4];
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        u32 o = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 lk = r.look();
            u32 ua, ub;
            const u32 ea = tf_decode_symbol(x, lk, ua, tab, idx_mask, cb);
            const u32 eb = tf_decode_symbol(x, lk << ua, ub, tab, idx_mask, cb);
            r.advance(lds, ua + ub);  // <= 2*13 bits of the 32-bit lookahead
            o = __builtin_amdgcn_perm(o, ea, 0x06050400u);  // o = (o << 8) | (ea & 0xFF)
            o = __builtin_amdgcn_perm(o, eb, 0x06050400u);
            asm volatile("" : "+v"(o) : : "memory");
        }
        ow[d] = o;
    }
    r.maybe_refill(lds);
    return make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) tans_decode_fast_kernel(TansFastDev P, const u8 *__restrict__ in,
                                                                     u64 in_size_bytes,
                                                                     const u64 *__restrict__ bit_off,
                                                                     const u32 *__restrict__ in_nbits, u64 n_chunks,
                                                                     u8 *__restrict__ out_sym, u64 out_stride,
                                                                     u32 out_cap, u32 *__restrict__ out_lens,
                                                                     u32 *__restrict__ consumed,
                                                                     u32 *__restrict__ status) {
    typedef AnsBitReader<THREADS> TfIn;
    constexpr u32 TF_RING_BYTES = TfIn::RING_BYTES;
    __shared__ __attribute__((aligned(16))) char s_lds[TF_RING_BYTES + 8192 * 4];
    char *lds = s_lds;
    const char *tab = s_lds + TF_RING_BYTES;
    for (u32 i = threadIdx.x; i < P.L; i += THREADS)
        reinterpret_cast<u32 *>(s_lds + TF_RING_BYTES)[i] = P.d_dec_tab[i];
    __syncthreads();
    const u64 c = (u64)blockIdx.x * THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 avail = in_nbits[c];
    u32 st = 0;
    if (avail < P.size_bits + P.nsb) {
        out_lens[c] = 0;
        consumed[c] = P.size_bits + P.nsb;
        if (status) status[c] = SCL_ST_TRUNCATED;
        return;
    }
    TfIn r;
    r.init(in, in_size_bytes, bit_off[c], lds, threadIdx.x);
    u32 n = r.get(lds, P.size_bits);
    u32 x = r.get(lds, P.nsb);
    out_lens[c] = n;
    if (x < P.L || x >= 2 * P.L) {  // KeyError on base_decode_step_table in the reference
        st |= SCL_ST_STATE;
        n = 0;
    }
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    const u32 st_header = st;
    const u32 idx_mask = (P.L - 1) << 2, cb = 32 - P.nsb;
    u8 *dst = out_sym + c * out_stride;
    u32 i = n;
    while (i & 15u) {
        u32 used;
        const u32 e = tf_decode_symbol(x, r.look(), used, tab, idx_mask, cb);
        r.advance(lds, used);
        dst[--i] = (u8)e;
        if ((i & 3u) == 0) r.maybe_refill(lds);
    }
    while (i & 127u) {
        const uint4 v = tf_decode16(x, r, lds, tab, idx_mask, cb);
        i -= 16;
        *reinterpret_cast<uint4 *>(dst + i) = v;
    }
    CoopLineStore cs;  // whole waves of equally long chunks store cooperatively (scl_ans_fast_io.h)
    cs.init(out_sym, c, out_stride, i);
#pragma nounroll
    while (i) {  // one full 128-byte line per iteration
        uint4 a[8];
#pragma unroll
        for (int b = 7; b >= 0; --b) a[b] = tf_decode16(x, r, lds, tab, idx_mask, cb);
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
    else if (st_header == 0 && x != P.L) st |= SCL_ST_STATE;  // assert state == INITIAL_STATE (tANS.py:277)
    consumed[c] = used_bits;
    if (status) status[c] = st;
}

// ---------------------------------------------------------------------------------------------------
// host side: repack the device-built reference tables (scl_tans.hip: tans_build_tables) for the fast kernels
// ---------------------------------------------------------------------------------------------------
int tans_fast_build_tables(scl_tans_model *m, const u32 *h_freq, const u32 *h_cum) {
    const TansDev &D = m->dev;
    m->fast = 0;
    if (D.L > 8192 || D.K < 1 || D.nsb > 14) return SCL_OK;
    const u32 L = D.L;
    std::vector<u32> enc(L), nbits(D.K), thresh(D.K), dsym(L), dxs(L);
    hipError_t e = hipMemcpy(enc.data(), m->d_enc, L * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(nbits.data(), m->d_nbits, D.K * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(thresh.data(), m->d_thresh, D.K * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(dsym.data(), m->d_dec_sym, L * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(dxs.data(), m->d_dec_xs, L * 4, hipMemcpyDeviceToHost);
    std::vector<uint4> fsym(256);
    std::vector<u16> fenc(L);
    std::vector<u32> fdec(L);
    for (u32 s = 0; s < 256; ++s) {
        const u32 src = s < D.K ? s : 0;
        // row of symbol s starts at entry RF*c[s] and is indexed by x_shrunk - RF*f[s]; as a byte offset from the
        // start of the step table, 2*(RF*c - RF*f) (may be negative: u32 wrap-around); the kernel adds the table's
        // LDS address (ring size + per-symbol table) when it stages the entries
        const i64 row = 2 * ((i64)D.RF * h_cum[src] - (i64)D.RF * h_freq[src]);
        fsym[s] = make_uint4(thresh[src], (u32)row, nbits[src] + 1, 0);
    }
    for (u32 i = 0; i < L; ++i) {
        fenc[i] = (u16)enc[i];
        fdec[i] = (dxs[i] << 8) | dsym[i];
    }
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_fenc_sym, 256 * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_fenc_tab, L * sizeof(u16));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_fdec_tab, L * sizeof(u32));
    if (e == hipSuccess) e = hipMemcpy(m->d_fenc_sym, fsym.data(), 256 * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_fenc_tab, fenc.data(), L * sizeof(u16), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_fdec_tab, fdec.data(), L * sizeof(u32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("tans_model_create: fast-path table upload failed: %s", hipGetErrorString(e));
        return SCL_E_HIP;
    }
    m->fdev.K = D.K;
    m->fdev.L = L;
    m->fdev.nsb = D.nsb;
    m->fdev.size_bits = D.size_bits;
    m->fdev.d_enc_sym = m->d_fenc_sym;
    m->fdev.d_enc_tab = m->d_fenc_tab;
    m->fdev.d_dec_tab = m->d_fdec_tab;
    m->fast = 1;
    return SCL_OK;
}

void tans_fast_encode_launch(const scl_tans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                             u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                             u32 *d_status, hipStream_t st) {
    if (n_chunks > 2ull * 256 * TF_THREADS_SMALL)
        hipLaunchKernelGGL((tans_encode_fast_kernel<TF_THREADS>), dim3((u32)((n_chunks + TF_THREADS - 1) / TF_THREADS)),
                           dim3(TF_THREADS), 0, st, m->fdev, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out,
                           out_stride, d_bit_off, d_nbits, d_status);
    else
        hipLaunchKernelGGL((tans_encode_fast_kernel<TF_THREADS_SMALL>),
                           dim3((u32)((n_chunks + TF_THREADS_SMALL - 1) / TF_THREADS_SMALL)), dim3(TF_THREADS_SMALL), 0, st,
                           m->fdev, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_bit_off,
                           d_nbits, d_status);
}

void tans_fast_decode_launch(const scl_tans_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                             const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                             u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st) {
    if (n_chunks > 2ull * 256 * TF_THREADS_SMALL)
        hipLaunchKernelGGL((tans_decode_fast_kernel<TF_THREADS>), dim3((u32)((n_chunks + TF_THREADS - 1) / TF_THREADS)),
                           dim3(TF_THREADS), 0, st, m->fdev, d_in, in_size_bytes, d_bit_off, d_in_nbits, n_chunks,
                           d_out_sym, out_stride, out_cap, d_out_lens, d_consumed, d_status);
    else
        hipLaunchKernelGGL((tans_decode_fast_kernel<TF_THREADS_SMALL>),
                           dim3((u32)((n_chunks + TF_THREADS_SMALL - 1) / TF_THREADS_SMALL)), dim3(TF_THREADS_SMALL), 0, st,
                           m->fdev, d_in, in_size_bytes, d_bit_off, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
}
