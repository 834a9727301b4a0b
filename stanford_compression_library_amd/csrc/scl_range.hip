// scl_range.hip -- batched carry-less byte-wise ("Russian") range coder for gfx950,
// one wavefront lane per chunk.
//
// Replaces reference scl/compressors/range_coder.py:
//   RangeCoderParams :55-76 (TOP = 2^(P-8), BOTTOM = 2^(P-16), MASK = 2^P - 1)
//   RangeEncoder: shrink_range :88-105, normalize :107-179, flush :181-186, encode_block :188-207
//   RangeDecoder: decode_symbol :225-238, normalize :240-267, decode_block :269-317
// Stream layout per chunk: [n : DATA_BLOCK_SIZE_BITS][one byte per normalisation step]...[P/8 flush bytes]
// State is u32 for P <= 32 and u64 for P in 40..64: low + range <= MASK always (no carry), c*(range//M) <= range.
// The decoder's vector search  max{s : low + c[s]*(range//M) <= state}  (:232-237) is evaluated as
// q = (state - low) // (range//M) followed by a search of q in the cumulative table (exact).
#include <string.h>

#include <vector>

#include "scl_range_internal.h"

template <typename ST>
__device__ __forceinline__ ST range_div_M(ST range, const RangeDev &P) {
    return (P.m_log2 != 0xFFFFFFFFu) ? (ST)(range >> P.m_log2) : (ST)(range / P.M);
}

// one normalisation decision: returns true if a byte must be shifted out (and fixes range first)
template <typename ST>
__device__ __forceinline__ bool range_needs_byte(ST low, ST &range, ST TOP, ST BOTTOM) {
    if ((ST)(low ^ (ST)(low + range)) < TOP) return true;  // top byte settled (:117)
    if (range < BOTTOM) {                          // underflow: clamp range to the byte boundary (:136-170)
        range = ((ST)0 - low) & (BOTTOM - 1);      // == (MASK + 1 - low) & (BOTTOM - 1) since BOTTOM | 2^P
        return true;
    }
    return false;
}

// SYM = u8: alphabets up to 256, tables staged in LDS.  SYM = u16 (the *_u16 entry points): alphabets up to 65536,
// tables read where they are in device memory; strides count SYMBOLS in both.
template <typename ST, typename SYM = u8>
__global__ void __launch_bounds__(256) range_encode_kernel(RangeDev P, const SYM *__restrict__ sym, u64 sym_stride,
                                                          const u32 *__restrict__ lens, u32 chunk_len, u64 n_chunks,
                                                          u8 *__restrict__ out, u64 out_stride,
                                                          u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits,
                                                          u32 *__restrict__ status) {
    __shared__ u32 s_f_lds[sizeof(SYM) == 1 ? 256 : 1];
    __shared__ u32 s_c_lds[sizeof(SYM) == 1 ? 256 : 1];
    const u32 *s_f = P.d_freq, *s_c = P.d_cum;
    if constexpr (sizeof(SYM) == 1) {
        scl_load_table(s_f_lds, P.d_freq, P.K);
        scl_load_table(s_c_lds, P.d_cum, P.K);
        __syncthreads();
        s_f = s_f_lds;
        s_c = s_c_lds;
    }
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 n = lens ? lens[c] : chunk_len;
    const SYM *src = sym + c * sym_stride;
    const ST MASK = (P.P == 8 * sizeof(ST)) ? (ST)~(ST)0 : (ST)(((ST)1 << P.P) - 1);
    const ST TOP = (ST)1 << (P.P - 8), BOTTOM = (ST)1 << (P.P - 16);
    const u32 SH = P.P - 8;
    FwdBitWriter w;
    w.init(out + c * out_stride, out_stride);
    u32 st = 0;
    if (P.size_bits < 32 && (n >> P.size_bits)) st |= SCL_ST_SIZE;
    w.put(n, P.size_bits);
    ST low = 0, range = MASK;
    for (u32 i = 0; i < n; ++i) {
        u32 s = src[i];
        if (s >= P.K) {
            st |= SCL_ST_SYMBOL;
            s = 0;
        }
        range = range_div_M<ST>(range, P);  // shrink_range :101-103
        low += (ST)s_c[s] * range;
        range *= (ST)s_f[s];
        while (range_needs_byte<ST>(low, range, TOP, BOTTOM)) {
            w.put((u32)(low >> SH), 8);
            low = (ST)(low << 8) & MASK;
            range <<= 8;
        }
    }
    for (u32 i = 0; i < P.P / 8; ++i) {  // flush :181-186
        w.put((u32)(low >> SH), 8);
        low = (ST)(low << 8) & MASK;
    }
    const u64 total = w.finish();
    if (w.overflow) st |= SCL_ST_CAPACITY;
    out_bit_off[c] = c * out_stride * 8;
    out_nbits[c] = (u32)total;
    if (status) status[c] = st;
}

template <typename ST, typename SYM = u8>
__global__ void __launch_bounds__(256) range_decode_kernel(RangeDev P, const u8 *__restrict__ in, u64 in_size_bytes,
                                                          const u64 *__restrict__ bit_off,
                                                          const u32 *__restrict__ in_nbits, u64 n_chunks,
                                                          SYM *__restrict__ out_sym, u64 out_stride, u32 out_cap,
                                                          u32 *__restrict__ out_lens, u32 *__restrict__ consumed,
                                                          u32 *__restrict__ status) {
    __shared__ u32 s_f_lds[sizeof(SYM) == 1 ? 256 : 1];
    __shared__ u32 s_c_lds[sizeof(SYM) == 1 ? 256 : 1];
    const u32 *s_f = P.d_freq, *s_c = P.d_cum;
    if constexpr (sizeof(SYM) == 1) {
        scl_load_table(s_f_lds, P.d_freq, P.K);
        scl_load_table(s_c_lds, P.d_cum, P.K);
        __syncthreads();
        s_f = s_f_lds;
        s_c = s_c_lds;
    }
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const ST MASK = (P.P == 8 * sizeof(ST)) ? (ST)~(ST)0 : (ST)(((ST)1 << P.P) - 1);
    const ST TOP = (ST)1 << (P.P - 8), BOTTOM = (ST)1 << (P.P - 16);
    BitReader r;
    r.init(in, in_size_bytes, bit_off[c], in_nbits[c]);
    const u64 start = r.pos;
    u32 st = 0;
    u32 n = r.get(P.size_bits);
    ST state = 0;
    for (u32 i = 0; i < P.P / 8; ++i) state = (ST)(state << 8) | (ST)r.get(8);  // :289-291
    if (r.truncated) {
        st |= SCL_ST_TRUNCATED;
        n = 0;
    }
    out_lens[c] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    SYM *dst = out_sym + c * out_stride;
    ST low = 0, range = MASK;
    for (u32 i = 0; i < n; ++i) {
        const ST rr = range_div_M<ST>(range, P);
        const ST q = (ST)(state - low) / rr;  // rr >= 1 because range >= BOTTOM >= M after normalize
        u32 s;
        if (P.d_slot2sym && q < P.M) {
            s = P.d_slot2sym[q];
        } else {
            u32 lo = 0, hi = P.K;
            while (hi - lo > 1) {
                const u32 mid = (lo + hi) >> 1;
                if ((ST)s_c[mid] <= q)
                    lo = mid;
                else
                    hi = mid;
            }
            s = lo;
        }
        dst[i] = (SYM)s;
        range = rr;
        low += (ST)s_c[s] * range;
        range *= (ST)s_f[s];
        while (range_needs_byte<ST>(low, range, TOP, BOTTOM)) {
            state = ((ST)(state << 8) | (ST)r.get(8)) & MASK;
            low = (ST)(low << 8) & MASK;
            range <<= 8;
        }
        if (r.truncated) break;
    }
    if (r.truncated) st |= SCL_ST_TRUNCATED;
    consumed[c] = (u32)(r.pos - start);
    if (status) status[c] = st;
}

// ---- host API -------------------------------------------------------------------------------------------
extern "C" int scl_range_model_create(const uint32_t *h_freq, uint32_t K, uint32_t precision, uint32_t size_bits,
                                      scl_range_model **out) {
    SCL_REQUIRE(out, "range_model_create: null output");
    *out = nullptr;
    SCL_REQUIRE(h_freq && K >= 1 && K <= SCL_MAX_ALPHABET, "range_model_create: alphabet size %u outside 1..65536", K);
    SCL_REQUIRE(precision % 8 == 0 && precision >= 16 && precision <= 64,
                "range_model_create: PRECISION %u is not a multiple of 8 in 16..64 (assert PRECISION %% 8 == 0, "
                "range_coder.py:64)",
                precision);
    SCL_REQUIRE(size_bits >= 1 && size_bits <= 32, "range_model_create: DATA_BLOCK_SIZE_BITS %u outside 1..32",
                size_bits);
    u64 M = 0;
    std::vector<u32> cum_v(K);
    u32 *cum = cum_v.data();
    for (u32 i = 0; i < K; ++i) {
        SCL_REQUIRE(h_freq[i] > 0, "range_model_create: zero frequency (assert min(freq) > 0, range_coder.py:84)");
        cum[i] = (u32)M;
        M += h_freq[i];
        SCL_REQUIRE(M <= (1ull << 31), "range_model_create: total_freq too large");
    }
    SCL_REQUIRE(M <= (1ull << (precision - 16)),
                "range_model_create: total_freq %llu > BOTTOM = 2^%u (assert at range_coder.py:85)",
                (unsigned long long)M, precision - 16);
    scl_range_model *m = new scl_range_model();
    m->device = scl_current_device();
    m->dev.K = K;
    m->dev.P = precision;
    m->dev.size_bits = size_bits;
    m->dev.M = (u32)M;
    m->dev.m_log2 = (M & (M - 1)) == 0 ? scl_bit_width_u64(M) - 1 : 0xFFFFFFFFu;
    const bool lut = M <= 65536 && K <= 256;  // byte-valued slot -> symbol table (the u16 kernels search d_cum)
    static thread_local u8 slot2sym[65536];
    if (lut)
        for (u32 s = 0; s < K; ++s)
            for (u32 j = 0; j < h_freq[s]; ++j) slot2sym[cum[s] + j] = (u8)s;
    const u64 tab_entries = K > 256 ? K : 256;
    hipError_t e = hipMalloc((void **)&m->d_freq, tab_entries * sizeof(u32));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_cum, tab_entries * sizeof(u32));
    if (e == hipSuccess && lut) e = hipMalloc((void **)&m->d_slot2sym, M);
    if (e == hipSuccess) e = hipMemcpy(m->d_freq, h_freq, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_cum, cum, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess && lut) e = hipMemcpy(m->d_slot2sym, slot2sym, M, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("range_model_create: device table upload failed: %s", hipGetErrorString(e));
        scl_range_model_destroy(m);
        return SCL_E_HIP;
    }
    m->dev.d_freq = m->d_freq;
    m->dev.d_cum = m->d_cum;
    m->dev.d_slot2sym = m->d_slot2sym;
    const int rc = K <= 256 ? range_fast_build_tables(m, h_freq, cum) : SCL_OK;
    if (rc != SCL_OK) {
        scl_range_model_destroy(m);
        return rc;
    }
    *out = m;
    return SCL_OK;
}

extern "C" void scl_range_model_destroy(scl_range_model *m) {
    if (!m) return;
    if (m->d_freq) (void)hipFree(m->d_freq);
    if (m->d_cum) (void)hipFree(m->d_cum);
    if (m->d_slot2sym) (void)hipFree(m->d_slot2sym);
    if (m->d_enc_tab) (void)hipFree(m->d_enc_tab);
    delete m;
}

extern "C" int scl_range_fast_path(const scl_range_model *m) { return (m && m->fast) ? 1 : 0; }

extern "C" uint64_t scl_range_slot_bytes(const scl_range_model *m, uint64_t n_symbols) {
    if (!m) return 0;
    // a symbol can shift out at most P/8 bytes (range drops from 2^P to >= 1), typically log2(M/f)/8
    const u64 bytes = (m->dev.size_bits + 7) / 8 + n_symbols * (m->dev.P / 8) + m->dev.P / 8;
    return scl_round_up(bytes + 4, 128);
}

extern "C" int scl_range_encode_batch(const scl_range_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                      const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks, uint8_t *d_out,
                                      uint64_t out_stride, uint64_t *d_out_bit_offset, uint32_t *d_out_nbits,
                                      uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits, "range_encode_batch: null pointer argument");
    SCL_REQUIRE(m->dev.K <= 256, "range_encode_batch: alphabet of %u symbols: use scl_range_encode_batch_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "range_encode_batch")) return rc_dev;
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride * 8 < (1ull << 32),
                "range_encode_batch: bad out_stride %llu", (unsigned long long)out_stride);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0, "range_encode_batch: d_out must be 16-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    const bool tuned = !scl_force_generic();
    RowRelay relay;  // rows that do not start on 16-byte boundaries are re-laid for the tuned kernels
    if (tuned && m->fast)
        if (int rc_r = relay.in(d_sym, sym_stride, chunk_len, n_chunks, (hipStream_t)stream)) return rc_r;
    if (tuned && m->fast && ((uintptr_t)d_sym & 15) == 0 && (sym_stride & 15) == 0)
        range_fast_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                                 d_out_nbits, d_status, (hipStream_t)stream);
    else if (m->dev.P <= 32)
        hipLaunchKernelGGL(range_encode_kernel<u32>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, m->dev,
                           d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                           d_out_nbits, d_status);
    else
        hipLaunchKernelGGL(range_encode_kernel<u64>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, m->dev,
                           d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                           d_out_nbits, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

// ---- wave-striped slots (ABI version 8; scl_range_fast.hip: RgOutT, scl_ans_fast_io.h: AnsBitReaderT) ------------------------
extern "C" int scl_range_striped_ok(const scl_range_model *m) {
    return (m && m->dev.K <= 256 && range_fast_striped_ok(m)) ? 1 : 0;
}

extern "C" int scl_range_encode_batch_striped(const scl_range_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                              const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                              uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                              uint32_t *d_out_nbits, uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits, "range_encode_batch_striped: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "range_encode_batch_striped")) return rc_dev;
    SCL_REQUIRE(scl_range_striped_ok(m), "range_encode_batch_striped: this model is not served by the striped kernels");
    SCL_REQUIRE(!scl_force_generic(), "range_encode_batch_striped: the calling thread keeps the tuned kernels out");
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride < (1ull << 24) && ((uintptr_t)d_out & 15) == 0,
                "range_encode_batch_striped: d_out must be 16-byte aligned and out_stride a multiple of 16 below 2^24");
    if (n_chunks == 0) return SCL_OK;
    RowRelay relay;  // rows that do not start on 16-byte boundaries are re-laid
    if (int rc_r = relay.in(d_sym, sym_stride, chunk_len, n_chunks, (hipStream_t)stream)) return rc_r;
    if (!scl_rows_aligned(d_sym, sym_stride)) {
        scl_set_error("range_encode_batch_striped: out of device memory re-laying unaligned symbol rows");
        return SCL_E_ALLOC;
    }
    range_fast_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                             d_out_nbits, d_status, (hipStream_t)stream, true);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_range_decode_batch_striped(const scl_range_model *m, const uint8_t *d_in, uint64_t in_stride,
                                              const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                              uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                                              uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                              uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "range_decode_batch_striped: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "range_decode_batch_striped")) return rc_dev;
    SCL_REQUIRE(scl_range_striped_ok(m), "range_decode_batch_striped: this model is not served by the striped kernels");
    SCL_REQUIRE(!scl_force_generic(), "range_decode_batch_striped: the calling thread keeps the tuned kernels out");
    SCL_REQUIRE(((uintptr_t)d_in & 15) == 0 && in_stride % 16 == 0 && in_stride > 0 && in_stride < (1ull << 24),
                "range_decode_batch_striped: d_in must be 16-byte aligned and in_stride a multiple of 16 below 2^24");
    if (n_chunks == 0) return SCL_OK;
    RowRelay relay;  // output rows the kernels cannot store to go through aligned scratch and are copied back
    if (int rc_r = relay.out_begin(d_out_sym, out_stride, out_cap, n_chunks, (hipStream_t)stream)) return rc_r;
    if (!scl_rows_aligned(d_out_sym, out_stride)) {
        scl_set_error("range_decode_batch_striped: out of device memory re-laying unaligned output rows");
        return SCL_E_ALLOC;
    }
    range_fast_decode_launch(m, d_in, in_stride, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                             d_out_lens, d_consumed, d_status, (hipStream_t)stream, true);
    SCL_HIP_TRY(hipGetLastError());
    return relay.out_end(d_out_lens);
}

extern "C" int scl_range_decode_batch(const scl_range_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                      const uint64_t *d_bit_offset, const uint32_t *d_in_nbits, uint64_t n_chunks,
                                      uint8_t *d_out_sym, uint64_t out_stride, uint32_t out_cap, uint32_t *d_out_lens,
                                      uint32_t *d_consumed, uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "range_decode_batch: null pointer argument");
    SCL_REQUIRE(m->dev.K <= 256, "range_decode_batch: alphabet of %u symbols: use scl_range_decode_batch_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "range_decode_batch")) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0, "range_decode_batch: d_in must be 4-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    const bool tuned = !scl_force_generic();
    RowRelay relay;  // output rows the tuned kernels cannot store to go through aligned scratch and are copied back
    if (tuned && m->fast && ((uintptr_t)d_in & 15) == 0)
        if (int rc_r = relay.out_begin(d_out_sym, out_stride, out_cap, n_chunks, (hipStream_t)stream)) return rc_r;
    if (tuned && m->fast && ((uintptr_t)d_in & 15) == 0 && ((uintptr_t)d_out_sym & 15) == 0 && (out_stride & 15) == 0)
        range_fast_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                                 out_cap, d_out_lens, d_consumed, d_status, (hipStream_t)stream);
    else if (m->dev.P <= 32)
        hipLaunchKernelGGL(range_decode_kernel<u32>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, m->dev,
                           d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
    else
        hipLaunchKernelGGL(range_decode_kernel<u64>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, m->dev,
                           d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return relay.out_end(d_out_lens);
}

// ---- uint16 symbol indices: alphabets up to 65536 (any model; the any-parameter kernels) ---------------------
extern "C" int scl_range_encode_batch_u16(const scl_range_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                                          const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                          uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                          uint32_t *d_out_nbits, uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits,
                "range_encode_batch_u16: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "range_encode_batch_u16")) return rc_dev;
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride * 8 < (1ull << 32),
                "range_encode_batch_u16: bad out_stride %llu", (unsigned long long)out_stride);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_sym & 1) == 0,
                "range_encode_batch_u16: d_out must be 16-byte aligned, d_sym 2-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    if (m->dev.P <= 32)
        hipLaunchKernelGGL((range_encode_kernel<u32, u16>), dim3(blocks), dim3(threads), 0, (hipStream_t)stream, m->dev,
                           d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                           d_out_nbits, d_status);
    else
        hipLaunchKernelGGL((range_encode_kernel<u64, u16>), dim3(blocks), dim3(threads), 0, (hipStream_t)stream, m->dev,
                           d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                           d_out_nbits, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_range_decode_batch_u16(const scl_range_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                          const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                          uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                                          uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                          uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "range_decode_batch_u16: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "range_decode_batch_u16")) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0 && ((uintptr_t)d_out_sym & 1) == 0,
                "range_decode_batch_u16: d_in must be 4-byte aligned, d_out_sym 2-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    RangeDev dev = m->dev;
    dev.d_slot2sym = nullptr;  // a table of BYTES: the u16 kernels search the cumulative counts
    if (m->dev.P <= 32)
        hipLaunchKernelGGL((range_decode_kernel<u32, u16>), dim3(blocks), dim3(threads), 0, (hipStream_t)stream, dev,
                           d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
    else
        hipLaunchKernelGGL((range_decode_kernel<u64, u16>), dim3(blocks), dim3(threads), 0, (hipStream_t)stream, dev,
                           d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

// ---- single-chunk host drivers --------------------------------------------------------------------------
static int range_run_enc(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                         u32 *d_nbits, u32 *d_status, void *, u64) {
    return scl_range_encode_batch((const scl_range_model *)model, d_sym, n, nullptr, n, 1, d_out, out_stride,
                                  d_bit_off, d_nbits, d_status, nullptr);
}
static u64 range_slot(const void *model, u64 n) { return scl_range_slot_bytes((const scl_range_model *)model, n); }
static int range_run_dec(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
                         u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *, u64) {
    return scl_range_decode_batch((const scl_range_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1, d_out_sym,
                                  scl_round_up((u64)out_cap + 1, 16), out_cap, d_out_len, d_consumed, d_status,
                                  nullptr);
}

extern "C" int scl_range_encode_host(const scl_range_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                                     uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {range_run_enc, range_slot, nullptr};
    return scl_host_encode_one(call, m, h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_range_decode_host(const scl_range_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                     uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {range_run_dec, nullptr};
    return scl_host_decode_one(call, m, h_in, in_nbits, h_out_sym, out_cap, n_out, consumed);
}

static int range_run_enc16(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                           u32 *d_nbits, u32 *d_status, void *, u64) {
    return scl_range_encode_batch_u16((const scl_range_model *)model, (const u16 *)d_sym, n, nullptr, n, 1, d_out,
                                      out_stride, d_bit_off, d_nbits, d_status, nullptr);
}
static int range_run_dec16(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off,
                           const u32 *d_in_nbits, u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed,
                           u32 *d_status, void *, u64) {
    return scl_range_decode_batch_u16((const scl_range_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1,
                                      (u16 *)d_out_sym, (u64)out_cap + 1, out_cap, d_out_len, d_consumed, d_status,
                                      nullptr);
}

extern "C" int scl_range_encode_host_u16(const scl_range_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                                         uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {range_run_enc16, range_slot, nullptr};
    call.sym_bytes = 2;
    return scl_host_encode_one(call, m, (const u8 *)h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_range_decode_host_u16(const scl_range_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                         uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {range_run_dec16, nullptr};
    call.sym_bytes = 2;
    return scl_host_decode_one(call, m, h_in, in_nbits, (u8 *)h_out_sym, out_cap, n_out, consumed);
}
