// scl_rans.hip -- batched rANS encode / decode for gfx950, one wavefront lane per chunk.
//
// Replaces the per-symbol loops of reference scl/compressors/rANS.py:
//   rANSEncoder.encode_block :186-210  (shrink_state :149-161, rans_base_encode_step :138-147)
//   rANSDecoder.decode_block :270-297  (rans_base_decode_step :234-249, expand_state :251-260)
// Stream layout per chunk (bit-exact with the reference):
//   [n : DATA_BLOCK_SIZE_BITS][x_final : NUM_STATE_BITS][field(s_{n-1})] ... [field(s_0)]
// where field(s_i) = the low k*NUM_BITS_OUT bits of the pre-shrink state, MSB first.
//
// Two kernel families:
//   * generic  : any M, NUM_BITS_OUT, RANGE_FACTOR with H < 2^63 (u32 or u64 state, real division,
//                the reference's while-loops kept as loops).  Used for parameter sets outside the
//                fast path; correctness first.
//   * fast     : H < 2^31, any total 2 <= M <= 4096, NUM_BITS_OUT = 1, RANGE_FACTOR = 2^r (the reference
//                defaults, BASELINE.json configs[1]; scl_rans_fast.hip): closed-form shift
//                count, exact reciprocal division, LDS-resident tables, slot -> symbol LUT decode.
//   * fast, b>1: NUM_BITS_OUT in {2, 4, 8, 16}, M = 2^m <= 2^12, RANGE_FACTOR = 2^r, H < 2^31
//                (scl_rans_fast_b.hip): same I/O, division in binary64.
#include <string.h>

#include <vector>

#include "scl_common.h"

#include "scl_rans_internal.h"

// =====================================================================================================
// generic kernels
// =====================================================================================================
// SYM = u8: alphabets up to 256, tables staged in LDS.  SYM = u16 (the *_u16 entry points): alphabets up to 65536,
// tables read where they are (d_freq / d_cum in device memory, L2-resident); strides count SYMBOLS in both.
template <typename ST, typename SYM = u8>
__global__ void __launch_bounds__(256) rans_encode_generic(RansDev P, const SYM *__restrict__ sym, u64 sym_stride,
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
    BackBitWriter w;
    w.init(out + c * out_stride, out_stride);
    u32 st = 0;
    ST x = (ST)P.L;  // INITIAL_STATE, rANS.py:116
    const u32 b = P.b;
    for (u32 i = 0; i < n; ++i) {
        u32 s = src[i];
        if (s >= P.K) {
            st |= SCL_ST_SYMBOL;
            s = 0;
        }
        const ST f = (ST)s_f[s];
        // max_shrunk_state = RF*f*2^b - 1 <= H (rANS.py:112); shrink_state :149-161 emits the low
        // b bits per step, each group in front of the previous one -> low k*b bits, MSB first
        const ST max_shrunk = (ST)(((ST)P.RF * f) << b) - 1;
        u32 kb = 0;
        ST xs = x;
        while (xs > max_shrunk) {
            xs >>= b;
            kb += b;
        }
        if (kb) {
            const u64 field = (kb >= 64) ? (u64)x : ((u64)x & ((1ull << kb) - 1));
            w.put64(field, kb);
        }
        // rans_base_encode_step :138-147
        x = (xs / f) * (ST)P.M + (ST)s_c[s] + (xs % f);
    }
    w.put64((u64)x, P.nsb);
    if (P.size_bits < 32 && (n >> P.size_bits)) st |= SCL_ST_SIZE;
    w.put(n, P.size_bits);
    const u64 total = w.finish();
    if (w.overflow) st |= SCL_ST_CAPACITY;
    out_bit_off[c] = (c + 1) * out_stride * 8 - total;
    out_nbits[c] = (u32)total;
    if (status) status[c] = st;
}

template <typename ST, typename SYM = u8>
__global__ void __launch_bounds__(256) rans_decode_generic(RansDev P, const u8 *__restrict__ in, u64 in_size_bytes,
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
    BitReader r;
    r.init(in, in_size_bytes, bit_off[c], in_nbits[c]);
    const u64 start = r.pos;
    u32 st = 0;
    u32 n = r.get(P.size_bits);
    ST x = (ST)r.get64(P.nsb);
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
    const u32 b = P.b;
    const ST M = (ST)P.M, L = (ST)P.L;
    const u32 st_header = st;
    for (u32 i = n; i-- > 0;) {
        // rans_base_decode_step :234-249
        ST block_id, slot;
        if (P.m_log2 != 0xFFFFFFFFu) {
            block_id = x >> P.m_log2;
            slot = x & (M - 1);
        } else {
            block_id = x / M;
            slot = x - block_id * M;
        }
        // largest s with cum[s] <= slot (np.searchsorted side="right" - 1, :217-232)
        u32 lo = 0, hi = P.K;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if ((ST)s_c[mid] <= slot)
                lo = mid;
            else
                hi = mid;
        }
        x = block_id * (ST)s_f[lo] + slot - (ST)s_c[lo];
        // expand_state :251-260
        while (x < L && !r.truncated) x = (ST)(x << b) + (ST)r.get(b);
        dst[i] = (SYM)lo;  // decoded last symbol first (:291)
        if (r.truncated) break;
    }
    if (r.truncated) st |= SCL_ST_TRUNCATED;
    else if (st_header == 0 && x != L) st |= SCL_ST_STATE;  // assert state == INITIAL_STATE, :295
    consumed[c] = (u32)(r.pos - start);
    if (status) status[c] = st;
}

// =====================================================================================================
// host API
// =====================================================================================================
static int rans_model_build(const u32 *h_freq, u32 K, u64 RF, u32 b, u32 size_bits, scl_rans_model **out) {
    SCL_REQUIRE(out, "rans_model_create: null output");
    *out = nullptr;
    SCL_REQUIRE(h_freq && K >= 1 && K <= SCL_MAX_ALPHABET, "rans_model_create: alphabet size %u outside 1..65536", K);
    SCL_REQUIRE(b >= 1 && b <= 32, "rans_model_create: NUM_BITS_OUT %u outside 1..32", b);
    SCL_REQUIRE(size_bits >= 1 && size_bits <= 32, "rans_model_create: DATA_BLOCK_SIZE_BITS %u outside 1..32",
                size_bits);
    SCL_REQUIRE(RF >= 1, "rans_model_create: RANGE_FACTOR must be >= 1");
    unsigned __int128 M = 0;
    std::vector<u32> cum_v(K);
    u32 *cum = cum_v.data();
    u32 fmin = 0xFFFFFFFFu;
    for (u32 i = 0; i < K; ++i) {
        SCL_REQUIRE(h_freq[i] > 0, "rans_model_create: zero frequency for symbol %u (division by zero, rANS.py:143)", i);
        SCL_REQUIRE(M < (1ull << 31), "rans_model_create: total frequency too large");
        cum[i] = (u32)M;
        M += h_freq[i];
        if (h_freq[i] < fmin) fmin = h_freq[i];
    }
    SCL_REQUIRE(M < (1ull << 31), "rans_model_create: total frequency too large");
    unsigned __int128 L = (unsigned __int128)RF * M;
    unsigned __int128 H = (L << b) - 1;
    SCL_REQUIRE((H >> 63) == 0, "rans_model_create: H >= 2^63 overflows the reference's int64 state (quirk Q7)");
    scl_rans_model *m = new scl_rans_model();
    ::memset((void *)m, 0, sizeof(*m));
    m->device = scl_current_device();  // AFTER the memset: the batch entry points check it (scl_check_device)
    m->dev.K = K;
    m->dev.b = b;
    m->dev.size_bits = size_bits;
    m->dev.M = (u64)M;
    m->dev.RF = RF;
    m->dev.L = (u64)L;
    m->H = (u64)H;
    m->dev.nsb = scl_bit_width_u64(m->H);
    m->dev.m_log2 = ((u64)M & ((u64)M - 1)) == 0 ? (scl_bit_width_u64((u64)M) - 1) : 0xFFFFFFFFu;
    m->state32 = (m->H >> 32) == 0 && b < 32;
    // worst-case field: state H shrunk to <= RF*fmin*2^b - 1 in steps of b bits
    {
        unsigned __int128 ms = (((unsigned __int128)RF * fmin) << b) - 1, x = H;
        u32 kb = 0;
        while (x > ms) {
            x >>= b;
            kb += b;
        }
        m->max_bits_per_symbol = kb;
    }
    const u64 tab_entries = K > 256 ? K : 256;
    hipError_t e = hipMalloc((void **)&m->d_freq, tab_entries * sizeof(u32));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_cum, tab_entries * sizeof(u32));
    if (e == hipSuccess) e = hipMemcpy(m->d_freq, h_freq, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_cum, cum, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("rans_model_create: device table upload failed: %s", hipGetErrorString(e));
        scl_rans_model_destroy(m);
        return SCL_E_HIP;
    }
    m->dev.d_freq = m->d_freq;
    m->dev.d_cum = m->d_cum;
    // the tuned kernels carry symbols as bytes: alphabets above 256 run the any-parameter kernels (*_u16 entry points)
    int rc = K <= 256 ? rans_fast_build_tables(m, h_freq, cum) : SCL_OK;
    if (rc == SCL_OK && !m->fast && K <= 256) rc = rans_fastb_build_tables(m, h_freq, cum);
    if (rc != SCL_OK) {
        scl_rans_model_destroy(m);
        return rc;
    }
    *out = m;
    return SCL_OK;
}

extern "C" int scl_rans_model_create(const uint32_t *h_freq, uint32_t K, uint64_t range_factor,
                                     uint32_t num_bits_out, uint32_t size_bits, scl_rans_model **out) {
    return rans_model_build(h_freq, K, range_factor, num_bits_out, size_bits, out);
}

extern "C" void scl_rans_model_destroy(scl_rans_model *m) {
    if (!m) return;
    if (m->d_freq) (void)hipFree(m->d_freq);
    if (m->d_cum) (void)hipFree(m->d_cum);
    if (m->d_enc_tab) (void)hipFree(m->d_enc_tab);
    if (m->d_dec_tab) (void)hipFree(m->d_dec_tab);
    if (m->d_encb_tab) (void)hipFree(m->d_encb_tab);
    if (m->d_encb_aux) (void)hipFree(m->d_encb_aux);
    if (m->d_decb_tab) (void)hipFree(m->d_decb_tab);
    delete m;
}

extern "C" int scl_rans_model_info(const scl_rans_model *m, scl_rans_info *info) {
    SCL_REQUIRE(m && info, "rans_model_info: null argument");
    info->M = m->dev.M;
    info->L = m->dev.L;
    info->H = m->H;
    info->K = m->dev.K;
    info->num_state_bits = m->dev.nsb;
    info->size_bits = m->dev.size_bits;
    info->num_bits_out = m->dev.b;
    info->max_bits_per_symbol = m->max_bits_per_symbol;
    info->fast_path = m->fast | m->fastb;
    info->device = m->device;
    return SCL_OK;
}

// Which encoder a batch of n_chunks equally long, 16-byte aligned rows would run with the calling thread's current
// settings: 'L' / 'S' = the headline kernels of scl_rans_fast.hip (NUM_BITS_OUT = 1, and since round 4 NUM_BITS_OUT in
// {4, 8, 16} within their bounds) with the 256-byte-ring / slot-ring writer, 'B' = the kernels of scl_rans_fast_b.hip
// (NUM_BITS_OUT = 2 and what those bounds exclude), 'G' = the any-parameter kernels.  For tests and tools (which switch the writer with
// SCL_RANS_ENC_WRITER and want to know that the switch took).
extern "C" int scl_rans_encoder_kind(const scl_rans_model *m, uint64_t n_chunks) {
    if (!m) return 0;
    if (scl_force_generic()) return 'G';
    if (m->fast) return rf_use_slot_writer(m, n_chunks) ? 'S' : 'L';
    return m->fastb ? 'B' : 'G';
}

// (ABI 6) the two kernels a batch of n_chunks aligned, equally long rows would run, named the way rocprofv3 prints them
// (template arguments included for the tuned kernels; the any-parameter kernels by their function name) -- so that a
// bench line's "kernel" fields can be matched mechanically against a committed kernel-trace summary.
extern "C" int scl_rans_kernel_names(const scl_rans_model *m, uint64_t n_chunks, char *enc, char *dec, uint64_t cap) {
    SCL_REQUIRE(m && (enc || dec) && cap >= 96, "rans_kernel_names: null argument or a buffer below 96 bytes");
    if (!scl_force_generic() && m->fast) {
        rans_fast_kernel_names(m, n_chunks, enc, dec, (size_t)cap);
        return SCL_OK;
    }
    const bool b = !scl_force_generic() && m->fastb;
    if (enc) snprintf(enc, (size_t)cap, "%s", b ? "rans_encode_fastb_kernel" : "rans_encode_generic");
    if (dec) snprintf(dec, (size_t)cap, "%s", b ? "rans_decode_fastb_kernel" : "rans_decode_generic");
    return SCL_OK;
}

extern "C" uint64_t scl_rans_slot_bytes(const scl_rans_model *m, uint64_t n_symbols) {
    if (!m) return 0;
    const u64 bits = (u64)m->dev.size_bits + m->dev.nsb + n_symbols * (u64)m->max_bits_per_symbol;
    return scl_round_up((bits + 7) / 8 + 4, 128);
}

static int check_batch_args(const char *what, const void *m, const void *a, const void *b, const void *c,
                            const void *d, u64 stride) {
    SCL_REQUIRE(m && a && b && c && d, "%s: null pointer argument", what);
    SCL_REQUIRE(stride % 16 == 0 && stride > 0, "%s: stream stride %llu is not a positive multiple of 16", what,
                (unsigned long long)stride);
    return SCL_OK;
}

extern "C" int scl_rans_encode_batch(const scl_rans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                     const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks, uint8_t *d_out,
                                     uint64_t out_stride, uint64_t *d_out_bit_offset, uint32_t *d_out_nbits,
                                     uint32_t *d_status, void *stream) {
    int rc = check_batch_args("rans_encode_batch", m, d_sym, d_out, d_out_bit_offset, d_out_nbits, out_stride);
    if (rc) return rc;
    SCL_REQUIRE(m->dev.K <= 256, "rans_encode_batch: alphabet of %u symbols: use scl_rans_encode_batch_u16", m->dev.K);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0, "rans_encode_batch: d_out must be 16-byte aligned");
    if (int rc_dev = scl_check_device(m->device, "rans_encode_batch")) return rc_dev;
    SCL_REQUIRE(out_stride * 8 < (1ull << 32), "rans_encode_batch: slot larger than 512 MiB");
    if (n_chunks == 0) return SCL_OK;
    hipStream_t st = (hipStream_t)stream;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    const bool tuned = !scl_force_generic();
    RowRelay relay;  // rows that do not start on 16-byte boundaries are re-laid for the tuned kernels
    if (tuned && (m->fast || m->fastb) && out_stride >= scl_rans_slot_bytes(m, chunk_len))
        if (int rc_r = relay.in(d_sym, sym_stride, chunk_len, n_chunks, st)) return rc_r;
    // fast path: qualifying model, 16-byte aligned rows, and slots that cannot overflow (it has no capacity check)
    if (tuned && m->fast && ((uintptr_t)d_sym & 15) == 0 && (sym_stride & 15) == 0 &&
        out_stride >= scl_rans_slot_bytes(m, chunk_len) && out_stride < (1ull << 24))  // 256 slots within 32-bit offsets
        rans_fast_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride,
                                d_out_bit_offset, d_out_nbits, d_status, st);
    else if (tuned && m->fastb && ((uintptr_t)d_sym & 15) == 0 && (sym_stride & 15) == 0 &&
             out_stride >= scl_rans_slot_bytes(m, chunk_len))
        rans_fastb_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride,
                                 d_out_bit_offset, d_out_nbits, d_status, st);
    else if (m->state32)
        hipLaunchKernelGGL(rans_encode_generic<u32>, dim3(blocks), dim3(threads), 0, st, m->dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status);
    else
        hipLaunchKernelGGL(rans_encode_generic<u64>, dim3(blocks), dim3(threads), 0, st, m->dev, d_sym, sym_stride,
                           d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_rans_decode_batch(const scl_rans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                     const uint64_t *d_bit_offset, const uint32_t *d_in_nbits, uint64_t n_chunks,
                                     uint8_t *d_out_sym, uint64_t out_stride, uint32_t out_cap, uint32_t *d_out_lens,
                                     uint32_t *d_consumed, uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "rans_decode_batch: null pointer argument");
    SCL_REQUIRE(m->dev.K <= 256, "rans_decode_batch: alphabet of %u symbols: use scl_rans_decode_batch_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "rans_decode_batch")) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0, "rans_decode_batch: d_in must be 4-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    hipStream_t st = (hipStream_t)stream;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    const bool tuned = !scl_force_generic();
    RowRelay relay;  // output rows the tuned kernels cannot store to go through aligned scratch and are copied back
    if (tuned && (m->fast || m->fastb) && ((uintptr_t)d_in & 15) == 0)
        if (int rc_r = relay.out_begin(d_out_sym, out_stride, out_cap, n_chunks, st)) return rc_r;
    if (tuned && m->fast && ((uintptr_t)d_in & 15) == 0 && ((uintptr_t)d_out_sym & 15) == 0 && (out_stride & 15) == 0)
        rans_fast_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                                out_cap, d_out_lens, d_consumed, d_status, st);
    else if (tuned && m->fastb && ((uintptr_t)d_in & 15) == 0 && ((uintptr_t)d_out_sym & 15) == 0 && (out_stride & 15) == 0)
        rans_fastb_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                                 out_cap, d_out_lens, d_consumed, d_status, st);
    else if (m->state32)
        hipLaunchKernelGGL(rans_decode_generic<u32>, dim3(blocks), dim3(threads), 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens, d_consumed,
                           d_status);
    else
        hipLaunchKernelGGL(rans_decode_generic<u64>, dim3(blocks), dim3(threads), 0, st, m->dev, d_in, in_size_bytes,
                           d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens, d_consumed,
                           d_status);
    SCL_HIP_TRY(hipGetLastError());
    return relay.out_end(d_out_lens);
}

// ---- uint16 symbol indices: alphabets up to 65536 (any model; the any-parameter kernels) -------------------
// ---- wave-striped slots (ABI version 8; scl_ans_fast_io.h: AnsBackWriterT / AnsBitReaderT) -----------------------------
// The same streams at the same LOGICAL bit positions, the 64 slots of a wave interleaved in 16-byte pieces in memory.
// Only the tuned kernels have a striped form: a model they do not serve is refused (scl_rans_striped_ok says so up front),
// and so is a call made while the calling thread keeps the tuned kernels out (scl_set_any_parameter_kernels).
int rans_striped_encode(const char *what, const scl_rans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                        u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                        u32 *d_status, hipStream_t st) {
    SCL_REQUIRE(m->fast && m->dev.K <= 256, "%s: this model is not served by the striped kernels (see scl_*_striped_ok)", what);
    SCL_REQUIRE(!scl_force_generic(), "%s: the calling thread keeps the tuned kernels out; striped slots have no other", what);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0, "%s: d_out must be 16-byte aligned", what);
    SCL_REQUIRE(out_stride >= scl_rans_slot_bytes(m, chunk_len) && out_stride < (1ull << 24),
                "%s: out_stride %llu: striped slots need scl_*_slot_bytes(chunk_len) <= out_stride < 2^24", what,
                (unsigned long long)out_stride);
    if (n_chunks == 0) return SCL_OK;
    RowRelay relay;  // rows that do not start on 16-byte boundaries are re-laid
    if (int rc_r = relay.in(d_sym, sym_stride, chunk_len, n_chunks, st)) return rc_r;
    if (!scl_rows_aligned(d_sym, sym_stride)) {
        scl_set_error("%s: out of device memory re-laying unaligned symbol rows (hipMallocAsync failed)", what);
        return SCL_E_ALLOC;
    }
    rans_fast_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_bit_off, d_nbits,
                            d_status, st, true);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

int rans_striped_decode(const char *what, const scl_rans_model *m, const u8 *d_in, u64 in_stride, const u64 *d_bit_off,
                        const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap, u32 *d_out_lens,
                        u32 *d_consumed, u32 *d_status, hipStream_t st) {
    SCL_REQUIRE(m->fast && m->dev.K <= 256, "%s: this model is not served by the striped kernels (see scl_*_striped_ok)", what);
    SCL_REQUIRE(!scl_force_generic(), "%s: the calling thread keeps the tuned kernels out; striped slots have no other", what);
    SCL_REQUIRE(((uintptr_t)d_in & 15) == 0 && in_stride % 16 == 0 && in_stride > 0 && in_stride < (1ull << 24),
                "%s: d_in must be 16-byte aligned and in_stride a multiple of 16 below 2^24", what);
    if (n_chunks == 0) return SCL_OK;
    RowRelay relay;  // output rows the kernels cannot store to go through aligned scratch and are copied back
    if (int rc_r = relay.out_begin(d_out_sym, out_stride, out_cap, n_chunks, st)) return rc_r;
    if (!scl_rows_aligned(d_out_sym, out_stride)) {
        scl_set_error("%s: out of device memory re-laying unaligned output rows (hipMallocAsync failed)", what);
        return SCL_E_ALLOC;
    }
    rans_fast_decode_launch(m, d_in, in_stride, d_bit_off, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                            d_out_lens, d_consumed, d_status, st, true);
    SCL_HIP_TRY(hipGetLastError());
    return relay.out_end(d_out_lens);
}

extern "C" int scl_rans_striped_ok(const scl_rans_model *m) { return (m && m->fast && m->dev.K <= 256) ? 1 : 0; }

extern "C" int scl_rans_kernel_names_striped(const scl_rans_model *m, uint64_t n_chunks, char *enc, char *dec,
                                             uint64_t cap) {
    SCL_REQUIRE(m && (enc || dec) && cap >= 96, "rans_kernel_names_striped: null argument or a buffer below 96 bytes");
    SCL_REQUIRE(scl_rans_striped_ok(m), "rans_kernel_names_striped: this model is not served by the striped kernels");
    rans_fast_kernel_names(m, n_chunks, enc, dec, (size_t)cap, true);
    return SCL_OK;
}

extern "C" int scl_rans_encode_batch_striped(const scl_rans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                             const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                             uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                             uint32_t *d_out_nbits, uint32_t *d_status, void *stream) {
    int rc = check_batch_args("rans_encode_batch_striped", m, d_sym, d_out, d_out_bit_offset, d_out_nbits, out_stride);
    if (rc) return rc;
    if (int rc_dev = scl_check_device(m->device, "rans_encode_batch_striped")) return rc_dev;
    return rans_striped_encode("rans_encode_batch_striped", m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out,
                               out_stride, d_out_bit_offset, d_out_nbits, d_status, (hipStream_t)stream);
}

extern "C" int scl_rans_decode_batch_striped(const scl_rans_model *m, const uint8_t *d_in, uint64_t in_stride,
                                             const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                             uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                                             uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                             uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "rans_decode_batch_striped: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "rans_decode_batch_striped")) return rc_dev;
    return rans_striped_decode("rans_decode_batch_striped", m, d_in, in_stride, d_bit_offset, d_in_nbits, n_chunks,
                               d_out_sym, out_stride, out_cap, d_out_lens, d_consumed, d_status, (hipStream_t)stream);
}

extern "C" int scl_rans_encode_batch_u16(const scl_rans_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                                         const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                         uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                         uint32_t *d_out_nbits, uint32_t *d_status, void *stream) {
    int rc = check_batch_args("rans_encode_batch_u16", m, d_sym, d_out, d_out_bit_offset, d_out_nbits, out_stride);
    if (rc) return rc;
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_sym & 1) == 0,
                "rans_encode_batch_u16: d_out must be 16-byte aligned, d_sym 2-byte aligned");
    if (int rc_dev = scl_check_device(m->device, "rans_encode_batch_u16")) return rc_dev;
    SCL_REQUIRE(out_stride * 8 < (1ull << 32), "rans_encode_batch_u16: slot larger than 512 MiB");
    if (n_chunks == 0) return SCL_OK;
    hipStream_t st = (hipStream_t)stream;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    if (m->state32)
        hipLaunchKernelGGL((rans_encode_generic<u32, u16>), dim3(blocks), dim3(threads), 0, st, m->dev, d_sym,
                           sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits,
                           d_status);
    else
        hipLaunchKernelGGL((rans_encode_generic<u64, u16>), dim3(blocks), dim3(threads), 0, st, m->dev, d_sym,
                           sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits,
                           d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_rans_decode_batch_u16(const scl_rans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                         const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                         uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                                         uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                         uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "rans_decode_batch_u16: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "rans_decode_batch_u16")) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0 && ((uintptr_t)d_out_sym & 1) == 0,
                "rans_decode_batch_u16: d_in must be 4-byte aligned, d_out_sym 2-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    hipStream_t st = (hipStream_t)stream;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    if (m->state32)
        hipLaunchKernelGGL((rans_decode_generic<u32, u16>), dim3(blocks), dim3(threads), 0, st, m->dev, d_in,
                           in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
    else
        hipLaunchKernelGGL((rans_decode_generic<u64, u16>), dim3(blocks), dim3(threads), 0, st, m->dev, d_in,
                           in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap,
                           d_out_lens, d_consumed, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

// ---- single-chunk host drivers ------------------------------------------------------------------------
static int rans_run_enc(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                        u32 *d_nbits, u32 *d_status, void *, u64) {
    return scl_rans_encode_batch((const scl_rans_model *)model, d_sym, n, nullptr, n, 1, d_out, out_stride, d_bit_off,
                                 d_nbits, d_status, nullptr);
}
static u64 rans_slot(const void *model, u64 n) { return scl_rans_slot_bytes((const scl_rans_model *)model, n); }
static int rans_run_dec(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
                        u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *, u64) {
    return scl_rans_decode_batch((const scl_rans_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1, d_out_sym,
                                 scl_round_up((u64)out_cap + 1, 16), out_cap, d_out_len, d_consumed, d_status, nullptr);
}

extern "C" int scl_rans_encode_host(const scl_rans_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                                    uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {rans_run_enc, rans_slot, nullptr};
    return scl_host_encode_one(call, m, h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_rans_decode_host(const scl_rans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                    uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {rans_run_dec, nullptr};
    return scl_host_decode_one(call, m, h_in, in_nbits, h_out_sym, out_cap, n_out, consumed);
}

static int rans_run_enc16(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                          u32 *d_nbits, u32 *d_status, void *, u64) {
    return scl_rans_encode_batch_u16((const scl_rans_model *)model, (const u16 *)d_sym, n, nullptr, n, 1, d_out,
                                     out_stride, d_bit_off, d_nbits, d_status, nullptr);
}
static int rans_run_dec16(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
                          u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *, u64) {
    return scl_rans_decode_batch_u16((const scl_rans_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1,
                                     (u16 *)d_out_sym, (u64)out_cap + 1, out_cap, d_out_len, d_consumed, d_status,
                                     nullptr);
}

extern "C" int scl_rans_encode_host_u16(const scl_rans_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                                        uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {rans_run_enc16, rans_slot, nullptr};
    call.sym_bytes = 2;
    return scl_host_encode_one(call, m, (const u8 *)h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_rans_decode_host_u16(const scl_rans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                        uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {rans_run_dec16, nullptr};
    call.sym_bytes = 2;
    return scl_host_decode_one(call, m, h_in, in_nbits, (u8 *)h_out_sym, out_cap, n_out, consumed);
}
