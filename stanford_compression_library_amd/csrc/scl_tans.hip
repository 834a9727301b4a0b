// scl_tans.hip -- batched tANS (table ANS = cached rANS) for gfx950, one wavefront lane per chunk.
//
// Replaces reference scl/compressors/tANS.py:
//   tANSParams :31-53 (M power of two, NUM_BITS_OUT == 1)
//   tANSEncoder: table builders :74-110, encode_symbol :126-157, encode_block :159-193
//   tANSDecoder: table builders :208-226, decode_symbol :239-250, decode_block :252-279
// The five lookup tables are built ON DEVICE by one pass over the state range (the reference calls
// the rANS base step once per entry from Python: 0.36 s at M = 4096).
// Table layout (flat, no hashing):
//   enc[RF*c[s] + (x_shrunk - RF*f[s])] = base_encode_step_table[(s, x_shrunk)]   RF*M entries, u32
//   nbits[s], thresh[s]                 = shrink_state_{num_out_bits_base,thresh}_table   K entries
//   dec_sym[x - L], dec_xs[x - L]       = base_decode_step_table[x] = (s, x_shrunk)  RF*M entries
//   expand_state_num_bits_table[x_shrunk] = NUM_STATE_BITS - bit_width(x_shrunk) is computed with clz.
// The bit stream is identical to rANS with the same parameters (same layout as scl_rans.hip).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "scl_tans_internal.h"
#include "scl_rans_internal.h"

__device__ __forceinline__ u32 tans_find_bin(const u32 *cum, u32 K, u32 slot) {
    u32 lo = 0, hi = K;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (cum[mid] <= slot)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__device__ __forceinline__ u32 tans_bit_width(u32 x) { return x == 0 ? 1u : 32u - (u32)__builtin_clz(x); }

// one thread per table entry
__global__ void tans_build_tables(u32 K, u32 M, u32 RF, u32 m_log2, u32 nsb, const u32 *__restrict__ freq,
                                  const u32 *__restrict__ cum, u32 *__restrict__ enc, u32 *__restrict__ nbits,
                                  u32 *__restrict__ thresh, u32 *__restrict__ dec_sym, u32 *__restrict__ dec_xs) {
    const u32 L = RF * M;
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < K) {
        // shrink_state_num_out_bits_base, tANS.py:74-86
        const u32 max_shrunk = 2 * RF * freq[e] - 1;
        const u32 nb = nsb - tans_bit_width(max_shrunk);
        nbits[e] = nb;
        thresh[e] = (max_shrunk + 1) << nb;
    }
    if (e >= L) return;
    {  // build_base_encode_step_table, tANS.py:88-99
        const u32 s = tans_find_bin(cum, K, e / RF);
        const u32 f = freq[s], c = cum[s];
        const u32 xs = e - RF * c + RF * f;
        enc[e] = (xs / f) * M + c + (xs % f);
    }
    {  // build_rans_base_decode_table, tANS.py:208-215
        const u32 x = L + e;
        const u32 block_id = x >> m_log2, slot = x & (M - 1);
        const u32 s = tans_find_bin(cum, K, slot);
        dec_sym[e] = s;
        dec_xs[e] = block_id * freq[s] + slot - cum[s];
    }
}

// ---- encode ---------------------------------------------------------------------------------------------
// SYM = u8: alphabets up to 256, per-symbol tables staged in LDS.  SYM = u16 (the *_u16 entry points): alphabets up to
// 65536, every table read where it is in device memory; strides count SYMBOLS in both.
template <typename SYM = u8>
__global__ void __launch_bounds__(256) tans_encode_kernel(TansDev P, const SYM *__restrict__ sym, u64 sym_stride,
                                                         const u32 *__restrict__ lens, u32 chunk_len, u64 n_chunks,
                                                         u8 *__restrict__ out, u64 out_stride,
                                                         u64 *__restrict__ out_bit_off, u32 *__restrict__ out_nbits,
                                                         u32 *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) u32 s_mem[];
    u32 *s_nbits = s_mem;          // [256]
    u32 *s_thresh = s_mem + 256;   // [256]
    int *s_off = (int *)(s_mem + 512);  // [256]  RF*c[s] - RF*f[s]
    u32 *s_enc = s_mem + 768;      // [L] when lds_tables
    constexpr bool WIDE_SYM = sizeof(SYM) > 1;
    if (!WIDE_SYM) {
        for (u32 i = threadIdx.x; i < P.K; i += blockDim.x) {
            s_nbits[i] = P.d_nbits[i];
            s_thresh[i] = P.d_thresh[i];
            s_off[i] = (int)(P.RF * P.d_cum[i]) - (int)(P.RF * P.d_freq[i]);
        }
        if (P.lds_tables) scl_load_table(s_enc, P.d_enc, P.L);
        __syncthreads();
    }
    const u32 *enc = (!WIDE_SYM && P.lds_tables) ? s_enc : P.d_enc;
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 n = lens ? lens[c] : chunk_len;
    const SYM *src = sym + c * sym_stride;
    BackBitWriter w;
    w.init(out + c * out_stride, out_stride);
    u32 st = 0;
    u32 x = P.L;
    for (u32 i = 0; i < n; ++i) {
        u32 s = src[i];
        if (s >= P.K) {
            st |= SCL_ST_SYMBOL;
            s = 0;
        }
        // encode_symbol, tANS.py:126-157: two table reads, one compare, one shift, one table read
        u32 nb1, thr;
        int off;
        if (WIDE_SYM) {
            nb1 = P.d_nbits[s];
            thr = P.d_thresh[s];
            off = (int)(P.RF * P.d_cum[s]) - (int)(P.RF * P.d_freq[s]);
        } else {
            nb1 = s_nbits[s];
            thr = s_thresh[s];
            off = s_off[s];
        }
        const u32 nb = nb1 + (x >= thr ? 1u : 0u);
        if (nb) w.put(x & ((1u << nb) - 1u), nb);
        x >>= nb;
        x = enc[(u32)(off + (int)x)];
    }
    w.put(x, P.nsb);
    if (P.size_bits < 32 && (n >> P.size_bits)) st |= SCL_ST_SIZE;
    w.put(n, P.size_bits);
    const u64 total = w.finish();
    if (w.overflow) st |= SCL_ST_CAPACITY;
    out_bit_off[c] = (c + 1) * out_stride * 8 - total;
    out_nbits[c] = (u32)total;
    if (status) status[c] = st;
}

// ---- decode ---------------------------------------------------------------------------------------------
template <typename SYM = u8>
__global__ void __launch_bounds__(256) tans_decode_kernel(TansDev P, const u8 *__restrict__ in, u64 in_size_bytes,
                                                         const u64 *__restrict__ bit_off,
                                                         const u32 *__restrict__ in_nbits, u64 n_chunks,
                                                         SYM *__restrict__ out_sym, u64 out_stride, u32 out_cap,
                                                         u32 *__restrict__ out_lens, u32 *__restrict__ consumed,
                                                         u32 *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) u32 s_mem[];
    u32 *s_sym = s_mem;        // [L] when lds_tables
    u32 *s_xs = s_mem + P.L;   // [L]
    if (P.lds_tables) {
        scl_load_table(s_sym, P.d_dec_sym, P.L);
        scl_load_table(s_xs, P.d_dec_xs, P.L);
    }
    __syncthreads();
    const u32 *dec_sym = P.lds_tables ? s_sym : P.d_dec_sym;
    const u32 *dec_xs = P.lds_tables ? s_xs : P.d_dec_xs;
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    BitReader r;
    r.init(in, in_size_bytes, bit_off[c], in_nbits[c]);
    const u64 start = r.pos;
    u32 st = 0;
    u32 n = r.get(P.size_bits);
    u32 x = r.get(P.nsb);
    if (r.truncated) {
        st |= SCL_ST_TRUNCATED;
        n = 0;
    } else if (x < P.L || x >= 2 * P.L) {
        st |= SCL_ST_STATE;  // KeyError on base_decode_step_table in the reference
        n = 0;
    }
    out_lens[c] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    SYM *dst = out_sym + c * out_stride;
    const u32 st_header = st;
    for (u32 i = n; i-- > 0;) {
        // decode_symbol, tANS.py:239-250
        const u32 s = dec_sym[x - P.L];
        const u32 xs = dec_xs[x - P.L];
        const u32 nb = P.nsb - tans_bit_width(xs);  // expand_state_num_bits_table :217-226
        const u32 rem = nb ? r.get(nb) : 0u;
        x = (xs << nb) + rem;
        dst[i] = (SYM)s;
        if (r.truncated) break;
    }
    if (r.truncated) st |= SCL_ST_TRUNCATED;
    else if (st_header == 0 && x != P.L) st |= SCL_ST_STATE;  // assert state == INITIAL_STATE, tANS.py:277
    consumed[c] = (u32)(r.pos - start);
    if (status) status[c] = st;
}

// ---- host API -------------------------------------------------------------------------------------------
#define TANS_LDS_BUDGET (64u * 1024u)

// SCL_TANS_KERNELS=table in the environment: models whose tables fit LDS run the lookup-table kernels
// (scl_tans_fast.hip) instead of the table-free rANS kernels (tests run every tANS case both ways)
static bool tans_table_kernels_forced() {
    const char *e = getenv("SCL_TANS_KERNELS");
    return e && e[0] == 't';
}

// builds base_encode_step_table / base_decode_step_table / the per-symbol tables on the device, once
static int tans_ensure_tables(const scl_tans_model *cm) {
    scl_tans_model *m = const_cast<scl_tans_model *>(cm);
    std::lock_guard<std::mutex> guard(m->build_lock);
    if (m->built) return SCL_OK;
    const u64 L = m->dev.L;
    const u32 K = m->dev.K;
    hipError_t e = hipSuccess;
    u32 **tabs[] = {&m->d_enc, &m->d_dec_sym, &m->d_dec_xs};
    for (u32 **p : tabs)
        if (e == hipSuccess && !*p) e = hipMalloc((void **)p, L * sizeof(u32));
    if (e == hipSuccess) {
        const u32 n_thr = (u32)(L > K ? L : K);
        hipLaunchKernelGGL(tans_build_tables, dim3((n_thr + 255) / 256), dim3(256), 0, 0, K, m->dev.M, m->dev.RF,
                           m->dev.m_log2, m->dev.nsb, m->d_freq, m->d_cum, m->d_enc, m->d_nbits, m->d_thresh,
                           m->d_dec_sym, m->d_dec_xs);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        scl_set_error("tans: device table build (%llu entries) failed: %s", (unsigned long long)L, hipGetErrorString(e));
        return SCL_E_HIP;
    }
    m->dev.d_enc = m->d_enc;
    m->dev.d_dec_sym = m->d_dec_sym;
    m->dev.d_dec_xs = m->d_dec_xs;
    m->built = 1;
    return SCL_OK;
}

extern "C" int scl_tans_model_create(const uint32_t *h_freq, uint32_t K, uint64_t range_factor, uint32_t size_bits,
                                     scl_tans_model **out) {
    SCL_REQUIRE(out, "tans_model_create: null output");
    *out = nullptr;
    SCL_REQUIRE(h_freq && K >= 1 && K <= SCL_MAX_ALPHABET, "tans_model_create: alphabet size %u outside 1..65536", K);
    SCL_REQUIRE(size_bits >= 1 && size_bits <= 32, "tans_model_create: DATA_BLOCK_SIZE_BITS %u outside 1..32",
                size_bits);
    SCL_REQUIRE(range_factor >= 1 && range_factor <= (1ull << 30), "tans_model_create: RANGE_FACTOR outside 1..2^30");
    u64 M = 0;
    std::vector<u32> cum_v(K);
    u32 *cum = cum_v.data(), fmin = 0xFFFFFFFFu;
    for (u32 i = 0; i < K; ++i) {
        SCL_REQUIRE(h_freq[i] > 0, "tans_model_create: zero frequency for symbol %u", i);
        cum[i] = (u32)M;
        M += h_freq[i];
        SCL_REQUIRE(M <= (1ull << 30), "tans_model_create: total frequency too large");
        if (h_freq[i] < fmin) fmin = h_freq[i];
    }
    SCL_REQUIRE((M & (M - 1)) == 0,
                "tans_model_create: total frequency %llu is not a power of two (assert at tANS.py:42-44)",
                (unsigned long long)M);
    const u64 L = range_factor * M;
    // companion rANS model (same stream, no tables): see scl_tans_model::rans
    // (round 3: also for tables that fit LDS -- on this machine the table-free kernels are the faster way to write the
    // same stream, 0.55 / 0.56 ms against 0.72 / 0.58 ms per GiB for the LDS-table kernels at RANGE_FACTOR = 1, whose
    // per-symbol gathers keep the LDS pipe 82 % busy; SCL_TANS_KERNELS=table keeps the lookup-table kernels in charge)
    scl_rans_model *rans = nullptr;
    if (L <= (1ull << 30) && K <= 256) {
        if (scl_rans_model_create(h_freq, K, range_factor, 1, size_bits, &rans) == SCL_OK && rans && !rans->fast) {
            scl_rans_model_destroy(rans);
            rans = nullptr;
        }
    }
    if (L > (1ull << 26) && !rans) {
        scl_set_error("tans_model_create: RANGE_FACTOR*M = %llu exceeds the 2^26-entry table budget and the model is "
                      "outside the table-free rANS kernels (M <= 4096, RANGE_FACTOR a power of two <= 2^23)",
                      (unsigned long long)L);
        return SCL_E_PARAM;
    }
    scl_tans_model *m = new scl_tans_model();
    m->device = scl_current_device();
    m->rans = rans;
    m->tables = (L <= (1ull << 26)) ? 1u : 0u;
    m->dev.K = K;
    m->dev.size_bits = size_bits;
    m->dev.M = (u32)M;
    m->dev.RF = (u32)range_factor;
    m->dev.L = (u32)L;
    m->dev.nsb = scl_bit_width_u64(2 * L - 1);
    m->dev.m_log2 = scl_bit_width_u64(M) - 1;
    {
        u64 ms = 2 * range_factor * fmin - 1, x = 2 * L - 1;
        u32 kb = 0;
        while (x > ms) {
            x >>= 1;
            ++kb;
        }
        m->max_bits_per_symbol = kb;
    }
    hipError_t e = hipSuccess;
    auto alloc = [&](u32 **p, u64 n) {
        if (e == hipSuccess) e = hipMalloc((void **)p, (n ? n : 1) * sizeof(u32));
    };
    const u64 tab_entries = K > 256 ? K : 256;
    alloc(&m->d_freq, tab_entries);
    alloc(&m->d_cum, tab_entries);
    alloc(&m->d_nbits, tab_entries);
    alloc(&m->d_thresh, tab_entries);
    if (e == hipSuccess) e = hipMemcpy(m->d_freq, h_freq, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_cum, cum, K * sizeof(u32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("tans_model_create: device table upload failed: %s", hipGetErrorString(e));
        scl_tans_model_destroy(m);
        return SCL_E_HIP;
    }
    m->dev.d_freq = m->d_freq;
    m->dev.d_cum = m->d_cum;
    m->dev.d_nbits = m->d_nbits;
    m->dev.d_thresh = m->d_thresh;
    m->dev.lds_tables = (m->tables && 2 * L * sizeof(u32) <= TANS_LDS_BUDGET) ? 1u : 0u;
    // A model with a companion rANS handle is served by the table-free kernels: its 3 x L-entry tables (768 MiB at
    // 2^26 entries) are only built if somebody asks for them (scl_tans_model_tables, or rows the tuned kernels
    // cannot take) -- tans_ensure_tables.
    const bool small = m->tables && L <= 8192;  // LDS-sized tables are built at once (the tuned table kernels need them)
    int rc = (m->tables && (!m->rans || small)) ? tans_ensure_tables(m) : SCL_OK;
    if (rc == SCL_OK && m->tables && (!m->rans || small) && K <= 256) rc = tans_fast_build_tables(m, h_freq, cum);
    if (rc != SCL_OK) {
        scl_tans_model_destroy(m);
        return rc;
    }
    *out = m;
    return SCL_OK;
}

extern "C" void scl_tans_model_destroy(scl_tans_model *m) {
    if (!m) return;
    u32 *ptrs[] = {m->d_freq, m->d_cum, m->d_enc, m->d_nbits, m->d_thresh, m->d_dec_sym, m->d_dec_xs};
    for (u32 *p : ptrs)
        if (p) (void)hipFree(p);
    if (m->d_fenc_sym) (void)hipFree(m->d_fenc_sym);
    if (m->d_fenc_tab) (void)hipFree(m->d_fenc_tab);
    if (m->d_fdec_tab) (void)hipFree(m->d_fdec_tab);
    if (m->rans) scl_rans_model_destroy(m->rans);
    delete m;
}

extern "C" int scl_tans_model_info(const scl_tans_model *m, scl_rans_info *info) {
    SCL_REQUIRE(m && info, "tans_model_info: null argument");
    info->M = m->dev.M;
    info->L = m->dev.L;
    info->H = 2ull * m->dev.L - 1;
    info->K = m->dev.K;
    info->num_state_bits = m->dev.nsb;
    info->size_bits = m->dev.size_bits;
    info->num_bits_out = 1;
    info->max_bits_per_symbol = m->max_bits_per_symbol;
    info->fast_path = (m->fast || m->rans) ? 1u : m->dev.lds_tables;
    info->device = m->device;
    return SCL_OK;
}

extern "C" uint64_t scl_tans_slot_bytes(const scl_tans_model *m, uint64_t n_symbols) {
    if (!m) return 0;
    const u64 bits = (u64)m->dev.size_bits + m->dev.nsb + n_symbols * (u64)m->max_bits_per_symbol;
    return scl_round_up((bits + 7) / 8 + 4, 128);
}

extern "C" int scl_tans_model_tables(const scl_tans_model *m, uint32_t *h_enc, uint32_t *h_nbits, uint32_t *h_thresh,
                                     uint32_t *h_dec_sym, uint32_t *h_dec_xs) {
    SCL_REQUIRE(m, "tans_model_tables: null model");
    SCL_REQUIRE(m->tables, "tans_model_tables: RANGE_FACTOR*M = %llu entries are above the 2^26-entry budget; this "
                           "model runs on the table-free rANS kernels", (unsigned long long)m->dev.L);
    if (int rc = tans_ensure_tables(m)) return rc;
    const u64 Lb = (u64)m->dev.L * sizeof(u32), Kb = (u64)m->dev.K * sizeof(u32);
    if (h_enc) SCL_HIP_TRY(hipMemcpy(h_enc, m->d_enc, Lb, hipMemcpyDeviceToHost));
    if (h_nbits) SCL_HIP_TRY(hipMemcpy(h_nbits, m->d_nbits, Kb, hipMemcpyDeviceToHost));
    if (h_thresh) SCL_HIP_TRY(hipMemcpy(h_thresh, m->d_thresh, Kb, hipMemcpyDeviceToHost));
    if (h_dec_sym) SCL_HIP_TRY(hipMemcpy(h_dec_sym, m->d_dec_sym, Lb, hipMemcpyDeviceToHost));
    if (h_dec_xs) SCL_HIP_TRY(hipMemcpy(h_dec_xs, m->d_dec_xs, Lb, hipMemcpyDeviceToHost));
    return SCL_OK;
}

// (ABI 6) as scl_rans_kernel_names: a tANS model the table-free rANS kernels can serve runs on them (same stream) unless
// SCL_TANS_KERNELS=table is set
extern "C" int scl_tans_kernel_names(const scl_tans_model *m, uint64_t n_chunks, char *enc, char *dec, uint64_t cap) {
    SCL_REQUIRE(m && (enc || dec) && cap >= 96, "tans_kernel_names: null argument or a buffer below 96 bytes");
    const bool tuned = !scl_force_generic();
    const bool table_first = m->fast && tans_table_kernels_forced();
    if ((tuned || !m->tables) && m->rans && !(table_first && tuned)) {
        rans_fast_kernel_names(m->rans, n_chunks, enc, dec, (size_t)cap);
        return SCL_OK;
    }
    const bool f = tuned && m->fast;
    if (enc) snprintf(enc, (size_t)cap, "%s", f ? "tans_encode_fast_kernel" : "tans_encode_kernel");
    if (dec) snprintf(dec, (size_t)cap, "%s", f ? "tans_decode_fast_kernel" : "tans_decode_kernel");
    return SCL_OK;
}

// ---- wave-striped slots (ABI version 8): tANS models the table-free rANS kernels serve, on their striped form ----------
extern "C" int scl_tans_striped_ok(const scl_tans_model *m) {
    return (m && m->rans && m->rans->fast && m->dev.K <= 256) ? 1 : 0;
}

extern "C" int scl_tans_kernel_names_striped(const scl_tans_model *m, uint64_t n_chunks, char *enc, char *dec,
                                             uint64_t cap) {
    SCL_REQUIRE(m && (enc || dec) && cap >= 96, "tans_kernel_names_striped: null argument or a buffer below 96 bytes");
    SCL_REQUIRE(scl_tans_striped_ok(m), "tans_kernel_names_striped: this model is not served by the striped kernels");
    rans_fast_kernel_names(m->rans, n_chunks, enc, dec, (size_t)cap, true);
    return SCL_OK;
}

extern "C" int scl_tans_encode_batch_striped(const scl_tans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                             const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                             uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                             uint32_t *d_out_nbits, uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits, "tans_encode_batch_striped: null pointer argument");
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0, "tans_encode_batch_striped: bad out_stride %llu",
                (unsigned long long)out_stride);
    if (int rc_dev = scl_check_device(m->device, "tans_encode_batch_striped")) return rc_dev;
    SCL_REQUIRE(scl_tans_striped_ok(m), "tans_encode_batch_striped: this model is not served by the striped kernels");
    return rans_striped_encode("tans_encode_batch_striped", m->rans, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out,
                               out_stride, d_out_bit_offset, d_out_nbits, d_status, (hipStream_t)stream);
}

extern "C" int scl_tans_decode_batch_striped(const scl_tans_model *m, const uint8_t *d_in, uint64_t in_stride,
                                             const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                             uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                                             uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                             uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "tans_decode_batch_striped: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "tans_decode_batch_striped")) return rc_dev;
    SCL_REQUIRE(scl_tans_striped_ok(m), "tans_decode_batch_striped: this model is not served by the striped kernels");
    return rans_striped_decode("tans_decode_batch_striped", m->rans, d_in, in_stride, d_bit_offset, d_in_nbits, n_chunks,
                               d_out_sym, out_stride, out_cap, d_out_lens, d_consumed, d_status, (hipStream_t)stream);
}

extern "C" int scl_tans_encode_batch(const scl_tans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                     const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks, uint8_t *d_out,
                                     uint64_t out_stride, uint64_t *d_out_bit_offset, uint32_t *d_out_nbits,
                                     uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits, "tans_encode_batch: null pointer argument");
    SCL_REQUIRE(m->dev.K <= 256, "tans_encode_batch: alphabet of %u symbols: use scl_tans_encode_batch_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "tans_encode_batch")) return rc_dev;
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride * 8 < (1ull << 32),
                "tans_encode_batch: bad out_stride %llu", (unsigned long long)out_stride);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0, "tans_encode_batch: d_out must be 16-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    const bool tuned = !scl_force_generic();
    const bool table_first = m->fast && tans_table_kernels_forced();  // else the table-free kernels when the model has them
    RowRelay relay;  // rows that do not start on 16-byte boundaries are re-laid for the tuned kernels
    if ((tuned || !m->tables) && (m->fast || m->rans))
        if (int rc_r = relay.in(d_sym, sym_stride, chunk_len, n_chunks, (hipStream_t)stream)) return rc_r;
    const bool rows_ok = ((uintptr_t)d_sym & 15) == 0 && (sym_stride & 15) == 0;
    const bool fast_ok = tuned && m->fast && rows_ok && out_stride >= scl_tans_slot_bytes(m, chunk_len);
    // same stream from the table-free rANS kernels (32-bit slot offsets per workgroup)
    const bool rans_ok = (tuned || !m->tables) && m->rans && rows_ok &&
                         out_stride >= scl_rans_slot_bytes(m->rans, chunk_len) && out_stride < (1ull << 24);
    if (rans_ok && !(table_first && fast_ok)) {
        rans_fast_encode_launch(m->rans, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride,
                                d_out_bit_offset, d_out_nbits, d_status, (hipStream_t)stream);
        SCL_HIP_TRY(hipGetLastError());
        return SCL_OK;
    }
    if (fast_ok) {
        tans_fast_encode_launch(m, d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset,
                                d_out_nbits, d_status, (hipStream_t)stream);
        SCL_HIP_TRY(hipGetLastError());
        return SCL_OK;
    }
    if (!m->tables && relay.failed) {  // the rows WERE the problem, and the scratch to re-lay them could not be had
        scl_set_error("tans_encode_batch: out of device memory re-laying unaligned symbol rows (hipMallocAsync failed) and "
                      "this model has no lookup tables for the any-parameter kernels");
        return SCL_E_ALLOC;
    }
    SCL_REQUIRE(m->tables, "tans_encode_batch: this model has no lookup tables (RANGE_FACTOR*M > 2^26); it needs "
                           "16-byte aligned symbol rows and slots of scl_tans_slot_bytes");
    if (int rc = tans_ensure_tables(m)) return rc;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    const u32 lds = (768 + (m->dev.lds_tables ? m->dev.L : 0)) * sizeof(u32);
    hipLaunchKernelGGL(tans_encode_kernel<u8>, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, m->dev, d_sym,
                       sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits,
                       d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_tans_decode_batch(const scl_tans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                     const uint64_t *d_bit_offset, const uint32_t *d_in_nbits, uint64_t n_chunks,
                                     uint8_t *d_out_sym, uint64_t out_stride, uint32_t out_cap, uint32_t *d_out_lens,
                                     uint32_t *d_consumed, uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "tans_decode_batch: null pointer argument");
    SCL_REQUIRE(m->dev.K <= 256, "tans_decode_batch: alphabet of %u symbols: use scl_tans_decode_batch_u16", m->dev.K);
    if (int rc_dev = scl_check_device(m->device, "tans_decode_batch")) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0, "tans_decode_batch: d_in must be 4-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    const bool tuned = !scl_force_generic();
    RowRelay relay;  // output rows the tuned kernels cannot store to go through aligned scratch and are copied back
    if ((tuned || !m->tables) && (m->fast || m->rans) && ((uintptr_t)d_in & 15) == 0)
        if (int rc_r = relay.out_begin(d_out_sym, out_stride, out_cap, n_chunks, (hipStream_t)stream)) return rc_r;
    const bool table_first = m->fast && tans_table_kernels_forced();
    const bool bufs_ok = ((uintptr_t)d_in & 15) == 0 && ((uintptr_t)d_out_sym & 15) == 0 && (out_stride & 15) == 0;
    const bool fast_ok = tuned && m->fast && bufs_ok;
    const bool rans_ok = (tuned || !m->tables) && m->rans && bufs_ok;
    if (rans_ok && !(table_first && fast_ok)) {
        rans_fast_decode_launch(m->rans, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                                out_cap, d_out_lens, d_consumed, d_status, (hipStream_t)stream);
        SCL_HIP_TRY(hipGetLastError());
        return relay.out_end(d_out_lens);
    }
    if (fast_ok) {
        tans_fast_decode_launch(m, d_in, in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride,
                                out_cap, d_out_lens, d_consumed, d_status, (hipStream_t)stream);
        SCL_HIP_TRY(hipGetLastError());
        return relay.out_end(d_out_lens);
    }
    if (!m->tables && relay.failed) {
        scl_set_error("tans_decode_batch: out of device memory re-laying unaligned output rows (hipMallocAsync failed) and "
                      "this model has no lookup tables for the any-parameter kernels");
        return SCL_E_ALLOC;
    }
    SCL_REQUIRE(m->tables, "tans_decode_batch: this model has no lookup tables (RANGE_FACTOR*M > 2^26); it needs "
                           "16-byte aligned buffers");
    if (int rc = tans_ensure_tables(m)) return rc;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    const u32 lds = (m->dev.lds_tables ? 2 * m->dev.L : 4) * sizeof(u32);
    hipLaunchKernelGGL(tans_decode_kernel<u8>, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, m->dev, d_in,
                       in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                       d_consumed, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return relay.out_end(d_out_lens);
}

// ---- uint16 symbol indices: alphabets up to 65536 (lookup tables in device memory) --------------------------
extern "C" int scl_tans_encode_batch_u16(const scl_tans_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                                         const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                         uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                         uint32_t *d_out_nbits, uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_sym && d_out && d_out_bit_offset && d_out_nbits,
                "tans_encode_batch_u16: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "tans_encode_batch_u16")) return rc_dev;
    SCL_REQUIRE(out_stride % 16 == 0 && out_stride > 0 && out_stride * 8 < (1ull << 32),
                "tans_encode_batch_u16: bad out_stride %llu", (unsigned long long)out_stride);
    SCL_REQUIRE(((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_sym & 1) == 0,
                "tans_encode_batch_u16: d_out must be 16-byte aligned, d_sym 2-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    SCL_REQUIRE(m->tables, "tans_encode_batch_u16: this model has no lookup tables (RANGE_FACTOR*M > 2^26)");
    if (int rc = tans_ensure_tables(m)) return rc;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    hipLaunchKernelGGL(tans_encode_kernel<u16>, dim3(blocks), dim3(threads), 16, (hipStream_t)stream, m->dev, d_sym,
                       sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_out_bit_offset, d_out_nbits,
                       d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_tans_decode_batch_u16(const scl_tans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                         const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                         uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                                         uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                         uint32_t *d_status, void *stream) {
    SCL_REQUIRE(m && d_in && d_bit_offset && d_in_nbits && d_out_sym && d_out_lens && d_consumed,
                "tans_decode_batch_u16: null pointer argument");
    if (int rc_dev = scl_check_device(m->device, "tans_decode_batch_u16")) return rc_dev;
    SCL_REQUIRE(((uintptr_t)d_in & 3) == 0 && ((uintptr_t)d_out_sym & 1) == 0,
                "tans_decode_batch_u16: d_in must be 4-byte aligned, d_out_sym 2-byte aligned");
    if (n_chunks == 0) return SCL_OK;
    SCL_REQUIRE(m->tables, "tans_decode_batch_u16: this model has no lookup tables (RANGE_FACTOR*M > 2^26)");
    if (int rc = tans_ensure_tables(m)) return rc;
    const u32 threads = 256;
    const u32 blocks = (u32)((n_chunks + threads - 1) / threads);
    const u32 lds = (m->dev.lds_tables ? 2 * m->dev.L : 4) * sizeof(u32);
    hipLaunchKernelGGL(tans_decode_kernel<u16>, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, m->dev, d_in,
                       in_size_bytes, d_bit_offset, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,
                       d_consumed, d_status);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

// ---- single-chunk host drivers --------------------------------------------------------------------------
static int tans_run_enc(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                        u32 *d_nbits, u32 *d_status, void *, u64) {
    const scl_tans_model *m = (const scl_tans_model *)model;
    // one row: its stride is free, and a multiple of 16 lets a table-less model reach the kernels that serve it
    return scl_tans_encode_batch(m, d_sym, (m->tables && !m->rans) ? n : scl_round_up(n, 16), nullptr, n, 1, d_out, out_stride,
                                 d_bit_off, d_nbits, d_status, nullptr);
}
static u64 tans_slot(const void *model, u64 n) { return scl_tans_slot_bytes((const scl_tans_model *)model, n); }
static int tans_run_dec(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
                        u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *, u64) {
    return scl_tans_decode_batch((const scl_tans_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1, d_out_sym,
                                 scl_round_up((u64)out_cap + 1, 16), out_cap, d_out_len, d_consumed, d_status, nullptr);
}

extern "C" int scl_tans_encode_host(const scl_tans_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                                    uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {tans_run_enc, tans_slot, nullptr};
    return scl_host_encode_one(call, m, h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_tans_decode_host(const scl_tans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                    uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {tans_run_dec, nullptr};
    return scl_host_decode_one(call, m, h_in, in_nbits, h_out_sym, out_cap, n_out, consumed);
}

static int tans_run_enc16(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
                          u32 *d_nbits, u32 *d_status, void *, u64) {
    return scl_tans_encode_batch_u16((const scl_tans_model *)model, (const u16 *)d_sym, n, nullptr, n, 1, d_out,
                                     out_stride, d_bit_off, d_nbits, d_status, nullptr);
}
static int tans_run_dec16(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
                          u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *, u64) {
    return scl_tans_decode_batch_u16((const scl_tans_model *)model, d_in, in_bytes, d_bit_off, d_in_nbits, 1,
                                     (u16 *)d_out_sym, (u64)out_cap + 1, out_cap, d_out_len, d_consumed, d_status,
                                     nullptr);
}

extern "C" int scl_tans_encode_host_u16(const scl_tans_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                                        uint64_t out_cap_bytes, uint64_t *nbits) {
    HostEncodeCall call = {tans_run_enc16, tans_slot, nullptr};
    call.sym_bytes = 2;
    return scl_host_encode_one(call, m, (const u8 *)h_sym, n, h_out, out_cap_bytes, nbits);
}

extern "C" int scl_tans_decode_host_u16(const scl_tans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                        uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed) {
    HostDecodeCall call = {tans_run_dec16, nullptr};
    call.sym_bytes = 2;
    return scl_host_decode_one(call, m, h_in, in_nbits, (u8 *)h_out_sym, out_cap, n_out, consumed);
}
