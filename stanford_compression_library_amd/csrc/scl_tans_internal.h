// scl_tans_internal.h -- model layout shared by scl_tans.hip (table builder, generic kernels, host API) and
// scl_tans_fast.hip (the gfx950 fast path).  Internal to csrc/.
#pragma once
#include <mutex>
#include "scl_common.h"

struct TansDev {
    u32 K;
    u32 size_bits;
    u32 nsb;
    u32 m_log2;
    u32 M, RF, L;  // L = RF*M <= 2^30
    const u32 *d_freq;
    const u32 *d_cum;
    const u32 *d_enc;
    const u32 *d_nbits;
    const u32 *d_thresh;
    const u32 *d_dec_sym;
    const u32 *d_dec_xs;
    u32 lds_tables;  // 1: enc / dec tables fit the LDS budget and are staged per workgroup
};

// fast path: L = RANGE_FACTOR*M <= 8192 (tables live in LDS next to the stream rings)
struct TansFastDev {
    u32 K, L, nsb, size_bits;
    const uint4 *d_enc_sym;  // [256] {thresh, byte offset of the symbol's row in the encode table, nbits_base + 1, 0}
    const u16 *d_enc_tab;    // [L]   base_encode_step_table, flat (see scl_tans.hip)
    const u32 *d_dec_tab;    // [L]   state - L -> (x_shrunk << 8) | symbol
};

struct scl_rans_model;

struct scl_tans_model {
    int device;  // hipGetDevice() at create: the tables live there (scl_check_device)
    TansDev dev;
    TansFastDev fdev;
    u32 fast;
    // tANS is rANS with its per-step results cached (tANS.py:88-99, :208-215) and writes the same stream: when the
    // tables do not fit LDS the tuned rANS kernels serve the model without any table; with RANGE_FACTOR * M above
    // the 2^26-entry budget (the reference's default RANGE_FACTOR = 2^16 with M = 4096 asks for 2^28 entries) that
    // is the only route and no lookup tables are built (`tables` = 0)
    scl_rans_model *rans;
    u32 tables;  // the lookup tables CAN be built (RANGE_FACTOR * M <= 2^26) ...
    u32 built;   // ... and have been (eagerly without a companion rANS model, else on first use: tans_ensure_tables)
    std::mutex build_lock;
    u32 max_bits_per_symbol;
    u32 *d_freq, *d_cum, *d_enc, *d_nbits, *d_thresh, *d_dec_sym, *d_dec_xs;
    uint4 *d_fenc_sym;
    u16 *d_fenc_tab;
    u32 *d_fdec_tab;
};

int tans_fast_build_tables(scl_tans_model *m, const u32 *h_freq, const u32 *h_cum);
void tans_fast_encode_launch(const scl_tans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                             u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                             u32 *d_status, hipStream_t st);
void tans_fast_decode_launch(const scl_tans_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                             const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                             u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st);
