// scl_aec_internal.h -- model handle of the arithmetic coder, shared by scl_aec.hip (any parameters) and
// scl_aec_fast.hip (small-alphabet adaptive models with per-lane context tables in LDS).  Internal to csrc/.
#pragma once
#include "scl_common.h"

struct AecDev {
    int kind;
    u32 K, k;
    u32 P, size_bits;
    u64 max_total;
    u64 cells;  // per-chunk scratch cells (u32)
    u64 ctx_mod;  // K^k
    const u32 *d_freq;  // [K] initial frequencies (FIXED / IID)
    const u32 *d_cum;   // [K] exclusive cumulative of d_freq (FIXED)
    u32 total0;         // sum of initial frequencies
    u32 fenwick;        // ORDERK with a large alphabet: two-level rows of (count - 1), see scl_aec.hip
    u32 row_cells;      // two-level rows: 16 block totals + 16 * ceil(K / 16) counts
};

struct scl_aec_model {
    int device;  // hipGetDevice() at create: the tables live there (scl_check_device)
    AecDev dev;
    u32 *d_freq, *d_cum;
    u32 h_freq[256];  // host copy of the initial frequencies (all ones for ORDERK)
    u32 *d_iid_init;  // IID, alphabet > 16: the two-level cumulative table of scl_aec_iid.hip (17 rows x 8 u32)
};

// scl_aec_fast.hip
bool aec_fast_ok(const scl_aec_model *m, u64 max_symbols);
void aec_fast_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                            u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                            u32 *d_status, hipStream_t st);
void aec_fast_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                            const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                            u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st);
// scl_aec_split.hip: the encoder of the same models with the model side and the coder side of a chunk in two waves
int aec_split_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                            u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                            u32 *d_status, hipStream_t st);
// scl_aec_wide.hip: order-k models in the two-level row layout (large alphabets), tuned arithmetic over device-memory rows
bool aec_wide_ok(const scl_aec_model *m, u64 max_symbols);
u64 aec_wide_scratch_bytes(const scl_aec_model *m, u64 n_chunks);
void aec_wide_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                            u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                            u32 *d_status, u32 *d_scratch, hipStream_t st);
void aec_wide_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                            const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                            u32 *d_out_lens, u32 *d_consumed, u32 *d_status, u32 *d_scratch, hipStream_t st);
// scl_aec_sparse.hip: the same models with one 64-byte line per context (the symbols seen in it) until it has been seen 28
// times, then its dense row: one table piece read and written per symbol instead of two
u64 aec_sparse_zero_bytes(const scl_aec_model *m, u64 n_chunks);
void aec_sparse_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                              u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                              u32 *d_status, u32 *d_scratch, hipStream_t st);
void aec_sparse_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, u32 *d_scratch, hipStream_t st);
// scl_aec_static.hip
bool aec_static_ok(const scl_aec_model *m);
void aec_static_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                              u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset,
                              u32 *d_out_nbits, u32 *d_status, hipStream_t st);
void aec_static_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st);
// scl_aec_iid.hip
bool aec_iid_ok(const scl_aec_model *m, u64 max_symbols);
void aec_iid_build_init(const u32 *h_freq, u32 K, u32 *out136);
void aec_iid_encode_launch(const scl_aec_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens, u32 chunk_len,
                           u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_out_bit_offset, u32 *d_out_nbits,
                           u32 *d_status, hipStream_t st);
void aec_iid_decode_launch(const scl_aec_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_offset,
                           const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                           u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st);
