// scl_range_internal.h -- model layout shared by scl_range.hip (generic kernels, host API) and
// scl_range_fast.hip (the gfx950 fast path).  Internal to csrc/.
#pragma once
#include "scl_common.h"

struct RangeDev {
    u32 K;
    u32 P;          // PRECISION
    u32 size_bits;
    u32 M;
    u32 m_log2;     // log2(M) if power of two else 0xFFFFFFFF
    const u32 *d_freq;
    const u32 *d_cum;
    const u8 *d_slot2sym;  // [M] slot -> symbol (decode LUT), null when M is too large
};

// fast path: PRECISION = 32, DATA_BLOCK_SIZE_BITS = 32, any total M <= 4096
struct RangeFastDev {
    u32 K;
    u32 m_log2;  // log2(M) if power of two else 0xFFFFFFFF
    u32 M;
    u32 uni_t;   // 256 symbols of one power-of-two frequency 2^uni_t (total a power of two): the table-free kernels;
                 // 0xFFFFFFFF otherwise
    const uint2 *d_enc_tab;  // [256] {cum, freq}
    const u8 *d_slot2sym;    // [M]
};

struct scl_range_model {
    int device;  // hipGetDevice() at create: the tables live there (scl_check_device)
    RangeDev dev;
    RangeFastDev fdev;
    u32 fast;
    u32 *d_freq, *d_cum;
    u8 *d_slot2sym;
    uint2 *d_enc_tab;
};

int range_fast_build_tables(scl_range_model *m, const u32 *h_freq, const u32 *h_cum);
// striped: wave-striped slots (RgOutT / AnsBitReaderT; range_fast_striped_ok must hold) -- d_out holds round_up(n_chunks,
// 64) slots; the decoder's `in_size_bytes` is then the slot stride
bool range_fast_striped_ok(const scl_range_model *m);
void range_fast_encode_launch(const scl_range_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                              u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                              u32 *d_status, hipStream_t st, bool striped = false);
void range_fast_decode_launch(const scl_range_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st, bool striped = false);
