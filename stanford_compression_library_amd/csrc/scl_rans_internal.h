// scl_rans_internal.h -- model layout shared by scl_rans.hip (generic kernels, host API) and
// scl_rans_fast.hip (the gfx950 fast path).  Internal to csrc/.
#pragma once
#include "scl_common.h"

struct RansDev {
    u32 K;
    u32 b;
    u32 size_bits;
    u32 nsb;
    u64 M, RF, L;
    u32 m_log2;  // log2(M) if M is a power of two, else 0xFFFFFFFF
    const u32 *d_freq;
    const u32 *d_cum;
};

// fast path: H < 2^31, 2 <= M <= 4096, NUM_BITS_OUT = 1, RANGE_FACTOR = 2^r
struct RansFastDev {
    u32 K;
    u32 nsb;        // NUM_STATE_BITS = r + m + 1 <= 30
    u32 size_bits;
    u32 m_log2;     // log2(M) if M is a power of two, else 0xFFFFFFFF
    u32 L;
    u32 M;
    u32 enc_msh;    // encoder quotient shift MSH | pre-shift << 8 | r << 16 (rans_fast_build_tables);
                    // NUM_BITS_OUT = b > 1: (r + b) << 16 | b << 24 (rf_encode_entry_b)
    u32 enc_folded; // 1: the pre-shift of a small table is folded into the reciprocals (rcp << pre), the quotient shift is 0:
                    // the MSH_T = -1 flavour of the encoder (rf_encode_entry)
    u32 b;          // NUM_BITS_OUT: 1, or 4 / 8 / 16 on the same kernels since round 4
    u32 dec_sadd;   // b > 1: b - 1 - cbl, cbl = 32 - bit_width(L)   (rf_decode_symbol)
    u32 dec_notb;   // b > 1: ~(b - 1)
    const uint4 *d_enc_tab;  // [256] {rcp, M-f, cum, k_lo}
    const uint2 *d_dec_tab;  // [M]   slot -> {f | sym << 24, slot - cum}
};

// fast path for NUM_BITS_OUT = b in {2, 4, 8, 16} (scl_rans_fast_b.hip): M a power of two <= 4096, H < 2^31
struct RansFastBDev {
    u32 K;
    u32 nsb;
    u32 size_bits;
    u32 m_log2;
    u32 L;
    u32 M;
    u32 b;
    u32 cbl;             // 32 - bit_width(L): leading zeros of a state that needs no refill
    const uint4 *d_enc;  // [256] {1/f as binary64 (lo, hi), thresh, cum}
    const u32 *d_aux;    // [256] (M - f) | (b k1) << 24
    const uint2 *d_dec;  // [M]   slot -> {f | sym << 24, slot - cum}
};

struct scl_rans_model {
    int device;  // hipGetDevice() at create: the tables live there (scl_check_device)
    RansDev dev;
    RansFastDev fdev;
    RansFastBDev fbdev;
    u32 fastb;
    uint4 *d_encb_tab;
    u32 *d_encb_aux;
    uint2 *d_decb_tab;
    u64 H;
    u32 max_bits_per_symbol;
    u32 state32;  // H < 2^32
    u32 fast;
    u32 enc_lockstep;  // most symbols always release the same number of bits: lanes run in lockstep (writer choice)
    u32 *d_freq;
    u32 *d_cum;
    uint4 *d_enc_tab;
    uint2 *d_dec_tab;
};

// scl_rans_fast.hip
int rans_fast_build_tables(scl_rans_model *m, const u32 *h_freq, const u32 *h_cum);
bool rf_use_slot_writer(const scl_rans_model *m, u64 n_chunks);
// striped: wave-striped slots (AnsBackWriterT / AnsBitReaderT, scl_ans_fast_io.h) -- d_out holds round_up(n_chunks, 64)
// slots; the decoder's `in_size_bytes` is then the slot stride
void rans_fast_kernel_names(const scl_rans_model *m, u64 n_chunks, char *enc, char *dec, size_t cap, bool striped = false);
void rans_fast_encode_launch(const scl_rans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                             u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                             u32 *d_status, hipStream_t st, bool striped = false);
void rans_fast_decode_launch(const scl_rans_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                             const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                             u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st, bool striped = false);

// scl_rans.hip: the striped entry points' bodies (shared with the tANS models the table-free rANS kernels serve)
int rans_striped_encode(const char *what, const scl_rans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                        u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                        u32 *d_status, hipStream_t st);
int rans_striped_decode(const char *what, const scl_rans_model *m, const u8 *d_in, u64 in_stride, const u64 *d_bit_off,
                        const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap, u32 *d_out_lens,
                        u32 *d_consumed, u32 *d_status, hipStream_t st);

// scl_rans_fast_b.hip (NUM_BITS_OUT > 1)
int rans_fastb_build_tables(scl_rans_model *m, const u32 *h_freq, const u32 *h_cum);
void rans_fastb_encode_launch(const scl_rans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                              u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                              u32 *d_status, hipStream_t st);
void rans_fastb_decode_launch(const scl_rans_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st);
