/*
 * scl_hip.h -- C ABI of the MI355X (gfx950) batched entropy-coding core.
 *
 * This is the drop-in boundary for ONE path of the Stanford Compression Library: the per-symbol
 * inner loops of its rANS / tANS / range / arithmetic coders.  Every "batch" entry point runs N
 * independent chunks, one wavefront lane per chunk, and each chunk's output is bit-identical to
 * one call of the reference method it replaces with a fresh coder object:
 *
 *   scl_rans_encode_batch   <->  rANSEncoder.encode_block        scl/compressors/rANS.py:186-210
 *   scl_rans_decode_batch   <->  rANSDecoder.decode_block        scl/compressors/rANS.py:270-297
 *   scl_tans_encode_batch   <->  tANSEncoder.encode_block        scl/compressors/tANS.py:159-193
 *   scl_tans_decode_batch   <->  tANSDecoder.decode_block        scl/compressors/tANS.py:252-279
 *   scl_range_encode_batch  <->  RangeEncoder.encode_block       scl/compressors/range_coder.py:188-207
 *   scl_range_decode_batch  <->  RangeDecoder.decode_block       scl/compressors/range_coder.py:269-317
 *   scl_aec_encode_batch    <->  ArithmeticEncoder.encode_block  scl/compressors/arithmetic_coding.py:80-161
 *   scl_aec_decode_batch    <->  ArithmeticDecoder.decode_block  scl/compressors/arithmetic_coding.py:203-287
 *   (model handles)         <->  rANSParams / tANSParams / RangeCoderParams+Frequencies /
 *                                AECParams+FreqModelBase subclasses (probability_models.py:15-160)
 *   scl_streams_compact     <->  the BitArray each encode_block returns (left-aligned bits) and,
 *                                with SCL_COMPACT_FRAMED, EncodedBlockWriter.write_block
 *                                (scl/core/encoded_stream.py:150-175)
 *
 * Conventions
 *   - plain C: pointers, sizes, opaque handles; no C++ or torch types.
 *   - every function returns an int status: SCL_OK or a negative SCL_E_* code;
 *     scl_last_error() gives a thread-local message for the last failure.
 *   - pointers named d_* are DEVICE pointers (HBM) owned by the caller; h_* are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous
 *     on that stream; nothing synchronises unless stated.
 *   - symbols are alphabet indices (position in Frequencies.freq_dict, prob_dist.py:193-205): uint8 for alphabets
 *     up to 256 entries -- the entry points BASELINE.json's configurations use, served by the tuned kernels -- and
 *     uint16 for alphabets up to 65536 through the *_u16 twins at the end of this header (any-parameter kernels).
 *   - a bit stream is MSB-first packed bytes (bitarray "big" endianness, bitarray_utils.py:25).
 *     A stream is described by (bit_offset, nbits): bit_offset is the absolute position of its
 *     first bit counted from the buffer's base pointer.  Encoders WRITE these descriptors;
 *     decoders READ them, so encoder output feeds the decoder without any repacking:
 *       * rANS / tANS streams are produced back to front (the reference prepends every field,
 *         rANS.py:196) and therefore end exactly at the end of their slot:
 *           bit_offset[c] = 8*(c+1)*out_stride - nbits[c]
 *         The slot bits in front of the stream are zero (this is the front padding of
 *         scl/core/encoded_stream.py:23-46).
 *       * range / arithmetic streams are produced front to back: bit_offset[c] = 8*c*out_stride.
 *     scl_streams_compact turns either into dense, byte-aligned, left-aligned streams.
 *   - per-chunk status words (d_status): 0 = ok, else a bit mask of SCL_ST_*.
 *   - all buffers holding streams must be 16-byte aligned, strides multiples of 16 bytes, and the
 *     input buffer of a decoder must stay readable for 16 bytes past the last stream byte.
 *     scl_*_slot_bytes returns multiples of 128: with 128-byte aligned base pointers (any hipMalloc /
 *     torch allocation) every lane then moves whole, aligned 128-byte lines, which is what the tuned
 *     kernels are built around (smaller alignments are accepted and served by the generic kernels or at
 *     reduced speed).
 */
#ifndef SCL_HIP_H
#define SCL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define SCL_OK 0
#define SCL_E_PARAM (-2)    /* parameter set rejected (also what the reference asserts on)      */
#define SCL_E_ALLOC (-7)    /* device / host allocation failed                                  */
#define SCL_E_HIP (-8)      /* a HIP runtime call failed (see scl_last_error)                    */
#define SCL_E_NODEVICE (-9) /* no gfx950 device visible                                          */
#define SCL_E_CHUNK (-10)   /* host convenience call: the chunk's status word was non-zero       */

/* per-chunk status bits */
#define SCL_ST_CAPACITY 0x1u  /* output slot too small / block larger than out_cap              */
#define SCL_ST_SYMBOL 0x2u    /* symbol index >= K              (KeyError, prob_dist.py:208)     */
#define SCL_ST_TRUNCATED 0x4u /* decoder needed bits past in_nbits                               */
#define SCL_ST_STATE 0x8u     /* final state != INITIAL_STATE   (assert, rANS.py:295)            */
#define SCL_ST_TOTAL 0x10u    /* total_freq >= MAX_ALLOWED_TOTAL_FREQ (assert, arithmetic_coding.py:110) */
#define SCL_ST_SIZE 0x20u     /* block size does not fit DATA_BLOCK_SIZE_BITS                    */

/* ---- library ------------------------------------------------------------------------------ */
const char *scl_last_error(void);
int scl_device_count(int *count);
int scl_abi_version(void);
/* Which kernels serve the batch calls of the CALLING THREAD (ABI version 5): on = 1 keeps the tuned kernels out (every call
   runs the any-parameter kernels -- how the tests compare the two implementations of every coder word for word),
   on = 0 lets the library choose, on = -1 (the initial state) follows the environment variable
   SCL_ANY_PARAMETER_KERNELS as before.  Thread-local: other threads' calls are not affected.  Returns the previous value. */
int scl_set_any_parameter_kernels(int on);

/* ---- rANS ---------------------------------------------------------------------------------- */
typedef struct scl_rans_model scl_rans_model;

typedef struct scl_rans_info {
    uint64_t M, L, H;        /* rANSParams.M / .L / .H              (rANS.py:100-104)            */
    uint32_t K;
    uint32_t num_state_bits; /* rANSParams.NUM_STATE_BITS           (rANS.py:119)                */
    uint32_t size_bits;      /* DATA_BLOCK_SIZE_BITS                                             */
    uint32_t num_bits_out;   /* NUM_BITS_OUT                                                     */
    uint32_t max_bits_per_symbol; /* worst-case field width, for slot sizing                     */
    uint32_t fast_path;      /* 1 if tuned kernels serve this model: u32 state with NUM_BITS_OUT = 1
                                and any total <= 4096, or NUM_BITS_OUT in {2,4,8,16} with a
                                power-of-two total <= 4096 (RANGE_FACTOR a power of two)        */
    int32_t device;          /* the HIP device the handle's tables live on (current at create);
                                batch calls refuse any other current device                     */
} scl_rans_info;

/* rANSParams(freqs, DATA_BLOCK_SIZE_BITS, NUM_BITS_OUT, RANGE_FACTOR) -> device-resident tables.
   Rejects: K == 0 or > 65536, any freq == 0, H >= 2^63 (quirks Q7/Q8), size_bits or num_bits_out
   outside 1..32.  Models of more than 256 symbols are coded through the *_u16 entry points. */
int scl_rans_model_create(const uint32_t *h_freq, uint32_t K, uint64_t range_factor,
                          uint32_t num_bits_out, uint32_t size_bits, scl_rans_model **out);
void scl_rans_model_destroy(scl_rans_model *m);
int scl_rans_model_info(const scl_rans_model *m, scl_rans_info *info);
/* (ABI version 5) which encoder a batch of n_chunks aligned, equally long rows would run with the calling thread's
   current settings: 'L' / 'S' = the headline kernels (NUM_BITS_OUT = 1; NUM_BITS_OUT in {4, 8, 16} within their bounds)
   with the 256-byte-ring / slot-ring writer (chosen by batch size; SCL_RANS_ENC_WRITER=L|S in the environment forces
   one, read at every call), 'B' = the NUM_BITS_OUT > 1 kernels that serve what those bounds exclude (NUM_BITS_OUT = 2,
   ...), 'G' = any-parameter kernels.  For tests and tools. */
int scl_rans_encoder_kind(const scl_rans_model *m, uint64_t n_chunks);
/* (ABI version 6) the encode and the decode kernel such a batch would run, as rocprofv3 prints them (template arguments
   included for the tuned kernels), NUL-terminated into enc / dec (either may be NULL) of `cap` >= 96 bytes: what lets a
   measurement name its kernels so that they can be matched against a kernel-trace summary.  No reference counterpart. */
int scl_rans_kernel_names(const scl_rans_model *m, uint64_t n_chunks, char *enc, char *dec, uint64_t cap);
/* bytes a slot must have so that any block of n symbols fits (multiple of 128: whole cache lines, so the
   64-byte store bursts of the fast kernels never straddle a sector) */
uint64_t scl_rans_slot_bytes(const scl_rans_model *m, uint64_t n_symbols);

/* chunk c reads symbols d_sym[c*sym_stride .. +len_c) with len_c = d_lens ? d_lens[c] : chunk_len */
int scl_rans_encode_batch(const scl_rans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                          const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                          uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                          uint32_t *d_out_nbits, uint32_t *d_status, void *stream);

/* chunk c decodes the stream (d_bit_offset[c], d_in_nbits[c]) of d_in into
   d_out_sym[c*out_stride .. ); at most out_cap symbols.  d_out_lens[c] = block size from the
   header, d_consumed[c] = num_bits_consumed (trailing bits are tolerated and not counted). */
int scl_rans_decode_batch(const scl_rans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                          const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                          uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                          uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                          uint32_t *d_status, void *stream);

/* ---- tANS (cached rANS: M power of two, NUM_BITS_OUT == 1; tANS.py:31-53) -------------------
 * Lookup tables have RANGE_FACTOR * M entries.  Up to 8192 entries they sit in LDS (tuned kernels); up to 2^26 they
 * are built in device memory; beyond that (the reference's inherited default RANGE_FACTOR = 2^16 with M = 4096 asks
 * for 2^28) no table is built and the model runs on the table-free rANS kernels, which write the same stream --
 * scl_tans_model_tables then fails and the batch entry points need 16-byte aligned rows and slots.
 * Since round 3 the table-free kernels are also the DEFAULT for LDS-sized tables wherever they apply (they are the faster
 * way to write the same stream on this machine); SCL_TANS_KERNELS=table in the environment keeps the lookup-table
 * kernels in charge.  scl_tans_model_tables exports the tables either way. */
typedef struct scl_tans_model scl_tans_model;

int scl_tans_model_create(const uint32_t *h_freq, uint32_t K, uint64_t range_factor,
                          uint32_t size_bits, scl_tans_model **out);
void scl_tans_model_destroy(scl_tans_model *m);
int scl_tans_model_info(const scl_tans_model *m, scl_rans_info *info);
uint64_t scl_tans_slot_bytes(const scl_tans_model *m, uint64_t n_symbols);
/* (ABI version 6) as scl_rans_kernel_names (a tANS model the table-free rANS kernels can serve runs on them) */
int scl_tans_kernel_names(const scl_tans_model *m, uint64_t n_chunks, char *enc, char *dec, uint64_t cap);
/* copy the device lookup tables back (for parity with tANS.py:285-337); any pointer may be NULL.
   h_enc / h_dec_sym / h_dec_xs have RANGE_FACTOR*M entries, h_nbits / h_thresh have K. */
int scl_tans_model_tables(const scl_tans_model *m, uint32_t *h_enc, uint32_t *h_nbits,
                          uint32_t *h_thresh, uint32_t *h_dec_sym, uint32_t *h_dec_xs);
int scl_tans_encode_batch(const scl_tans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                          const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                          uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                          uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_tans_decode_batch(const scl_tans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                          const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                          uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                          uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                          uint32_t *d_status, void *stream);

/* ---- range coder (RangeCoderParams + Frequencies; range_coder.py:55-86) --------------------- */
typedef struct scl_range_model scl_range_model;

/* Rejects: precision not a multiple of 8 in 16..32, any freq == 0, total_freq > BOTTOM
   (asserts at range_coder.py:64,84-85). */
int scl_range_model_create(const uint32_t *h_freq, uint32_t K, uint32_t precision,
                           uint32_t size_bits, scl_range_model **out);
void scl_range_model_destroy(scl_range_model *m);
uint64_t scl_range_slot_bytes(const scl_range_model *m, uint64_t n_symbols);
/* 1 if the tuned kernels serve this model (PRECISION = 32, DATA_BLOCK_SIZE_BITS = 32), 0 if the any-parameter ones do */
int scl_range_fast_path(const scl_range_model *m);
int scl_range_encode_batch(const scl_range_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                           const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                           uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                           uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_range_decode_batch(const scl_range_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                           const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                           uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                           uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                           uint32_t *d_status, void *stream);

/* ---- arithmetic coder (AECParams + frequency model; arithmetic_coding.py:20-56) ------------- */
typedef struct scl_aec_model scl_aec_model;

#define SCL_MODEL_FIXED 0  /* FixedFreqModel          probability_models.py:57-67               */
#define SCL_MODEL_IID 1    /* AdaptiveIIDFreqModel    probability_models.py:70-92               */
#define SCL_MODEL_ORDERK 2 /* AdaptiveOrderKFreqModel probability_models.py:95-160              */

/* scl_aec_*_batch: every chunk starts from a FRESH copy of the model (one chunk == one new encoder
   object); coder objects that live across blocks (quirk Q4) use the *_resume entry points below.
   h_freq_init: initial frequencies [K] for FIXED / IID (ignored for ORDERK, which starts from
   all-ones counts).  order_k: context length for ORDERK (0..3).  max_total: the model's
   max_allowed_total_freq.  precision 8..62: up to 32 the tuned kernels serve the models listed at
   scl_aec_fast_path; 33..62 run the any-parameter kernels with low / high in 128 bits (row totals must stay
   below 2^32: SCL_ST_TOTAL otherwise). */
int scl_aec_model_create(int model_kind, const uint32_t *h_freq_init, uint32_t K, uint32_t order_k,
                         uint64_t max_total, uint32_t precision, uint32_t size_bits,
                         scl_aec_model **out);
void scl_aec_model_destroy(scl_aec_model *m);
uint64_t scl_aec_slot_bytes(const scl_aec_model *m, uint64_t n_symbols);
/* bytes of device scratch the adaptive models need for n_chunks concurrent coders (0 for FIXED) */
uint64_t scl_aec_scratch_bytes(const scl_aec_model *m, uint64_t n_chunks);
/* 1 if batches whose chunks hold at most max_symbols symbols are served by the per-lane-LDS-table kernels
   (adaptive models, alphabet <= 16, <= 16 contexts; needs 16-byte aligned rows and slots of
   scl_aec_slot_bytes) -- the kernels BASELINE.json configs[3] names; 0 if the generic kernels serve them. */
int scl_aec_fast_path(const scl_aec_model *m, uint64_t max_symbols);
int scl_aec_encode_batch(const scl_aec_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                         const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                         uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                         uint32_t *d_out_nbits, uint32_t *d_status, void *d_scratch,
                         uint64_t scratch_bytes, void *stream);
int scl_aec_decode_batch(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                         const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                         uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                         uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                         uint32_t *d_status, void *d_scratch, uint64_t scratch_bytes,
                         void *stream);

/* ---- coder objects that live across blocks (quirk Q4) ----------------------------------------
 * The reference's ArithmeticEncoder / ArithmeticDecoder own their freq_model and never reset it
 * (arithmetic_coding.py:52-56,118), so DataEncoder.encode (core/data_encoder_decoder.py:57-69) codes
 * block i+1 with the counts and the order-k context block i left behind.  The *_resume entry points
 * reproduce that: chunk c of the call CONTINUES coder c of a caller-owned device state buffer
 * (scl_aec_state_bytes(m, n_coders) bytes, 256-byte aligned) and leaves the advanced state there.
 * The layout of the state buffer is a function of the n_coders it was reset with, so every call names that SAME
 * n_coders (state_bytes >= scl_aec_state_bytes(m, n_coders)); a batch may continue fewer coders than the state
 * holds (n_chunks <= n_coders: chunk c continues coder c), never more.
 * They run the any-parameter kernels.  FIXED models have nothing to carry and forward to the
 * plain entry points (d_state may be NULL).
 *
 * Canonical host form of one coder's state (upload / download / *_host_resume):
 *   h_counts : scl_aec_state_counts(m) actual counts -- IID: [K] = freqs_current.freq_list;
 *              ORDERK: [K^(k+1)] row-major, last axis = next symbol = freqs_kplus1_tuple.ravel()
 *              (probability_models.py:110);
 *   h_past_k : the last k symbol indices, oldest first = past_k (probability_models.py:116). */
uint64_t scl_aec_state_bytes(const scl_aec_model *m, uint64_t n_coders);
uint64_t scl_aec_state_counts(const scl_aec_model *m);
/* all n_coders coders = freshly constructed models */
int scl_aec_state_reset(const scl_aec_model *m, void *d_state, uint64_t state_bytes, uint64_t n_coders,
                        void *stream);
/* one coder's state from / to the canonical host form; both synchronise `stream` */
int scl_aec_state_upload(const scl_aec_model *m, void *d_state, uint64_t n_coders, uint64_t coder,
                         const uint32_t *h_counts, const uint32_t *h_past_k, void *stream);
int scl_aec_state_download(const scl_aec_model *m, const void *d_state, uint64_t n_coders, uint64_t coder,
                           uint32_t *h_counts, uint32_t *h_past_k, void *stream);
int scl_aec_encode_batch_resume(const scl_aec_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                uint32_t *d_out_nbits, uint32_t *d_status, void *d_state,
                                uint64_t state_bytes, uint64_t n_coders, void *stream);
int scl_aec_decode_batch_resume(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                                uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                uint32_t *d_status, void *d_state, uint64_t state_bytes, uint64_t n_coders,
                                void *stream);

/* ---- stream compaction / framing ------------------------------------------------------------ */
#define SCL_COMPACT_DENSE 0  /* stream c left-aligned at byte d_out_byte_offset[c], zero tail   */
#define SCL_COMPACT_FRAMED 1 /* EncodedBlockWriter framing per stream:
                                [u32 BE payload bytes][3-bit pad count][pad zeros][stream bits]
                                (encoded_stream.py:23-46,94-103,150-175)                         */

/* Gathers n_chunks streams (d_bit_offset, d_nbits) of d_in into one dense buffer.
   d_out_byte_offset has n_chunks+1 entries (exclusive prefix sum of the per-stream byte sizes;
   the last entry is the total).  d_scratch must hold scl_streams_compact_scratch_bytes(n_chunks).
   Fails per call (not per chunk): if the total exceeds out_capacity nothing past it is written
   and d_out_byte_offset[n_chunks] still reports the required size. */
uint64_t scl_streams_compact_scratch_bytes(uint64_t n_chunks);
int scl_streams_compact(const uint8_t *d_in, const uint64_t *d_bit_offset, const uint32_t *d_nbits,
                        uint64_t n_chunks, int mode, uint8_t *d_out, uint64_t out_capacity,
                        uint64_t *d_out_byte_offset, void *d_scratch, void *stream);
/* (ABI version 5) the same with the first record at byte *d_base of d_out, d_base in DEVICE memory (NULL = 0); the offsets
   written are absolute (entry 0 = *d_base, entry n = where the next record would start).  Sub-batch i + 1 of a batch passes
   the address of sub-batch i's last offset entry (it may be the address its own entry 0 goes to): one dense buffer and one
   offset table without the host ever learning a size -- so that a sub-batch can be compacted on a second stream while the
   next one is still being encoded. */
int scl_streams_compact_at(const uint8_t *d_in, const uint64_t *d_bit_offset, const uint32_t *d_nbits,
                           uint64_t n_chunks, int mode, uint8_t *d_out, uint64_t out_capacity,
                           uint64_t *d_out_byte_offset, const uint64_t *d_base, void *d_scratch, void *stream);

/* ---- wave-striped slots (ABI version 8) --------------------------------------------------------------------------------
 * A second MEMORY layout for the slots of a batch, written and read by the tuned rANS kernels (and by the tANS models
 * those kernels serve).  Nothing changes logically: stream c is still nbits[c] bits at bit offset bit_offset[c] =
 * 8*(c+1)*out_stride - nbits[c] of a buffer of slots of out_stride bytes -- rANSEncoder.encode_block's bits
 * (scl/compressors/rANS.py:186-210), in the positions the plain entry points put them -- but the 64 slots of a wavefront
 * are interleaved in 16-byte pieces:
 *     logical byte A = c*out_stride + b   lives at   (c/64)*64*out_stride + (b/16)*1024 + 16*(c%64) + b%16
 * so that piece q of the 64 lanes of a wave is one contiguous kilobyte.  A lane then stores 16 bytes at a time where the
 * linear layout made it buffer whole 128-byte lines (256 B of LDS per lane: two waves per SIMD); the striped encoder runs
 * four waves per SIMD and stores row by row, 64 adjacent pieces per instruction.  d_out must hold
 * round_up(n_chunks, 64) * out_stride bytes.  Worth it for batches that fill the chip (>= ~196 608 chunks on MI355X:
 * 1 GiB headline batch encode 0.55 -> 0.52 ms); smaller batches are faster on the linear entry points.
 *   scl_*_striped_ok              1 if the striped entry points serve this model (= the tuned kernels do: fast_path with
 *                                 NUM_BITS_OUT = 1 or in {4, 8, 16}, alphabet <= 256), else 0 -- they then fail with
 *                                 SCL_E_PARAM, as they do while the calling thread keeps the tuned kernels out;
 *   scl_*_encode_batch_striped    arguments of scl_rans_encode_batch; out_stride >= scl_*_slot_bytes(chunk_len), < 2^24;
 *   scl_*_decode_batch_striped    arguments of scl_rans_decode_batch with in_stride (the encoder's out_stride) in place of
 *                                 in_size_bytes; stream c must lie inside logical slot c;
 *   scl_streams_compact_striped   scl_streams_compact_at on striped slots: the SAME dense / framed bytes as the linear
 *                                 path produces (BitArray.tobytes() of every block back to back, or the reference's file
 *                                 format, encoded_stream.py:150-175) -- the only form the reference ever sees;
 *   scl_*_kernel_names_striped    scl_*_kernel_names for the striped kernels. */
int scl_rans_striped_ok(const scl_rans_model *m);
int scl_tans_striped_ok(const scl_tans_model *m);
/* the range coder (RangeEncoder.encode_block / RangeDecoder.decode_block, range_coder.py:188-207, :269-317) on the same
   layout: streams grow front to back, bit_offset[c] = 8*c*out_stride as on linear slots; served: PRECISION = 32 models over
   uniform bytes (configs[2]) and over tables with totals 256..4096 (what the cooperative line store of the linear kernels
   cannot help: lanes that are not in lockstep) */
int scl_range_striped_ok(const scl_range_model *m);
int scl_range_encode_batch_striped(const scl_range_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                   const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                   uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                   uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_range_decode_batch_striped(const scl_range_model *m, const uint8_t *d_in, uint64_t in_stride,
                                   const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                   uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                                   uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                   uint32_t *d_status, void *stream);
int scl_rans_kernel_names_striped(const scl_rans_model *m, uint64_t n_chunks, char *enc, char *dec, uint64_t cap);
int scl_tans_kernel_names_striped(const scl_tans_model *m, uint64_t n_chunks, char *enc, char *dec, uint64_t cap);
int scl_rans_encode_batch_striped(const scl_rans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                  const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                  uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                  uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_rans_decode_batch_striped(const scl_rans_model *m, const uint8_t *d_in, uint64_t in_stride,
                                  const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                  uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                                  uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                  uint32_t *d_status, void *stream);
int scl_tans_encode_batch_striped(const scl_tans_model *m, const uint8_t *d_sym, uint64_t sym_stride,
                                  const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                  uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                  uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_tans_decode_batch_striped(const scl_tans_model *m, const uint8_t *d_in, uint64_t in_stride,
                                  const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                  uint64_t n_chunks, uint8_t *d_out_sym, uint64_t out_stride,
                                  uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                  uint32_t *d_status, void *stream);
int scl_streams_compact_striped(const uint8_t *d_in, uint64_t in_stride, const uint64_t *d_bit_offset,
                                const uint32_t *d_nbits, uint64_t n_chunks, int mode, uint8_t *d_out,
                                uint64_t out_capacity, uint64_t *d_out_byte_offset, const uint64_t *d_base,
                                void *d_scratch, void *stream);

/* (ABI version 7) Host-side index of a framed block file held in host memory -- the walk EncodedBlockReader.get_block makes
   one record at a time (encoded_stream.py:196-225) followed by Padder.remove_byte_padding (:48-58), for a whole buffer:
   for every COMPLETE record [u32 BE payload bytes][payload] starting at h_buf[0] writes where its stream starts (bit offset
   from h_buf), its number of stream bits and the value of its first size_bits bits (the block's DATA_BLOCK_SIZE_BITS
   header; size_bits <= 64).  Stops in front of the first record that crosses buf_size or after max_records; *n_records
   = records indexed, *consumed = bytes they span (what is left belongs to the next buffer -- or is a truncated file).
   No device, no stream.  SCL_E_PARAM (with *n_records / *consumed = what came before) for a record with an empty payload
   or one shorter than its padding and size header. */
int scl_framed_index_host(const uint8_t *h_buf, uint64_t buf_size, uint32_t size_bits, uint64_t max_records,
                          uint64_t *h_bit_offset, uint64_t *h_nbits, uint64_t *h_block_size,
                          uint64_t *n_records, uint64_t *consumed);

/* ---- multi-GPU: variable-length gather of compacted streams over RCCL (SURVEY.md 8e, configs[4]) ------
 * No reference counterpart (the reference has no communication).  One process per GPU; every rank encodes and
 * compacts its own block-contiguous shard, then the dense payloads go to one rank: an all-gather of the byte
 * counts, then one grouped ncclSend / ncclRecv per sender straight into the root's buffer at the prefix offsets
 * (xGMI is point to point: the root receives on all its links at once).  RCCL is dlopen'ed on first use.
 *   scl_rccl_unique_id      : rank 0 creates the 128-byte id and hands it to the others out of band (any channel);
 *   scl_rccl_comm_create    : collective over the `world` ranks; the communicator belongs to the current device;
 *   scl_rccl_allgather_u64  : collective; one u64 per rank -> h_out[world] on every rank (synchronises `stream`):
 *                             the byte counts, so that the root can size its buffer before anything is posted;
 *   scl_rccl_allgather_async: collective, asynchronous on `stream`, device to device: n_u64 values per rank ->
 *                             d_out[world * n_u64] on every rank; nothing waits for the host (the overlapped
 *                             pipeline queues it behind a sub-batch's compaction and reads it back with an event);
 *   scl_streams_gather_rccl : collective, asynchronous on `stream`: rank r's send_bytes bytes arrive at the root's
 *                             d_recv + h_rank_offsets[r]; h_rank_offsets[world + 1] (host) = exclusive prefix sum
 *                             of the counts, last entry = total.  d_recv matters on the root only;
 *   scl_streams_gatherv_rccl: n_parts such gathers (a payload and its per-chunk offset table, say) in ONE grouped
 *                             exchange: part p moves h_send_bytes[p] bytes from d_send[p] to the root's d_recv[p]
 *                             + h_rank_offsets[p * (world + 1) + r].
 *   scl_streams_gather_blocks_rccl: configs[4]'s exchange as one call: a (sub-)batch's dense payload and its
 *                             n_chunks + 1 record offsets go to the root in one grouped exchange, and the root shifts
 *                             every rank's offsets by the bytes of the ranks before it (one small kernel on `stream`),
 *                             so d_recv_offsets [sum chunks + 1] is the offset table a single process would have
 *                             produced; h_bytes_by_rank / h_chunks_by_rank [world] are the exchanged counts.
 *   Errors: all arguments are checked before anything is posted; after ncclGroupStart the group is closed on every
 *   path (first error recorded, ncclGroupEnd, then return).  A layout that contradicts this rank's own count is
 *   refused on this rank only -- the peers then wait in the exchange, so derive layouts from exchanged counts. */
/*   scl_rccl_comm_info      : (ABI version 5) what the communicator reports about itself -- ncclCommUserRank /
 *                             ncclCommCount -- and the device it belongs to;
 *   scl_rccl_inject_api     : (ABI version 5) TEST HOOK.  Every RCCL operation above goes through one table of eleven
 *                             plain-C function pointers (scl_rccl_api); by default it is filled from librccl.so on first
 *                             use, this call replaces it (NULL restores the default).  With host_memory != 0 the
 *                             buffers handed to the calls above are host memory and the library makes no HIP call on
 *                             these paths (its own copies are memcpy, the root's offset fix-up a loop): the tests run a
 *                             W-rank exchange -- layout checks, grouped posts, the root's receive offsets -- inside one
 *                             process on a machine without a GPU.  Not for production use. */
typedef struct scl_comm scl_comm;
typedef struct scl_rccl_api {
    int (*get_unique_id)(uint8_t *id128);                                       /* ncclGetUniqueId    */
    int (*comm_init_rank)(void **comm, int nranks, const uint8_t *id128, int rank); /* ncclCommInitRank */
    int (*comm_destroy)(void *comm);                                            /* ncclCommDestroy    */
    int (*comm_count)(void *comm, int *nranks);                                 /* ncclCommCount      */
    int (*comm_user_rank)(void *comm, int *rank);                               /* ncclCommUserRank   */
    int (*all_gather)(const void *send, void *recv, uint64_t count, int dtype, void *comm, void *stream);
    int (*send)(const void *buf, uint64_t count, int dtype, int peer, void *comm, void *stream);
    int (*recv)(void *buf, uint64_t count, int dtype, int peer, void *comm, void *stream);
    int (*group_start)(void);
    int (*group_end)(void);
    const char *(*error_string)(int result);
    int host_memory; /* 1: buffers are host memory, the library makes no HIP call around the exchange */
} scl_rccl_api;      /* dtype: 1 = uint8, 5 = uint64 (ncclDataType_t); results: 0 = success */
int scl_rccl_inject_api(const scl_rccl_api *api);
int scl_rccl_comm_info(scl_comm *c, int *rank, int *nranks, int *device);
int scl_rccl_unique_id(uint8_t *id128);
int scl_rccl_comm_create(const uint8_t *id128, int rank, int world, scl_comm **out);
void scl_rccl_comm_destroy(scl_comm *c);
int scl_rccl_allgather_u64(scl_comm *c, uint64_t value, uint64_t *h_out, void *stream);
int scl_rccl_allgather_async(scl_comm *c, const uint64_t *d_in, uint64_t *d_out, uint64_t n_u64, void *stream);
int scl_streams_gather_rccl(scl_comm *c, int root, const uint8_t *d_send, uint64_t send_bytes, uint8_t *d_recv,
                            const uint64_t *h_rank_offsets, void *stream);
int scl_streams_gatherv_rccl(scl_comm *c, int root, uint32_t n_parts, const uint8_t *const *d_send,
                             const uint64_t *h_send_bytes, uint8_t *const *d_recv, const uint64_t *h_rank_offsets,
                             void *stream);
int scl_streams_gather_blocks_rccl(scl_comm *c, int root, const uint8_t *d_payload, uint64_t payload_bytes,
                                   const uint64_t *d_offsets, uint64_t n_chunks, uint8_t *d_recv_payload,
                                   uint64_t *d_recv_offsets, const uint64_t *h_bytes_by_rank,
                                   const uint64_t *h_chunks_by_rank, void *stream);

/* ---- model construction helper (row f3): symbol histogram ------------------------------------------ */
/* d_counts[256] (uint64) += number of occurrences of every byte value in d_sym[0..n).  The caller zeroes
   d_counts.  Equals DataBlock.get_counts() (scl/core/data_block.py:37-62) for uint8 data. */
int scl_histogram_u8(const uint8_t *d_sym, uint64_t n, uint64_t *d_counts, void *stream);
/* the same for uint16 symbol indices of an alphabet of K <= 65536 symbols: d_counts[K] += occurrences; indices >= K are
   not counted but tallied in *d_out_of_range (uint32, zeroed by the caller).  ABI 4. */
int scl_histogram_u16(const uint16_t *d_sym, uint64_t n, uint32_t K, uint64_t *d_counts,
                      uint32_t *d_out_of_range, void *stream);

/* ---- host convenience (one chunk, host buffers; allocates, copies, runs N=1, synchronises) --- */
/* These back the drop-in encode_block / decode_block of the Python classes.  h_out receives the
   left-aligned stream (what BitArray.tobytes() would give); *nbits its length. */
int scl_rans_encode_host(const scl_rans_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                         uint64_t out_cap_bytes, uint64_t *nbits);
int scl_rans_decode_host(const scl_rans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                         uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
int scl_tans_encode_host(const scl_tans_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                         uint64_t out_cap_bytes, uint64_t *nbits);
int scl_tans_decode_host(const scl_tans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                         uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
int scl_range_encode_host(const scl_range_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                          uint64_t out_cap_bytes, uint64_t *nbits);
int scl_range_decode_host(const scl_range_model *m, const uint8_t *h_in, uint64_t in_nbits,
                          uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
int scl_aec_encode_host(const scl_aec_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                        uint64_t out_cap_bytes, uint64_t *nbits);
int scl_aec_decode_host(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits,
                        uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
/* one block of a coder whose state the caller keeps on the host between calls (h_counts / h_past_k are
   read before and rewritten after the block): ArithmeticEncoder.encode_block / ArithmeticDecoder.decode_block
   on an object that has coded blocks before (arithmetic_coding.py:52-56) */
int scl_aec_encode_host_resume(const scl_aec_model *m, const uint8_t *h_sym, uint64_t n, uint8_t *h_out,
                               uint64_t out_cap_bytes, uint64_t *nbits, uint32_t *h_counts,
                               uint32_t *h_past_k);
int scl_aec_decode_host_resume(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits,
                               uint8_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed,
                               uint32_t *h_counts, uint32_t *h_past_k);
/* peek the DATA_BLOCK_SIZE_BITS header of a host stream (so callers can size h_out_sym) */
int scl_stream_block_size_host(const uint8_t *h_in, uint64_t in_nbits, uint32_t size_bits,
                               uint64_t *n_out);

/* ---- alphabets of up to 65536 symbols: uint16 symbol indices (ABI 4) ---------------------------------
 * The reference codes any hashable alphabet (Frequencies.freq_dict, prob_dist.py:193-205); its LZ77 / text models
 * reach past 256 symbols.  Every batch / host entry point above has a *_u16 twin with the same arguments, the
 * symbol arrays typed uint16_t and the symbol strides counted in SYMBOLS.  They take ANY model handle (models of
 * more than 256 symbols are refused by the uint8 entry points with SCL_E_PARAM) and run the any-parameter kernels:
 * tables are read where they are in device memory, adaptive order-k / i.i.d. rows are scanned linearly (cost per
 * symbol grows with the alphabet).  For a model of at most 256 symbols the stream equals the uint8 entry point's. */
int scl_rans_encode_batch_u16(const scl_rans_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                              const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                              uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                              uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_rans_decode_batch_u16(const scl_rans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                              const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                              uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                              uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                              uint32_t *d_status, void *stream);
int scl_tans_encode_batch_u16(const scl_tans_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                              const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                              uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                              uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_tans_decode_batch_u16(const scl_tans_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                              const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                              uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                              uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                              uint32_t *d_status, void *stream);
int scl_range_encode_batch_u16(const scl_range_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                               const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                               uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                               uint32_t *d_out_nbits, uint32_t *d_status, void *stream);
int scl_range_decode_batch_u16(const scl_range_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                               const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                               uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                               uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                               uint32_t *d_status, void *stream);
int scl_aec_encode_batch_u16(const scl_aec_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                             const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                             uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                             uint32_t *d_out_nbits, uint32_t *d_status, void *d_scratch,
                             uint64_t scratch_bytes, void *stream);
int scl_aec_decode_batch_u16(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                             const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                             uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                             uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                             uint32_t *d_status, void *d_scratch, uint64_t scratch_bytes,
                             void *stream);
int scl_aec_encode_batch_resume_u16(const scl_aec_model *m, const uint16_t *d_sym, uint64_t sym_stride,
                                    const uint32_t *d_lens, uint32_t chunk_len, uint64_t n_chunks,
                                    uint8_t *d_out, uint64_t out_stride, uint64_t *d_out_bit_offset,
                                    uint32_t *d_out_nbits, uint32_t *d_status, void *d_state,
                                    uint64_t state_bytes, uint64_t n_coders, void *stream);
int scl_aec_decode_batch_resume_u16(const scl_aec_model *m, const uint8_t *d_in, uint64_t in_size_bytes,
                                    const uint64_t *d_bit_offset, const uint32_t *d_in_nbits,
                                    uint64_t n_chunks, uint16_t *d_out_sym, uint64_t out_stride,
                                    uint32_t out_cap, uint32_t *d_out_lens, uint32_t *d_consumed,
                                    uint32_t *d_status, void *d_state, uint64_t state_bytes, uint64_t n_coders,
                                    void *stream);
int scl_rans_encode_host_u16(const scl_rans_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                             uint64_t out_cap_bytes, uint64_t *nbits);
int scl_rans_decode_host_u16(const scl_rans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                             uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
int scl_tans_encode_host_u16(const scl_tans_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                             uint64_t out_cap_bytes, uint64_t *nbits);
int scl_tans_decode_host_u16(const scl_tans_model *m, const uint8_t *h_in, uint64_t in_nbits,
                             uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
int scl_range_encode_host_u16(const scl_range_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                              uint64_t out_cap_bytes, uint64_t *nbits);
int scl_range_decode_host_u16(const scl_range_model *m, const uint8_t *h_in, uint64_t in_nbits,
                              uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
int scl_aec_encode_host_u16(const scl_aec_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                            uint64_t out_cap_bytes, uint64_t *nbits);
int scl_aec_decode_host_u16(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits,
                            uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed);
int scl_aec_encode_host_resume_u16(const scl_aec_model *m, const uint16_t *h_sym, uint64_t n, uint8_t *h_out,
                                   uint64_t out_cap_bytes, uint64_t *nbits, uint32_t *h_counts,
                                   uint32_t *h_past_k);
int scl_aec_decode_host_resume_u16(const scl_aec_model *m, const uint8_t *h_in, uint64_t in_nbits,
                                   uint16_t *h_out_sym, uint64_t out_cap, uint64_t *n_out, uint64_t *consumed,
                                   uint32_t *h_counts, uint32_t *h_past_k);

#ifdef __cplusplus
}
#endif
#endif /* SCL_HIP_H */
