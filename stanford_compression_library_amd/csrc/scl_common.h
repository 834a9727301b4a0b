// scl_common.h -- shared host/device helpers of the gfx950 entropy-coding library.
// Internal to csrc/; the public contract is include/scl_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/scl_hip.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

#define SCL_WAVE 64
#define SCL_ABI_VERSION 8
#define SCL_MAX_ALPHABET 65536u  // uint16 symbol indices (the *_u16 entry points); the uint8 entry points stop at 256

// ---- host-side error plumbing ------------------------------------------------------------------
void scl_set_error(const char *fmt, ...);

#define SCL_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            scl_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                       \
            return SCL_E_HIP;                                                              \
        }                                                                                  \
    } while (0)

#define SCL_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            scl_set_error(__VA_ARGS__); \
            return SCL_E_PARAM;       \
        }                             \
    } while (0)

static inline u32 scl_bit_width_u64(u64 x) {  // get_bit_width, bitarray_utils.py:8-20
    u32 w = 0;
    if (x == 0) return 1;
    while (x) {
        ++w;
        x >>= 1;
    }
    return w;
}

static inline u64 scl_round_up(u64 x, u64 a) { return (x + a - 1) / a * a; }

// One model handle = one device: its tables live in the HBM of the device that was current at *_model_create.
// The batch entry points launch on the CURRENT device, so they refuse a handle made on another one (SCL_E_PARAM)
// instead of dereferencing a foreign device's pointers.  (scl_core.hip)
int scl_current_device(void);
int scl_check_device(int model_device, const char *what);

// Rows the tuned kernels cannot take as they are -- symbol rows that do not start on 16-byte boundaries, decoded rows
// whose stride is not a multiple of 16 -- are re-laid through stream-ordered scratch INSIDE the library (round 3; the
// Python wrapper used to copy them), so every caller of the C ABI reaches the tuned kernels.  (scl_core.hip)
// SCL_ANY_PARAMETER_KERNELS=1 in the environment keeps the tuned kernels out altogether (tests, stress tools: it is
// how the two implementations of every coder are compared word for word).
bool scl_force_generic(void);
struct RowRelay {
    u8 *scratch = nullptr;
    hipStream_t st = nullptr;
    u8 *user_out = nullptr;  // decode side: where the rows go back to
    u64 user_stride = 0, stride = 0, n_rows = 0;
    u32 row_bytes = 0, sym_bytes = 1;
    bool failed = false;  // the scratch could not be allocated: rows stay as they are (scl_last_error has the reason)
    // encode side: d_sym / sym_stride are replaced by an aligned copy when they are not aligned
    int in(const u8 *&d_sym, u64 &sym_stride, u32 chunk_len, u64 n_chunks, hipStream_t stream);
    // decode side: d_out / out_stride are replaced by aligned scratch; out_end() copies the rows back
    int out_begin(u8 *&d_out, u64 &out_stride, u32 out_cap, u64 n_chunks, hipStream_t stream);
    int out_end(const u32 *d_out_lens);
    ~RowRelay();
};
static inline bool scl_rows_aligned(const void *p, u64 stride) { return (((uintptr_t)p | stride) & 15) == 0; }

// RAII-less scratch helper for the *_host convenience calls
struct ScratchDev {
    void *p = nullptr;
    int alloc(u64 bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) {
            scl_set_error("hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
            p = nullptr;
            return SCL_E_ALLOC;
        }
        return SCL_OK;
    }
    ~ScratchDev() {
        if (p) (void)hipFree(p);
    }
};

// generic single-chunk host driver shared by the four coders (scl_core.hip)
struct HostEncodeCall {
    // launches the batch encoder for n_chunks = 1 on device buffers
    int (*run)(const void *model, const u8 *d_sym, u32 n, u8 *d_out, u64 out_stride, u64 *d_bit_off,
               u32 *d_nbits, u32 *d_status, void *d_scratch, u64 scratch_bytes);
    u64 (*slot_bytes)(const void *model, u64 n);
    u64 (*scratch_bytes)(const void *model);
    // optional hooks around the launch (coder state carried across blocks): pre() fills d_scratch before the
    // launch, post() reads it back after the device has finished; `user` is handed to both
    int (*pre)(const void *model, void *d_scratch, void *user) = nullptr;
    int (*post)(const void *model, const void *d_scratch, void *user) = nullptr;
    void *user = nullptr;
    u32 sym_bytes = 1;  // 2: h_sym / d_sym hold uint16 indices (run() forwards to the *_u16 batch entry point)
};
struct HostDecodeCall {
    int (*run)(const void *model, const u8 *d_in, u64 in_bytes, const u64 *d_bit_off, const u32 *d_in_nbits,
               u8 *d_out_sym, u32 out_cap, u32 *d_out_len, u32 *d_consumed, u32 *d_status, void *d_scratch,
               u64 scratch_bytes);
    u64 (*scratch_bytes)(const void *model);
    int (*pre)(const void *model, void *d_scratch, void *user) = nullptr;
    int (*post)(const void *model, const void *d_scratch, void *user) = nullptr;
    void *user = nullptr;
    u32 sym_bytes = 1;
};
int scl_host_encode_one(const HostEncodeCall &call, const void *model, const u8 *h_sym, u64 n, u8 *h_out,
                        u64 out_cap_bytes, u64 *nbits);
int scl_host_decode_one(const HostDecodeCall &call, const void *model, const u8 *h_in, u64 in_nbits,
                        u8 *h_out_sym, u64 out_cap, u64 *n_out, u64 *consumed);

// ---- device-side bit I/O (MSB-first streams) ------------------------------------------------------
#ifdef __HIPCC__

__device__ __forceinline__ u32 scl_bswap32(u32 v) { return __builtin_bswap32(v); }

// Writes a stream BACK TO FRONT into one slot: every put() lands in front of what is already
// there (the reference prepends every rANS field, rANS.py:196).  The stream ends at the end of
// the slot; 32-bit big-endian words are stored as they fill up.
struct BackBitWriter {
    u8 *slot;
    u64 acc;      // pending bits, right-aligned: the most recently put bits are the high ones
    u32 nacc;     // number of pending bits (< 32 between calls)
    i64 wend;     // byte offset of the right end of the pending region (multiple of 4)
    u64 stride;
    u32 overflow;

    __device__ __forceinline__ void init(u8 *slot_, u64 stride_) {
        slot = slot_;
        stride = stride_;
        acc = 0;
        nacc = 0;
        wend = (i64)stride_;
        overflow = 0;
    }
    __device__ __forceinline__ void flush_word() {
        wend -= 4;
        if (wend >= 0)
            *reinterpret_cast<u32 *>(slot + wend) = scl_bswap32((u32)acc);
        else
            overflow = 1;
        acc >>= 32;
        nacc -= 32;
    }
    // v < 2^w, w <= 32
    __device__ __forceinline__ void put(u32 v, u32 w) {
        acc |= (u64)v << nacc;
        nacc += w;
        if (nacc >= 32) flush_word();
    }
    __device__ __forceinline__ void put64(u64 v, u32 w) {  // w <= 64
        if (w > 32) {
            put((u32)v, 32);
            put((u32)(v >> 32), w - 32);
        } else {
            put((u32)v, w);
        }
    }
    // returns total stream bits (also when the slot overflowed: what it would have needed);
    // the partial leading word is stored with zero bits in front of the stream
    __device__ __forceinline__ u64 finish() {
        const u64 total = (u64)((i64)stride - wend) * 8 + nacc;
        if (nacc) {
            const i64 w = wend - 4;
            if (w >= 0)
                *reinterpret_cast<u32 *>(slot + w) = scl_bswap32((u32)acc);
            else
                overflow = 1;
        }
        return total;
    }
};

// Writes a stream FRONT TO BACK from the start of a slot (range / arithmetic coders append).
struct FwdBitWriter {
    u8 *slot;
    u64 acc;   // pending bits, right-aligned: the oldest pending bit is the most significant
    u32 nacc;
    u64 wpos;  // byte offset where the next full word goes (multiple of 4)
    u64 stride;
    u32 overflow;

    __device__ __forceinline__ void init(u8 *slot_, u64 stride_) {
        slot = slot_;
        stride = stride_;
        acc = 0;
        nacc = 0;
        wpos = 0;
        overflow = 0;
    }
    __device__ __forceinline__ void put(u32 v, u32 w) {  // v < 2^w, w <= 32
        acc = (acc << w) | v;
        nacc += w;
        if (nacc >= 32) {
            u32 word = (u32)(acc >> (nacc - 32));
            if (wpos + 4 <= stride)
                *reinterpret_cast<u32 *>(slot + wpos) = scl_bswap32(word);
            else
                overflow = 1;
            wpos += 4;
            nacc -= 32;
        }
    }
    __device__ __forceinline__ void put64(u64 v, u32 w) {
        if (w > 32) {
            put((u32)(v >> 32), w - 32);
            put((u32)v, 32);
        } else {
            put((u32)v, w);
        }
    }
    // run of `count` identical bits
    __device__ __forceinline__ void put_run(u32 bit, u64 count) {
        while (count >= 32) {
            put(bit ? 0xFFFFFFFFu : 0u, 32);
            count -= 32;
        }
        if (count) put(bit ? ((1u << count) - 1u) : 0u, (u32)count);
    }
    __device__ __forceinline__ u64 finish() {
        u64 total = wpos * 8 + nacc;
        if (nacc) {
            u32 word = (u32)(acc << (32 - nacc));
            if (wpos + 4 <= stride)
                *reinterpret_cast<u32 *>(slot + wpos) = scl_bswap32(word);
            else
                overflow = 1;
        }
        return total;
    }
};

// Forward reader over an arbitrary bit range [pos, end) of a buffer.
struct BitReader {
    const u8 *base;
    u64 pos, end;
    u64 size_bytes;  // readable bytes from base
    u32 truncated;

    __device__ __forceinline__ void init(const u8 *base_, u64 size_bytes_, u64 bit_off, u64 nbits) {
        base = base_;
        size_bytes = size_bytes_;
        pos = bit_off;
        end = bit_off + nbits;
        truncated = 0;
    }
    __device__ __forceinline__ u32 word_at(u64 idx) const {  // big-endian 32-bit word idx
        u64 b = idx * 4;
        if (b + 4 <= size_bytes) return scl_bswap32(*reinterpret_cast<const u32 *>(base + b));
        u32 v = 0;
        for (u32 i = 0; i < 4; ++i) v = (v << 8) | ((b + i < size_bytes) ? base[b + i] : 0u);
        return v;
    }
    // bits [p, p+w) without consuming; bits at or past `end` read as zero.  w <= 32.
    __device__ __forceinline__ u32 peek_at(u64 p, u32 w) const {
        if (w == 0) return 0;
        u64 idx = p >> 5;
        u32 sh = (u32)(p & 31);
        u64 win = ((u64)word_at(idx) << 32) | (sh + w > 32 ? word_at(idx + 1) : 0u);
        u32 v = (u32)((win >> (64 - sh - w)) & (w == 32 ? 0xFFFFFFFFull : ((1ull << w) - 1)));
        if (p + w > end) {  // zero the part that lies past the end
            u64 over = p + w - end;
            v = (over >= w) ? 0u : (u32)(((u64)v >> over) << over);
        }
        return v;
    }
    __device__ __forceinline__ u32 get(u32 w) {  // strict: running past the end is an error
        if (pos + w > end) {
            truncated = 1;
            pos += w;
            return 0;
        }
        u32 v = peek_at(pos, w);
        pos += w;
        return v;
    }
    __device__ __forceinline__ u64 get64(u32 w) {
        if (w > 32) {
            u64 hi = get(w - 32);
            return (hi << 32) | get(32);
        }
        return get(w);
    }
};

// cooperative table load into LDS
template <typename T>
__device__ __forceinline__ void scl_load_table(T *dst, const T *src, u32 n) {
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

#endif  // __HIPCC__
