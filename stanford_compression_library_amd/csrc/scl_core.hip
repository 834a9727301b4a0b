// scl_core.hip -- library plumbing: errors, device query, stream compaction / framing,
// and the single-chunk host drivers behind the drop-in encode_block / decode_block.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "scl_common.h"

// ---- errors ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void scl_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *scl_last_error(void) { return g_err; }

extern "C" int scl_abi_version(void) { return SCL_ABI_VERSION; }

extern "C" int scl_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        scl_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        if (count) *count = 0;
        return SCL_E_NODEVICE;
    }
    if (count) *count = n;
    return n > 0 ? SCL_OK : SCL_E_NODEVICE;
}

int scl_current_device(void) {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return d;
}

int scl_check_device(int model_device, const char *what) {
    const int cur = scl_current_device();
    if (cur == model_device) return SCL_OK;
    scl_set_error("%s: the model was created on device %d but the current device is %d (one model per device: "
                  "create a handle on every GPU that uses it, or hipSetDevice before the call)", what, model_device, cur);
    return SCL_E_PARAM;
}

// ---- rows the tuned kernels cannot take as they are ----------------------------------------------------
// per thread: -1 = follow the environment (read at every call), 0 = the library chooses, 1 = any-parameter kernels only
static thread_local int tl_any_parameter = -1;
extern "C" int scl_set_any_parameter_kernels(int on) {
    const int prev = tl_any_parameter;
    tl_any_parameter = on < 0 ? -1 : (on ? 1 : 0);
    return prev;
}
bool scl_force_generic(void) {
    if (tl_any_parameter >= 0) return tl_any_parameter == 1;
    const char *e = getenv("SCL_ANY_PARAMETER_KERNELS");
    return e && e[0] == '1';
}

// dst[r * dst_stride + b] = src[r * src_stride + b] for b < row_bytes: four bytes per lane, no alignment assumed
// (row_lens, when given: only the first row_lens[r] bytes of row r -- what a decoder really produced.  A length ABOVE
// row_bytes is a chunk the decoder refused (SCL_ST_CAPACITY: the tuned decoders store the header's n to out_lens before
// they check it against out_cap, and decode nothing): nothing of such a row goes back -- the scratch behind it is
// uninitialised pool memory)
__global__ void __launch_bounds__(256) relay_rows_kernel(u8 *__restrict__ dst, u64 dst_stride, const u8 *__restrict__ src,
                                                        u64 src_stride, u32 row_bytes, u64 n_rows,
                                                        const u32 *__restrict__ row_lens, u32 sym_bytes) {
    const u32 quads = (row_bytes + 3) / 4;
    const u64 total = n_rows * quads;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += (u64)gridDim.x * 256) {
        const u64 r = i / quads;
        const u32 b = (u32)(i - r * quads) * 4;
        u32 len = row_bytes;
        if (row_lens) {
            len = row_lens[r] * sym_bytes;
            if (row_lens[r] > row_bytes / sym_bytes) len = 0;  // refused chunk: see above
        }
        if (b >= len) continue;
        const u8 *s = src + r * src_stride + b;
        u8 *d = dst + r * dst_stride + b;
        const u32 n = min(4u, len - b);
        for (u32 j = 0; j < n; ++j) d[j] = s[j];
    }
}

static int relay_launch(u8 *dst, u64 dst_stride, const u8 *src, u64 src_stride, u32 row_bytes, u64 n_rows, hipStream_t st,
                        const u32 *row_lens = nullptr, u32 sym_bytes = 1) {
    if (!n_rows || !row_bytes) return SCL_OK;
    const u64 total = n_rows * ((row_bytes + 3) / 4);
    u64 blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(relay_rows_kernel, dim3((u32)blocks), dim3(256), 0, st, dst, dst_stride, src, src_stride, row_bytes,
                       n_rows, row_lens, sym_bytes);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

int RowRelay::in(const u8 *&d_sym, u64 &sym_stride, u32 chunk_len, u64 n_chunks, hipStream_t stream) {
    if (scl_rows_aligned(d_sym, sym_stride) || n_chunks == 0) return SCL_OK;
    if (n_chunks == 1 && ((uintptr_t)d_sym & 15) == 0) {  // one row: its stride is free
        sym_stride = scl_round_up((u64)chunk_len + 1, 16);
        return SCL_OK;
    }
    st = stream;
    stride = scl_round_up((u64)chunk_len, 16);
    if (stride == 0) stride = 16;
    hipError_t e = hipMallocAsync((void **)&scratch, n_chunks * stride + 16, st);
    if (e != hipSuccess) {
        // no scratch, no re-laying: the caller's rows stay as they are, which sends the call to the any-parameter kernels
        // (they take any alignment and need no scratch) instead of failing it.  The failure is RECORDED -- `failed`, and the
        // text scl_last_error() returns -- so that a caller without such a fallback (a table-less tANS model) can report
        // SCL_E_ALLOC instead of a misleading alignment message, and a silent ~100x slowdown can be explained.
        scratch = nullptr;
        failed = true;
        (void)hipGetLastError();
        scl_set_error("row relay: hipMallocAsync(%llu bytes) failed (%s): rows not re-laid, the any-parameter kernels serve "
                      "this call", (unsigned long long)(n_chunks * stride + 16), hipGetErrorString(e));
        return SCL_OK;
    }
    if (int rc = relay_launch(scratch, stride, d_sym, sym_stride, chunk_len, n_chunks, st)) return rc;
    d_sym = scratch;
    sym_stride = stride;
    return SCL_OK;
}

int RowRelay::out_begin(u8 *&d_out, u64 &out_stride, u32 out_cap, u64 n_chunks, hipStream_t stream) {
    if (scl_rows_aligned(d_out, out_stride) || n_chunks == 0) return SCL_OK;
    st = stream;
    stride = scl_round_up((u64)out_cap + 1, 16);
    hipError_t e = hipMallocAsync((void **)&scratch, n_chunks * stride + 16, st);
    if (e != hipSuccess) {  // as above: the any-parameter kernels store to the caller's rows directly
        scratch = nullptr;
        failed = true;
        (void)hipGetLastError();
        scl_set_error("row relay: hipMallocAsync(%llu bytes) failed (%s): output rows not re-laid, the any-parameter kernels "
                      "serve this call", (unsigned long long)(n_chunks * stride + 16), hipGetErrorString(e));
        return SCL_OK;
    }
    user_out = d_out;
    user_stride = out_stride;
    n_rows = n_chunks;
    row_bytes = out_cap;
    d_out = scratch;
    out_stride = stride;
    return SCL_OK;
}

// d_out_lens: the decoder's per-row symbol counts (device).  Only those bytes go back: whatever else the scratch holds --
// stale pool memory behind a row's symbols, the whole row of a chunk that failed -- never reaches the caller's buffer.
int RowRelay::out_end(const u32 *d_out_lens) {
    if (!user_out) return SCL_OK;
    return relay_launch(user_out, user_stride, scratch, stride, row_bytes * sym_bytes, n_rows, st, d_out_lens, sym_bytes);
}

RowRelay::~RowRelay() {
    if (scratch) (void)hipFreeAsync(scratch, st);
}

// ---- stream compaction ----------------------------------------------------------------------------
// Layout of one output record:
//   DENSE : ceil(nbits/8) bytes, stream left-aligned, zero tail bits
//   FRAMED: [u32 BE payload_bytes][payload], payload = 3-bit pad count, pad zeros, stream bits
//           (Padder.add_byte_padding + HeaderHandler.add_header, encoded_stream.py:23-46,94-103)
#define CP_THREADS 256
#define CP_ITEMS 8
#define CP_TILE (CP_THREADS * CP_ITEMS)

__device__ __forceinline__ u64 cp_record_bytes(u32 nbits, int mode) {
    if (mode == SCL_COMPACT_FRAMED) return 4ull + (((u64)nbits + 3 + 7) >> 3);
    return ((u64)nbits + 7) >> 3;
}

// pass 1: per-tile exclusive scan of record sizes -> d_off (tile-local), d_tile_sum
__global__ void __launch_bounds__(CP_THREADS) cp_scan_tiles(const u32 *__restrict__ nbits, u64 n, int mode,
                                                           u64 *__restrict__ off, u64 *__restrict__ tile_sum) {
    __shared__ u64 s_part[CP_THREADS];
    const u64 base = (u64)blockIdx.x * CP_TILE + (u64)threadIdx.x * CP_ITEMS;
    u64 v[CP_ITEMS], run = 0;
    for (int i = 0; i < CP_ITEMS; ++i) {
        u64 idx = base + i;
        v[i] = run;
        run += (idx < n) ? cp_record_bytes(nbits[idx], mode) : 0;
    }
    s_part[threadIdx.x] = run;
    __syncthreads();
    // Hillis-Steele over the 256 per-thread sums
    for (u32 d = 1; d < CP_THREADS; d <<= 1) {
        u64 t = (threadIdx.x >= d) ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += t;
        __syncthreads();
    }
    const u64 excl = s_part[threadIdx.x] - run;
    for (int i = 0; i < CP_ITEMS; ++i) {
        u64 idx = base + i;
        if (idx < n) off[idx] = excl + v[i];
    }
    if (threadIdx.x == CP_THREADS - 1) tile_sum[blockIdx.x] = s_part[threadIdx.x];
}

// pass 2: one workgroup scans the tile sums in place (exclusive) and stores the grand total
__global__ void __launch_bounds__(CP_THREADS) cp_scan_sums(u64 *__restrict__ tile_sum, u64 n_tiles,
                                                          u64 *__restrict__ total_out) {
    __shared__ u64 s_part[CP_THREADS];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 start = 0; start < n_tiles; start += CP_THREADS) {
        u64 idx = start + threadIdx.x;
        u64 mine = (idx < n_tiles) ? tile_sum[idx] : 0;
        s_part[threadIdx.x] = mine;
        __syncthreads();
        for (u32 d = 1; d < CP_THREADS; d <<= 1) {
            u64 t = (threadIdx.x >= d) ? s_part[threadIdx.x - d] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        if (idx < n_tiles) tile_sum[idx] = s_carry + s_part[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == CP_THREADS - 1) s_carry += s_part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

// pass 3: add tile bases; entry n = total
// (base: where the first record starts, a value in device memory -- scl_streams_compact_at; tile_sum[n_tiles + 1] holds it)
__global__ void cp_add_base(u64 *__restrict__ off, u64 n, const u64 *__restrict__ tile_sum,
                            const u64 *__restrict__ total, const u64 *__restrict__ base) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 b = base ? *base : 0;
    if (idx < n) off[idx] += tile_sum[idx / CP_TILE] + b;
    if (idx == n) off[n] = *total + b;
}

// pass 4: one wavefront per stream copies (bit-shifts) it to its record.
// The record payload is `lead` bits (FRAMED: 3-bit pad count + pad zeros; DENSE: none) followed by
// the stream; payload bytes [q, q+w) are assembled from at most 32 source bits.
__device__ __forceinline__ u32 cp_payload_bits(const BitReader &r, u64 q, u32 w_bytes, u32 lead, u32 pad) {
    const u64 b0 = 8 * q;
    const u32 wb = 8 * w_bytes;
    if (b0 >= lead) return r.peek_at(r.pos + b0 - lead, wb);
    u32 v = 0;
    if (b0 + wb > lead) v = r.peek_at(r.pos, (u32)(b0 + wb - lead));
    if (q == 0) v |= pad << (wb - 3);
    return v;
}

// The wave walks the record in 16-byte blocks ALIGNED IN THE DESTINATION (block k = bytes [16k - a, 16k - a + 16) of
// the payload, a = payload address mod 16), four blocks per lane and trip:
//   * an interior block (all 16 bytes inside the record, its 160 source bits inside the stream and past the lead
//     bits) is five aligned source words funnel-shifted into one aligned 16-byte store; the loads of all four
//     blocks are issued before the first store, so a typical 3.7 KiB record costs the wave ONE round trip to memory
//     for its bulk (the previous version -- head bytes, words up to the first boundary, bulk loop, tail words, tail
//     bytes, each a loop of its own waiting for its own loads -- took eleven, and a wave lived 18 us);
//   * the bytes before the first and after the last interior block go through cp_payload_bits, one byte per lane.
__global__ void __launch_bounds__(256) cp_copy(const u8 *__restrict__ in, const u64 *__restrict__ bit_off,
                                              const u32 *__restrict__ nbits, u64 n, int mode, u8 *__restrict__ out,
                                              u64 out_capacity, const u64 *__restrict__ rec_off) {
    const u32 lane = threadIdx.x & (SCL_WAVE - 1);
    const u64 c = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / SCL_WAVE;
    if (c >= n) return;
    const u64 o0 = rec_off[c], o1 = rec_off[c + 1];
    const u32 nb = nbits[c];
    const u64 boff = bit_off[c];
    if (o1 > out_capacity) return;  // caller sees the required size in rec_off[n]
    BitReader r;
    r.init(in, ~0ull >> 1, boff, nb);  // every stream byte lies inside the caller's buffer
    u8 *dst = out + o0;
    u64 rec_bytes = o1 - o0;
    u32 lead = 0, pad = 0;
    if (mode == SCL_COMPACT_FRAMED) {
        const u32 payload = (u32)(rec_bytes - 4);
        pad = payload * 8 - 3 - nb;
        lead = 3 + pad;
        if (lane == 0) {
            dst[0] = (u8)(payload >> 24);
            dst[1] = (u8)(payload >> 16);
            dst[2] = (u8)(payload >> 8);
            dst[3] = (u8)payload;
        }
        dst += 4;
        rec_bytes -= 4;
    }
    const u32 a = (u32)(reinterpret_cast<uintptr_t>(dst) & 15);
    u8 *dst_al = dst - a;  // block k lives at dst_al + 16 k
    const u32 *in32 = reinterpret_cast<const u32 *>(in);
    const i64 src_base = (i64)boff - (i64)lead - 8 * (i64)a;  // source bit of byte 0 of block 0 (may lie before the stream)
    const i64 src_al = src_base & ~31ll;                       // ... rounded down to a word
    const u32 sh = (u32)src_base & 31u;                        // the same for every block (128 k is a multiple of 32)
    const i64 stream_end = (i64)boff + (i64)nb;
    // interior blocks k_first .. k_last: inside the record, past the lead bits, five source words inside the stream
    const i64 k_first = max((i64)((a + 15) / 16), ((i64)lead + 8 * (i64)a + 127) / 128);
    i64 k_last = (i64)((a + rec_bytes) / 16) - 1;
    {
        const i64 room = stream_end - 160 - src_al;
        const i64 k_src = room >= 0 ? room / 128 : -1;
        if (k_src < k_last) k_last = k_src;
    }
    const bool any_fast = k_last >= k_first;
    const u64 q_lo = any_fast ? (u64)(16 * k_first - a) : rec_bytes;  // payload bytes [0, q_lo) and [q_hi, rec_bytes)
    const u64 q_hi = any_fast ? (u64)(16 * (k_last + 1) - a) : rec_bytes;  // are the edges: one byte per lane
    for (u64 q = lane; q < q_lo; q += SCL_WAVE) dst[q] = (u8)cp_payload_bits(r, q, 1, lead, pad);
    for (u64 q = q_hi + lane; q < rec_bytes; q += SCL_WAVE) dst[q] = (u8)cp_payload_bits(r, q, 1, lead, pad);
    for (i64 k0 = k_first; k0 <= k_last; k0 += 4 * SCL_WAVE) {
        u32 w[4][5];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 k = k0 + (i64)j * SCL_WAVE + lane;
            if (k <= k_last) {
                const u32 *src = in32 + ((src_al + 128 * k) >> 5);
#pragma unroll
                for (int i = 0; i < 5; ++i) w[j][i] = src[i];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 k = k0 + (i64)j * SCL_WAVE + lane;
            if (k <= k_last) {
                const u32 b0 = scl_bswap32(w[j][0]), b1 = scl_bswap32(w[j][1]), b2 = scl_bswap32(w[j][2]);
                const u32 b3 = scl_bswap32(w[j][3]), b4 = scl_bswap32(w[j][4]);
                uint4 o;  // (x << sh) | (y >> (32 - sh)); v_alignbit takes its shift modulo 32, hence the select
                o.x = sh ? __builtin_amdgcn_alignbit(b0, b1, 32 - sh) : b0;
                o.y = sh ? __builtin_amdgcn_alignbit(b1, b2, 32 - sh) : b1;
                o.z = sh ? __builtin_amdgcn_alignbit(b2, b3, 32 - sh) : b2;
                o.w = sh ? __builtin_amdgcn_alignbit(b3, b4, 32 - sh) : b3;
                o.x = scl_bswap32(o.x), o.y = scl_bswap32(o.y), o.z = scl_bswap32(o.z), o.w = scl_bswap32(o.w);
                typedef u32 u32x4_nt __attribute__((ext_vector_type(4)));
                const u32x4_nt t = {o.x, o.y, o.z, o.w};  // written once, whole sectors per quad of lanes
                __builtin_nontemporal_store(t, reinterpret_cast<u32x4_nt *>(dst_al + 16 * k));
            }
        }
    }
}

// pass 4 for WAVE-STRIPED slots (ABI version 8; scl_ans_fast_io.h: byte b of the logical slot of stream c lives at
// (c / 64) * 64 * stride + (b / 16) * 1024 + 16 * (c % 64) + b % 16): the same record, byte for byte, as cp_copy writes
// from linear slots.  A destination block's 160 source bits are five consecutive logical words = parts of two consecutive 16-byte pieces;
// 128 k bits further is exactly one piece further, so the offset of the first word inside its piece is the same for every
// block of a stream: two 16-byte loads per block and a choice of five words out of eight.
struct StripedSrc {
    const u32 *w32;  // word 0 of this stream's logical slot (piece 0)
    u64 end;         // first bit after the stream, relative to the slot start
    __device__ __forceinline__ u32 word_at(u64 idx) const {  // big-endian logical word idx; zero past the stream's last word
        if (idx * 32 >= end) return 0;
        return scl_bswap32(w32[(idx >> 2) * 256 + (idx & 3)]);
    }
    __device__ __forceinline__ u32 peek_at(u64 p, u32 w) const {  // bits [p, p + w), zero at or past `end`; w <= 32
        if (w == 0) return 0;
        const u64 idx = p >> 5;
        const u32 sh = (u32)(p & 31);
        const u64 win = ((u64)word_at(idx) << 32) | (sh + w > 32 ? word_at(idx + 1) : 0u);
        u32 v = (u32)((win >> (64 - sh - w)) & (w == 32 ? 0xFFFFFFFFull : ((1ull << w) - 1)));
        if (p + w > end) {
            const u64 over = p + w - end;
            v = (over >= w) ? 0u : (u32)(((u64)v >> over) << over);
        }
        return v;
    }
};
__device__ __forceinline__ u32 cps_payload_bits(const StripedSrc &r, u64 pos, u64 q, u32 w_bytes, u32 lead, u32 pad) {
    const u64 b0 = 8 * q;
    const u32 wb = 8 * w_bytes;
    if (b0 >= lead) return r.peek_at(pos + b0 - lead, wb);
    u32 v = 0;
    if (b0 + wb > lead) v = r.peek_at(pos, (u32)(b0 + wb - lead));
    if (q == 0) v |= pad << (wb - 3);
    return v;
}
// Work split: one wavefront = the EIGHT streams of a lane group (c = 8 g .. 8 g + 7: the lanes that share every 128-byte
// line of their rows).  Lane (k, i) = (lane & 7, lane >> 3) works for stream k on the piece rows r0 + i, r0 + i + 8, ...:
// a load instruction then reads 8 rows x 8 adjacent pieces = eight WHOLE lines, and the eight lanes of a stream write
// eight consecutive 16-byte blocks of its record (a 128-byte run) per store instruction.  (One wave per stream, as
// cp_copy has it, made every load instruction touch 64 lines, 16 bytes of each: 0.80 ms per GiB batch against 0.38.)
#define CPS_THREADS 256
__global__ void __launch_bounds__(CPS_THREADS) cp_copy_striped(const u8 *__restrict__ in, u64 stride,
                                                               const u64 *__restrict__ bit_off,
                                                               const u32 *__restrict__ nbits, u64 n, int mode,
                                                               u8 *__restrict__ out, u64 out_capacity,
                                                               const u64 *__restrict__ rec_off) {
    const u32 lane = threadIdx.x & (SCL_WAVE - 1);
    const u32 k = lane & 7u, i = lane >> 3;
    const u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / SCL_WAVE;  // lane group
    const u64 c = g * 8 + k;
    bool live = c < n && rec_off[min(c + 1, n)] <= out_capacity;  // caller sees the required size in rec_off[n]
    // a descriptor that does not lie inside its own logical slot (the contract of the striped entry points) is not followed
    // out of the buffer: its record is left unwritten
    if (live) {
        const u64 rel = bit_off[c] - c * stride * 8;
        live = rel <= stride * 8 && (u64)nbits[c] <= stride * 8 - rel;
    }
    // ---- this lane's stream (the eight lanes of a stream compute the same values) -----------------------------------
    i64 k_first = 1, k_last = 0, rowbase = 0;
    u32 sh = 0, o = 0;
    u8 *dst_al = nullptr;
    const u8 *slot = in + ((c >> 6) * (stride << 6) + ((c & 63u) << 4));
    if (live) {
        const u64 o0 = rec_off[c], o1 = rec_off[c + 1];
        const u32 nb = nbits[c];
        const u64 pos = bit_off[c] - c * stride * 8;  // the stream's first bit inside its logical slot
        StripedSrc r;
        r.w32 = reinterpret_cast<const u32 *>(slot);
        r.end = pos + nb;
        u8 *dst = out + o0;
        u64 rec_bytes = o1 - o0;
        u32 lead = 0, pad = 0;
        if (mode == SCL_COMPACT_FRAMED) {
            const u32 payload = (u32)(rec_bytes - 4);
            pad = payload * 8 - 3 - nb;
            lead = 3 + pad;
            if (i == 0) {
                dst[0] = (u8)(payload >> 24);
                dst[1] = (u8)(payload >> 16);
                dst[2] = (u8)(payload >> 8);
                dst[3] = (u8)payload;
            }
            dst += 4;
            rec_bytes -= 4;
        }
        const u32 a = (u32)(reinterpret_cast<uintptr_t>(dst) & 15);
        dst_al = dst - a;  // block kk lives at dst_al + 16 kk
        const i64 src_base = (i64)pos - (i64)lead - 8 * (i64)a;  // source bit of byte 0 of block 0 (may lie before the stream)
        const i64 src_al = src_base & ~31ll;
        sh = (u32)src_base & 31u;
        const i64 stream_end = (i64)pos + (i64)nb;
        // interior blocks k_first .. k_last: inside the record, past the lead bits, five source words inside the stream
        k_first = max((i64)((a + 15) / 16), ((i64)lead + 8 * (i64)a + 127) / 128);
        k_last = (i64)((a + rec_bytes) / 16) - 1;
        {
            const i64 room = stream_end - 160 - src_al;
            const i64 k_src = room >= 0 ? room / 128 : -1;
            if (k_src < k_last) k_last = k_src;
        }
        const bool any_fast = k_last >= k_first;
        const u64 q_lo = any_fast ? (u64)(16 * k_first - a) : rec_bytes;  // payload bytes [0, q_lo) and [q_hi, rec_bytes)
        const u64 q_hi = any_fast ? (u64)(16 * (k_last + 1) - a) : rec_bytes;  // are the edges: one byte per lane of the stream
        for (u64 q = i; q < q_lo; q += 8) dst[q] = (u8)cps_payload_bits(r, pos, q, 1, lead, pad);
        for (u64 q = q_hi + i; q < rec_bytes; q += 8) dst[q] = (u8)cps_payload_bits(r, pos, q, 1, lead, pad);
        // word (src_al >> 5) + 4 kk is the first of block kk's five: piece rowbase + kk, word offset o in it
        const i64 w00 = src_al >> 5;  // may be negative only when no interior block exists
        o = (u32)(w00 & 3);
        rowbase = w00 >> 2;
        if (!any_fast) k_first = 1, k_last = 0;
    }
    // ---- every stream walks ITS OWN blocks in groups of eight that start on a 128-byte line of the destination: lane (k, i)
    // takes block kk0 + 8 t + i of stream k, so the eight lanes of a stream store one whole, aligned line per instruction
    const i64 kk0 = live ? (k_first - (i64)(((reinterpret_cast<uintptr_t>(dst_al) >> 4) + (u64)k_first) & 7u)) : 0;
    i64 trips = (live && k_last >= k_first) ? (k_last - kk0) / 8 + 1 : 0;
    for (int d = 1; d < 8; d <<= 1) trips = max(trips, (i64)__shfl_xor((long long)trips, d));
    const uint4 *piece = reinterpret_cast<const uint4 *>(slot);
    for (i64 t0 = 0; t0 < trips; t0 += 4) {
        uint4 pa[4], pb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 kk = kk0 + 8 * (t0 + j) + i, row = rowbase + kk;
            if (kk >= k_first && kk <= k_last) {
                pa[j] = piece[row * 64];
                pb[j] = piece[(row + 1) * 64];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 kk = kk0 + 8 * (t0 + j) + i;
            if (kk >= k_first && kk <= k_last) {
                const u32 w8[8] = {pa[j].x, pa[j].y, pa[j].z, pa[j].w, pb[j].x, pb[j].y, pb[j].z, pb[j].w};
                u32 b[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) {  // b[t] = word o + t of the eight (o differs from stream to stream)
                    const u32 lo2 = (o & 1u) ? w8[t + 1] : w8[t];
                    const u32 hi2 = (o & 1u) ? w8[t + 3] : w8[t + 2];
                    b[t] = scl_bswap32((o & 2u) ? hi2 : lo2);
                }
                uint4 v;  // (x << sh) | (y >> (32 - sh)); v_alignbit takes its shift modulo 32, hence the select
                v.x = sh ? __builtin_amdgcn_alignbit(b[0], b[1], 32 - sh) : b[0];
                v.y = sh ? __builtin_amdgcn_alignbit(b[1], b[2], 32 - sh) : b[1];
                v.z = sh ? __builtin_amdgcn_alignbit(b[2], b[3], 32 - sh) : b[2];
                v.w = sh ? __builtin_amdgcn_alignbit(b[3], b[4], 32 - sh) : b[3];
                v.x = scl_bswap32(v.x), v.y = scl_bswap32(v.y), v.z = scl_bswap32(v.z), v.w = scl_bswap32(v.w);
                typedef u32 u32x4_nt __attribute__((ext_vector_type(4)));
                const u32x4_nt t4 = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(t4, reinterpret_cast<u32x4_nt *>(dst_al + 16 * kk));
            }
        }
    }
}

extern "C" uint64_t scl_streams_compact_scratch_bytes(uint64_t n_chunks) {
    u64 n_tiles = (n_chunks + CP_TILE - 1) / CP_TILE;
    return scl_round_up((n_tiles + 3) * sizeof(u64), 256);  // tile sums, the total, the base of scl_streams_compact_at
}

// The same with the first record at byte *d_base of d_out, d_base in DEVICE memory (NULL: 0): the offsets written are
// absolute (d_out_byte_offset[0] = *d_base, [n] = where the next record would start).  A batch can so be compacted in
// sub-batches into ONE dense buffer without the host ever knowing the sizes: sub-batch i + 1 passes the address of
// sub-batch i's last offset entry -- which may be the very address it writes its own first entry to (the value is read
// into the scratch before anything is written).  What lets the compaction of one sub-batch run on a second stream while the
// next one is still being encoded (backend/models.py encode_dense_pipelined).
// striped_stride != 0: d_in holds wave-striped slots of that stride (stream c inside logical slot c)
static int compact_impl(const uint8_t *d_in, u64 striped_stride, const uint64_t *d_bit_offset, const uint32_t *d_nbits,
                        uint64_t n_chunks, int mode, uint8_t *d_out, uint64_t out_capacity,
                        uint64_t *d_out_byte_offset, const uint64_t *d_base, void *d_scratch, void *stream) {
    SCL_REQUIRE(mode == SCL_COMPACT_DENSE || mode == SCL_COMPACT_FRAMED, "compact: unknown mode %d", mode);
    SCL_REQUIRE(d_in && d_bit_offset && d_nbits && d_out && d_out_byte_offset && d_scratch,
                "compact: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const u64 n_tiles = (n_chunks + CP_TILE - 1) / CP_TILE;
    u64 *tile_sum = (u64 *)d_scratch;
    u64 *total = tile_sum + n_tiles;
    u64 *base = nullptr;
    if (d_base) {  // first, before any offset entry is written: d_base may alias d_out_byte_offset[0]
        base = total + 1;
        SCL_HIP_TRY(hipMemcpyAsync(base, d_base, sizeof(u64), hipMemcpyDeviceToDevice, st));
    }
    if (n_chunks == 0) {
        if (base)
            SCL_HIP_TRY(hipMemcpyAsync(d_out_byte_offset, base, sizeof(u64), hipMemcpyDeviceToDevice, st));
        else
            SCL_HIP_TRY(hipMemsetAsync(d_out_byte_offset, 0, sizeof(u64), st));
        return SCL_OK;
    }
    hipLaunchKernelGGL(cp_scan_tiles, dim3((u32)n_tiles), dim3(CP_THREADS), 0, st, d_nbits, n_chunks, mode,
                       d_out_byte_offset, tile_sum);
    hipLaunchKernelGGL(cp_scan_sums, dim3(1), dim3(CP_THREADS), 0, st, tile_sum, n_tiles, total);
    hipLaunchKernelGGL(cp_add_base, dim3((u32)((n_chunks + 1 + 255) / 256)), dim3(256), 0, st, d_out_byte_offset,
                       n_chunks, tile_sum, total, base);
    if (striped_stride) {
        const u64 spb = 8 * (CPS_THREADS / SCL_WAVE);  // streams per workgroup: eight per wavefront
        hipLaunchKernelGGL(cp_copy_striped, dim3((u32)((n_chunks + spb - 1) / spb)), dim3(CPS_THREADS), 0, st, d_in,
                           striped_stride, d_bit_offset, d_nbits, n_chunks, mode, d_out, out_capacity, d_out_byte_offset);
    } else {
        const u64 waves_per_block = 256 / SCL_WAVE;
        hipLaunchKernelGGL(cp_copy, dim3((u32)((n_chunks + waves_per_block - 1) / waves_per_block)), dim3(256), 0, st,
                           d_in, d_bit_offset, d_nbits, n_chunks, mode, d_out, out_capacity, d_out_byte_offset);
    }
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

extern "C" int scl_streams_compact_at(const uint8_t *d_in, const uint64_t *d_bit_offset, const uint32_t *d_nbits,
                                      uint64_t n_chunks, int mode, uint8_t *d_out, uint64_t out_capacity,
                                      uint64_t *d_out_byte_offset, const uint64_t *d_base, void *d_scratch,
                                      void *stream) {
    return compact_impl(d_in, 0, d_bit_offset, d_nbits, n_chunks, mode, d_out, out_capacity, d_out_byte_offset, d_base,
                        d_scratch, stream);
}

// (ABI version 8) scl_streams_compact_at for WAVE-STRIPED slots of `in_stride` bytes (what scl_rans_encode_batch_striped /
// scl_tans_encode_batch_striped write): the same dense / framed records, byte for byte, as the linear form produces.
extern "C" int scl_streams_compact_striped(const uint8_t *d_in, uint64_t in_stride, const uint64_t *d_bit_offset,
                                           const uint32_t *d_nbits, uint64_t n_chunks, int mode, uint8_t *d_out,
                                           uint64_t out_capacity, uint64_t *d_out_byte_offset, const uint64_t *d_base,
                                           void *d_scratch, void *stream) {
    SCL_REQUIRE(in_stride % 16 == 0 && in_stride > 0 && in_stride < (1ull << 24) && ((uintptr_t)d_in & 15) == 0,
                "compact_striped: d_in must be 16-byte aligned and in_stride a multiple of 16 below 2^24");
    return compact_impl(d_in, in_stride, d_bit_offset, d_nbits, n_chunks, mode, d_out, out_capacity, d_out_byte_offset,
                        d_base, d_scratch, stream);
}

extern "C" int scl_streams_compact(const uint8_t *d_in, const uint64_t *d_bit_offset, const uint32_t *d_nbits,
                                   uint64_t n_chunks, int mode, uint8_t *d_out, uint64_t out_capacity,
                                   uint64_t *d_out_byte_offset, void *d_scratch, void *stream) {
    return scl_streams_compact_at(d_in, d_bit_offset, d_nbits, n_chunks, mode, d_out, out_capacity, d_out_byte_offset,
                                  nullptr, d_scratch, stream);
}

// ---- host-side index of a framed block file (row f1 / f2) ----------------------------------------------
// The serial walk EncodedBlockReader.get_block makes one record at a time (encoded_stream.py:196-225: 4-byte big-endian
// payload size, payload; Padder.remove_byte_padding :48-58: 3-bit pad count, the pad bits, then the block's bits), for a
// whole buffer: every header says where the next record starts, so the walk cannot be split -- but it is a few
// instructions per record in C against microseconds in Python, and it is all the host has to do before a batched decode.
// The block's own DATA_BLOCK_SIZE_BITS header (the first size_bits bits of its stream) is read out as well: the caller
// sizes the decoder's output rows from it.
extern "C" int scl_framed_index_host(const uint8_t *h_buf, uint64_t buf_size, uint32_t size_bits, uint64_t max_records,
                                     uint64_t *h_bit_offset, uint64_t *h_nbits, uint64_t *h_block_size,
                                     uint64_t *n_records, uint64_t *consumed) {
    SCL_REQUIRE(n_records && consumed, "framed index: null count pointer");
    *n_records = 0;
    *consumed = 0;
    SCL_REQUIRE(h_buf || buf_size == 0, "framed index: null buffer");
    SCL_REQUIRE(size_bits <= 64, "framed index: size_bits %u > 64", size_bits);
    SCL_REQUIRE(max_records == 0 || (h_bit_offset && h_nbits && h_block_size), "framed index: null output array");
    u64 pos = 0, n = 0;
    while (n < max_records && pos + 4 <= buf_size) {
        const u64 payload = ((u64)h_buf[pos] << 24) | ((u64)h_buf[pos + 1] << 16) | ((u64)h_buf[pos + 2] << 8) | h_buf[pos + 3];
        if (payload > buf_size - pos - 4) break;  // incomplete record: the caller reads on (or reports a truncated file)
        if (payload == 0) {  // EncodedBlockWriter never writes one: at least the three pad-count bits are there
            *n_records = n;
            *consumed = pos;
            scl_set_error("framed index: record %llu at byte %llu has an empty payload", (unsigned long long)n,
                          (unsigned long long)pos);
            return SCL_E_PARAM;
        }
        const u32 pad = h_buf[pos + 4] >> 5;
        const u64 o = 8 * (pos + 4) + 3 + pad;
        if (8 * payload < 3 + (u64)pad + size_bits) {
            *n_records = n;
            *consumed = pos;
            scl_set_error("framed index: record %llu at byte %llu is shorter than its padding and its %u-bit size header",
                          (unsigned long long)n, (unsigned long long)pos, size_bits);
            return SCL_E_PARAM;
        }
        u64 v = 0;  // the first size_bits bits of the stream, MSB first
        for (u32 b = 0; b < size_bits; ++b) {
            const u64 bit = o + b;
            v = (v << 1) | ((h_buf[bit >> 3] >> (7 - (bit & 7))) & 1u);
        }
        h_bit_offset[n] = o;
        h_nbits[n] = 8 * payload - 3 - pad;
        h_block_size[n] = v;
        ++n;
        pos += 4 + payload;
    }
    *n_records = n;
    *consumed = pos;
    return SCL_OK;
}

// ---- symbol histogram (row f3) ----------------------------------------------------------------------
// 16 bytes per lane per load; 32 sub-histograms in LDS, one per lane modulo 32, each 257 words long -- bin s of
// sub-histogram j sits in bank (j + s) mod 32, so the 64 lanes of a ds_add that all hold the SAME symbol (runs of zeros,
// text: the inputs a frequency model is built for) hit 32 different banks two at a time instead of one address 64 at a
// time (round 5: a constant input 3.53 -> see tools/time_histogram.py; random bytes unchanged, they are bound by the rate
// of LDS atomics as such).  One global atomic per non-empty bin and workgroup at the end.
#define HG_SUB 32
#define HG_STRIDE 257
__global__ void __launch_bounds__(256) histogram_u8_kernel(const uint4 *__restrict__ sym16, u64 n16,
                                                          const u8 *__restrict__ tail, u32 n_tail,
                                                          unsigned long long *__restrict__ counts) {
    __shared__ u32 s_h[HG_SUB * HG_STRIDE];
    for (u32 i = threadIdx.x; i < HG_SUB * HG_STRIDE; i += 256) s_h[i] = 0;
    __syncthreads();
    u32 *h = s_h + (threadIdx.x & (HG_SUB - 1)) * HG_STRIDE;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) {
        const uint4 v = sym16[i];
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            atomicAdd(&h[w[d] & 0xFF], 1u);
            atomicAdd(&h[(w[d] >> 8) & 0xFF], 1u);
            atomicAdd(&h[(w[d] >> 16) & 0xFF], 1u);
            atomicAdd(&h[w[d] >> 24], 1u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < n_tail) atomicAdd(&h[tail[threadIdx.x]], 1u);
    __syncthreads();
    u32 total = 0;
#pragma unroll 8
    for (u32 j = 0; j < HG_SUB; ++j) total += s_h[j * HG_STRIDE + threadIdx.x];
    if (total) atomicAdd(&counts[threadIdx.x], (unsigned long long)total);
}

extern "C" int scl_histogram_u8(const uint8_t *d_sym, uint64_t n, uint64_t *d_counts, void *stream) {
    SCL_REQUIRE(d_counts && (d_sym || n == 0), "histogram_u8: null pointer argument");
    SCL_REQUIRE(((uintptr_t)d_sym & 15) == 0, "histogram_u8: d_sym must be 16-byte aligned");
    if (n == 0) return SCL_OK;
    const u64 n16 = n >> 4;
    u32 blocks = (u32)((n16 + 255) / 256);
    if (blocks > 1024) blocks = 1024;  // grid-stride above 4 workgroups per CU (33 KiB of LDS each)
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(histogram_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4 *>(d_sym), n16, d_sym + (n16 << 4), (u32)(n & 15),
                       reinterpret_cast<unsigned long long *>(d_counts));
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

// uint16 symbol indices (alphabets up to 65536, ABI 4): workgroup-private u32 histograms in LDS when the alphabet fits
// (K <= 16384: 64 KiB), one global atomic per non-zero bin per workgroup; larger alphabets count straight into HBM.
// 16 bytes (eight symbols) per lane per load over the 16-byte aligned body of the array (round 5: 2-byte loads moved
// 1.0 TB/s); the few symbols in front of and behind it are counted by workgroup 0.
__device__ __forceinline__ void hg16_count(u32 s, u32 K, u32 lds_bins, u32 *s_bins, unsigned long long *counts, u32 &n_bad) {
    if (s >= K)
        ++n_bad;
    else if (lds_bins)
        atomicAdd(&s_bins[s], 1u);
    else
        atomicAdd(&counts[s], 1ull);
}
__global__ void __launch_bounds__(256) histogram_u16_kernel(const u16 *__restrict__ sym, u64 n, u32 K, u32 lds_bins,
                                                           unsigned long long *__restrict__ counts,
                                                           u32 *__restrict__ bad) {
    extern __shared__ u32 s_bins[];
    for (u32 i = threadIdx.x; i < lds_bins; i += 256) s_bins[i] = 0;
    __syncthreads();
    u32 n_bad = 0;
    // head: symbols in front of the first 16-byte boundary (at most 7), body: whole 16-byte pieces, tail: the rest
    const u64 head = min((u64)(((16u - (u32)((uintptr_t)sym & 15u)) & 15u) >> 1), n);
    const u64 n8 = (n - head) >> 3;
    const uint4 *body = reinterpret_cast<const uint4 *>(sym + head);
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n8; i += (u64)gridDim.x * 256) {
        const uint4 v = body[i];
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            hg16_count(w[d] & 0xFFFFu, K, lds_bins, s_bins, counts, n_bad);
            hg16_count(w[d] >> 16, K, lds_bins, s_bins, counts, n_bad);
        }
    }
    if (blockIdx.x == 0) {
        const u64 tail0 = head + (n8 << 3);
        if (threadIdx.x < head) hg16_count(sym[threadIdx.x], K, lds_bins, s_bins, counts, n_bad);
        if (tail0 + threadIdx.x < n && threadIdx.x < 8) hg16_count(sym[tail0 + threadIdx.x], K, lds_bins, s_bins, counts, n_bad);
    }
    if (n_bad) atomicAdd(bad, n_bad);
    __syncthreads();
    for (u32 i = threadIdx.x; i < lds_bins; i += 256)
        if (s_bins[i]) atomicAdd(&counts[i], (unsigned long long)s_bins[i]);
}

extern "C" int scl_histogram_u16(const uint16_t *d_sym, uint64_t n, uint32_t K, uint64_t *d_counts,
                                 uint32_t *d_out_of_range, void *stream) {
    SCL_REQUIRE(d_counts && d_out_of_range && (d_sym || n == 0), "histogram_u16: null pointer argument");
    SCL_REQUIRE(K >= 1 && K <= SCL_MAX_ALPHABET, "histogram_u16: alphabet size %u outside 1..65536", K);
    SCL_REQUIRE(((uintptr_t)d_sym & 1) == 0, "histogram_u16: d_sym must be 2-byte aligned");
    if (n == 0) return SCL_OK;
    u32 blocks = (u32)((n + 4095) / 4096);  // >= 16 symbols per lane
    if (blocks > 1024) blocks = 1024;
    const u32 lds_bins = K <= 16384 ? K : 0;
    hipLaunchKernelGGL(histogram_u16_kernel, dim3(blocks), dim3(256), lds_bins * sizeof(u32), (hipStream_t)stream, d_sym,
                       n, K, lds_bins, reinterpret_cast<unsigned long long *>(d_counts), d_out_of_range);
    SCL_HIP_TRY(hipGetLastError());
    return SCL_OK;
}

// ---- host convenience drivers -----------------------------------------------------------------------
extern "C" int scl_stream_block_size_host(const uint8_t *h_in, uint64_t in_nbits, uint32_t size_bits,
                                          uint64_t *n_out) {
    SCL_REQUIRE(h_in && n_out && size_bits >= 1 && size_bits <= 32, "block_size: bad arguments");
    SCL_REQUIRE(in_nbits >= size_bits, "block_size: stream shorter than its size header");
    u64 v = 0;
    for (u32 i = 0; i < size_bits; ++i) v = (v << 1) | ((h_in[i >> 3] >> (7 - (i & 7))) & 1u);
    *n_out = v;
    return SCL_OK;
}

static int status_to_error(u32 st, const char *what) {
    if (st == 0) return SCL_OK;
    scl_set_error("%s: chunk status 0x%x%s%s%s%s%s%s", what, st, (st & SCL_ST_CAPACITY) ? " CAPACITY" : "",
                  (st & SCL_ST_SYMBOL) ? " SYMBOL" : "", (st & SCL_ST_TRUNCATED) ? " TRUNCATED" : "",
                  (st & SCL_ST_STATE) ? " STATE" : "", (st & SCL_ST_TOTAL) ? " TOTAL" : "",
                  (st & SCL_ST_SIZE) ? " SIZE" : "");
    return SCL_E_CHUNK;
}

int scl_host_encode_one(const HostEncodeCall &call, const void *model, const u8 *h_sym, u64 n, u8 *h_out,
                        u64 out_cap_bytes, u64 *nbits) {
    SCL_REQUIRE(model && h_out && nbits && (h_sym || n == 0), "encode_host: null argument");
    SCL_REQUIRE(n < (1ull << 32), "encode_host: block too large");
    const u64 slot = call.slot_bytes(model, n);
    const u64 scratch_bytes = call.scratch_bytes ? call.scratch_bytes(model) : 0;
    ScratchDev d_sym, d_slot, d_meta, d_dense, d_scr, d_cscr;
    int rc;
    if ((rc = d_sym.alloc(n * call.sym_bytes + 16)) || (rc = d_slot.alloc(slot)) || (rc = d_meta.alloc(64)) ||
        (rc = d_dense.alloc(slot + 16)) || (rc = d_scr.alloc(scratch_bytes)) ||
        (rc = d_cscr.alloc(scl_streams_compact_scratch_bytes(1))))
        return rc;
    u64 *d_bit_off = (u64 *)d_meta.p;           // [0]
    u64 *d_rec_off = (u64 *)d_meta.p + 1;       // [1..2]
    u32 *d_nbits = (u32 *)((u64 *)d_meta.p + 4);  // byte 32
    u32 *d_status = d_nbits + 1;
    if (n) SCL_HIP_TRY(hipMemcpy(d_sym.p, h_sym, n * call.sym_bytes, hipMemcpyHostToDevice));
    SCL_HIP_TRY(hipMemset(d_meta.p, 0, 64));
    if (call.pre && (rc = call.pre(model, d_scr.p, call.user))) return rc;
    rc = call.run(model, (const u8 *)d_sym.p, (u32)n, (u8 *)d_slot.p, slot, d_bit_off, d_nbits, d_status, d_scr.p,
                  scratch_bytes);
    if (rc) return rc;
    rc = scl_streams_compact((const u8 *)d_slot.p, d_bit_off, d_nbits, 1, SCL_COMPACT_DENSE, (u8 *)d_dense.p, slot + 16,
                             d_rec_off, d_cscr.p, nullptr);
    if (rc) return rc;
    SCL_HIP_TRY(hipDeviceSynchronize());
    u32 meta[2];
    SCL_HIP_TRY(hipMemcpy(meta, d_nbits, 8, hipMemcpyDeviceToHost));
    if ((rc = status_to_error(meta[1], "encode_host"))) return rc;
    const u64 bytes = ((u64)meta[0] + 7) / 8;
    if (bytes > out_cap_bytes) {
        scl_set_error("encode_host: output needs %llu bytes, capacity %llu", (unsigned long long)bytes,
                      (unsigned long long)out_cap_bytes);
        return SCL_E_PARAM;
    }
    SCL_HIP_TRY(hipMemcpy(h_out, d_dense.p, bytes, hipMemcpyDeviceToHost));
    *nbits = meta[0];
    if (call.post && (rc = call.post(model, d_scr.p, call.user))) return rc;
    return SCL_OK;
}

int scl_host_decode_one(const HostDecodeCall &call, const void *model, const u8 *h_in, u64 in_nbits, u8 *h_out_sym,
                        u64 out_cap, u64 *n_out, u64 *consumed) {
    SCL_REQUIRE(model && h_in && n_out && consumed && (h_out_sym || out_cap == 0), "decode_host: null argument");
    SCL_REQUIRE(in_nbits < (1ull << 32) && out_cap < (1ull << 32), "decode_host: stream too large");
    const u64 in_bytes = (in_nbits + 7) / 8;
    const u64 scratch_bytes = call.scratch_bytes ? call.scratch_bytes(model) : 0;
    ScratchDev d_in, d_out, d_meta, d_scr;
    int rc;
    if ((rc = d_in.alloc(in_bytes + 32)) || (rc = d_out.alloc((out_cap + 16) * call.sym_bytes)) || (rc = d_meta.alloc(64)) ||
        (rc = d_scr.alloc(scratch_bytes)))
        return rc;
    SCL_HIP_TRY(hipMemset(d_in.p, 0, in_bytes + 32));
    SCL_HIP_TRY(hipMemcpy(d_in.p, h_in, in_bytes, hipMemcpyHostToDevice));
    u64 h_bit_off = 0;
    u32 h_nb = (u32)in_nbits;
    u64 *d_bit_off = (u64 *)d_meta.p;
    u32 *d_nb = (u32 *)((u64 *)d_meta.p + 1);
    u32 *d_len = d_nb + 1, *d_used = d_nb + 2, *d_status = d_nb + 3;
    SCL_HIP_TRY(hipMemset(d_meta.p, 0, 64));
    SCL_HIP_TRY(hipMemcpy(d_bit_off, &h_bit_off, 8, hipMemcpyHostToDevice));
    SCL_HIP_TRY(hipMemcpy(d_nb, &h_nb, 4, hipMemcpyHostToDevice));
    if (call.pre && (rc = call.pre(model, d_scr.p, call.user))) return rc;
    rc = call.run(model, (const u8 *)d_in.p, in_bytes + 32, d_bit_off, d_nb, (u8 *)d_out.p, (u32)out_cap, d_len, d_used,
                  d_status, d_scr.p, scratch_bytes);
    if (rc) return rc;
    SCL_HIP_TRY(hipDeviceSynchronize());
    u32 meta[3];
    SCL_HIP_TRY(hipMemcpy(meta, d_len, 12, hipMemcpyDeviceToHost));
    *n_out = meta[0];
    *consumed = meta[1];
    if ((rc = status_to_error(meta[2], "decode_host"))) return rc;
    if (meta[0]) SCL_HIP_TRY(hipMemcpy(h_out_sym, d_out.p, (u64)meta[0] * call.sym_bytes, hipMemcpyDeviceToHost));
    if (call.post && (rc = call.post(model, d_scr.p, call.user))) return rc;
    return SCL_OK;
}
