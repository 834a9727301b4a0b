// scl_range_fast.hip -- gfx950 fast path of the batched 32-bit range coder (BASELINE.json configs[2]).
//
// Same byte stream as scl_range.hip / reference scl/compressors/range_coder.py:188-207 (encode) and
// :269-317 (decode).  Serves PRECISION = 32, DATA_BLOCK_SIZE_BITS = 32, any total_freq the coder allows (<= 2^16); every other
// parameter set runs the generic kernels.  Memory access follows the rule established for rANS
// (scl_rans_fast.hip, profiles/r01_v2 -> r01_v3): a lane only ever moves whole lines / 64-byte sectors.
//   encode: 128-byte input lines in registers (next line prefetched); emitted bytes are gathered per symbol
//           (a symbol releases 0..4 bytes), appended to a 64-bit byte accumulator, completed words parked in a
//           per-lane LDS ring ([word][thread], conflict-free) and stored 64 bytes at a time, front to back.
//   decode: stream read through the LDS word ring of scl_rans_fast.hip's decoder; the symbol search
//           max{s : low + c[s]*r <= state} (:232-237) is evaluated as q = (state - low) / r (exact, in binary64
//           with a Newton-refined v_rcp_f32), then ONE read of a slot -> {c, f, s} table; 128 symbols per store burst.
#include <vector>

#include <type_traits>

#include "scl_ans_fast_io.h"  // scl_transpose8: the 8 x 8 register transpose behind the cooperative line store
#include "scl_range_internal.h"

#define RGE_THREADS 256
#define RGE_RING_BYTES (32 * RGE_THREADS * 4)
#define RGD_THREADS 1024
#define RGD_RING_BYTES (32 * RGD_THREADS * 4)
#define RG_TOP (1u << 24)
#define RG_BOTTOM (1u << 16)

// ---------------------------------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------------------------------
struct RgOut {
    static constexpr bool STRIPED = false;
    u64 acc;   // pending bytes as a big-endian number in its low cnt bits (the next byte of the stream is the most significant
               // of them; what lies above bit cnt is left-over and never read) -- round 5: the bytes a symbol releases are then
               // simply the high word of (u64)low << sh, no byte swap and no field extraction per symbol
    u32 cnt;   // number of pending BITS (a multiple of 8), < 32 between symbols -- bits, not bytes: every count on the
               // per-symbol path is a shift amount, and 8 * bytes was an instruction each time (round 4)
    u32 ra;    // LDS byte address of the ring word written next
    u32 fa;    // LDS byte address of the oldest unflushed word
    u32 pend;  // completed words not yet in memory
    u32 nfl;   // words already in memory
    u32 overflow;
    u64 cap;   // slot capacity in bytes
    u8 *slot;
    uint4 held[4];  // first 64-byte half of the current line (stored together with the second half)
    u32 have_held;
    // Cooperative line store (round 5): a lane storing its own 128-byte line makes every store instruction touch 64 different
    // lines, 16 bytes of each -- the write shape that cost the rANS encoder a quarter of its time in round 2, and
    // this encoder 0.2 of its 0.82 ms (RG_ABLATE_NOSTORE).  When the whole wave reaches a flush point with a complete line
    // at the same position of its slot -- lanes of a batch of equally long chunks whose streams grow at the same rate
    // (uniform bytes: configs[2]) do so at nearly every line -- the 64 lines are transposed in registers across the lanes
    // l, l + 8, ..., l + 56 (scl_transpose8) and every store instruction writes eight WHOLE lines, eight lanes per line.
    // Anything else (ragged batches, partial waves, lanes out of step) keeps the lane's own stores.
    bool coop;      // wave-uniform: all 64 lanes code chunks of one length
    u8 *coop_base;  // lane (l0, k) = (lane & 7, lane >> 3): piece k of the lines of the lanes l0 + 8 j
    u64 stride;

    __device__ __forceinline__ void init(u32 tid, u8 *slot_, u64 cap_, bool coop_ = false) {
        const u32 lane = tid & 63u;
        coop = coop_;
        stride = cap_;  // slots are out_stride apart and out_stride long
        coop_base = slot_ - (u64)lane * cap_ + (u64)(lane & 7u) * cap_ + 16u * (lane >> 3);
        overflow = 0;
        cap = cap_;
        have_held = 0;
        held[0] = held[1] = held[2] = held[3] = make_uint4(0, 0, 0, 0);
        acc = 0;
        cnt = 0;
        ra = fa = tid * 4;
        pend = 0;
        nfl = 0;
        slot = slot_;
    }
    // append nbits / 8 (0..4) bytes: `bytes` is their big-endian number (the first byte in stream order the most significant)
    __device__ __forceinline__ void put_bytes(char *lds, u32 bytes, u32 nbits) {
        acc = (acc << nbits) | bytes;
        cnt += nbits;
        if (cnt >= 32) {
            cnt -= 32;
            *reinterpret_cast<u32 *>(lds + ra) = __builtin_bswap32((u32)(acc >> cnt));  // v_alignbit + v_perm
            ra = (ra + RGE_THREADS * 4) & (RGE_RING_BYTES - 1);
            ++pend;
        }
    }
    // 16 pending words -> 64 contiguous bytes; call at least every 16 symbols (<= 16 new words, ring of 32)
    // COOP: compiled in for the table-free modes only -- the modes with a table sit at their 128 registers (four waves per
    // SIMD) and spill 52-68 bytes with the transpose
    template <bool COOP = false>
    __device__ __forceinline__ void maybe_flush(char *lds) {
        if (COOP && coop) {  // (wave-uniform; every lane of the wave makes this call)
            const bool full = pend >= 16 && have_held;
            if (__builtin_amdgcn_ballot_w64(full && nfl == (u32)__builtin_amdgcn_readfirstlane((int)nfl)) == ~0ull) {
                const char *r = lds + fa;
                u32 w[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = *reinterpret_cast<const u32 *>(r + j * RGE_THREADS * 4);
                uint4 a[8] = {held[0], held[1], held[2], held[3], make_uint4(w[0], w[1], w[2], w[3]),
                              make_uint4(w[4], w[5], w[6], w[7]), make_uint4(w[8], w[9], w[10], w[11]),
                              make_uint4(w[12], w[13], w[14], w[15])};
                if (4 * (u64)nfl + 64 <= cap) {  // the same for all lanes
                    scl_transpose8(a);
                    u8 *p = coop_base + 4 * (u64)(nfl - 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        *reinterpret_cast<uint4 *>(p + (u64)(8 * j) * stride) = a[j];
                    }
                } else {
                    overflow = 1;
                }
                have_held = 0;
                nfl += 16;
                pend -= 16;
                fa ^= 16 * RGE_THREADS * 4;
                return;
            }
        }
        if (pend >= 16) {
            const char *r = lds + fa;
            u32 w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = *reinterpret_cast<const u32 *>(r + j * RGE_THREADS * 4);
            const uint4 q0 = make_uint4(w[0], w[1], w[2], w[3]), q1 = make_uint4(w[4], w[5], w[6], w[7]);
            const uint4 q2 = make_uint4(w[8], w[9], w[10], w[11]), q3 = make_uint4(w[12], w[13], w[14], w[15]);
            if (have_held) {  // second half of a line: store the whole 128 bytes at once
                if (4 * (u64)nfl + 64 <= cap) {
                    uint4 *p = reinterpret_cast<uint4 *>(slot + 4 * (u64)(nfl - 16));
                    p[0] = held[0];
                    p[1] = held[1];
                    p[2] = held[2];
                    p[3] = held[3];
                    p[4] = q0;
                    p[5] = q1;
                    p[6] = q2;
                    p[7] = q3;
                } else {
                    overflow = 1;
                }
                have_held = 0;
            } else {
                held[0] = q0;
                held[1] = q1;
                held[2] = q2;
                held[3] = q3;
                have_held = 1;
            }
            nfl += 16;
            pend -= 16;
            fa ^= 16 * RGE_THREADS * 4;
        }
    }
    __device__ __forceinline__ u64 finish(char *lds) {  // returns total bytes
        maybe_flush(lds);
        u32 *w32 = reinterpret_cast<u32 *>(slot);
        const u64 words = (u64)nfl + pend;
        const u32 cnt_bytes = cnt >> 3;
        if (words * 4 + cnt_bytes > cap) {
            overflow = 1;
            return words * 4 + cnt_bytes;
        }
        if (have_held) {  // counted in nfl already: bytes [4*(nfl-16), 4*nfl)
            uint4 *p = reinterpret_cast<uint4 *>(slot + 4 * (u64)(nfl - 16));
            p[0] = held[0];
            p[1] = held[1];
            p[2] = held[2];
            p[3] = held[3];
        }
        u32 a = fa;
        for (u32 j = 0; j < pend; ++j) {
            w32[nfl + j] = *reinterpret_cast<const u32 *>(lds + a);
            a = (a + RGE_THREADS * 4) & (RGE_RING_BYTES - 1);
        }
        for (u32 j = 0; j < cnt_bytes; ++j) slot[words * 4 + j] = (u8)(acc >> (cnt - 8 * (j + 1)));
        return words * 4 + cnt_bytes;
    }
};

// Round 6: the same byte accumulator over WAVE-STRIPED slots (scl_ans_fast_io.h: byte b of the logical slot of lane l lives
// at wave_slot + (b / 16) * 1024 + 16 l + b % 16, so piece q of the 64 lanes of a wave is one contiguous kilobyte).  What the
// cooperative line store above buys only for lanes in lockstep (uniform bytes), this buys for any table: a lane keeps
// 16-byte pieces, not 128-byte lines, and the WAVE stores the next row when every lane present holds a complete piece (four
// nested ballots per flush point; AnsBackWriterT has the argument) -- 64 adjacent pieces per store instruction, no transpose,
// no 64-byte half line held in registers.  A lane holding more than 15 words when it leaves a flush point (32 ring words - the
// 16 that 16 symbols can complete) stores its own oldest pieces.  Streams grow front to back: rows ascend.
struct RgOutT {
    static constexpr bool STRIPED = true;
    static constexpr u32 ROW = RGE_THREADS * 4;
    u64 acc;   // as RgOut
    u32 cnt;
    u32 ra;    // LDS byte address of the ring word written next
    u32 fa;    // LDS byte address of the oldest unflushed word (a multiple of four rows | the thread column)
    u32 pend;  // completed words not yet in memory
    u32 goff;  // byte offset (from the workgroup's output base) of this lane's next piece
    u32 goff0, glimit;  // ... of its first piece / the first offset past its last one (the slot's capacity)
    u32 overflow;
    u8 *wg_out;

    __device__ __forceinline__ void init(u32 tid, u8 *wg_out_, u32 out_stride) {
        wg_out = wg_out_;
        acc = 0;
        cnt = 0;
        ra = fa = tid * 4;
        pend = 0;
        overflow = 0;
        goff = goff0 = (tid >> 6) * (out_stride << 6) + ((tid & 63u) << 4);
        glimit = goff0 + (out_stride >> 4) * 1024u;
    }
    __device__ __forceinline__ void put_bytes(char *lds, u32 bytes, u32 nbits) {
        acc = (acc << nbits) | bytes;
        cnt += nbits;
        if (cnt >= 32) {
            cnt -= 32;
            *reinterpret_cast<u32 *>(lds + ra) = __builtin_bswap32((u32)(acc >> cnt));
            ra = (ra + ROW) & (RGE_RING_BYTES - 1);
            ++pend;
        }
    }
    // piece R (0 = oldest) of the complete ones -> its row; nothing is updated (rows_done)
    template <u32 R>
    __device__ __forceinline__ void store_row(char *lds) {
        const char *r = lds + ((fa + 4u * R * ROW) & (RGE_RING_BYTES - 1));  // (a piece never wraps: four rows, fa a multiple of four)
        const u32 w0 = *reinterpret_cast<const u32 *>(r), w1 = *reinterpret_cast<const u32 *>(r + ROW);
        const u32 w2 = *reinterpret_cast<const u32 *>(r + 2 * ROW), w3 = *reinterpret_cast<const u32 *>(r + 3 * ROW);
        if (goff + 1024u * R < glimit) {
            typedef u32 u32x4_nt __attribute__((ext_vector_type(4)));
            const u32x4_nt t = {w0, w1, w2, w3};
            __builtin_nontemporal_store(t, reinterpret_cast<u32x4_nt *>(wg_out + goff + 1024u * R));
        } else {
            overflow = 1;
        }
    }
    __device__ __forceinline__ void rows_done(u32 m) {
        goff += 1024u * m;
        fa = (fa + 4u * m * ROW) & (RGE_RING_BYTES - 1);
        pend -= 4u * m;
    }
    // call at least every 16 symbols (<= 16 new words)
    template <bool COOP_UNUSED = false>
    __device__ __forceinline__ void maybe_flush(char *lds) {
        const u32 p = pend;
        const u64 all = __builtin_amdgcn_ballot_w64(true);
        if (__builtin_amdgcn_ballot_w64(p >= 4u) == all) {
            store_row<0>(lds);
            if (__builtin_amdgcn_ballot_w64(p >= 8u) == all) {
                store_row<1>(lds);
                if (__builtin_amdgcn_ballot_w64(p >= 12u) == all) {
                    store_row<2>(lds);
                    if (__builtin_amdgcn_ballot_w64(p >= 16u) == all) {
                        store_row<3>(lds);
                        rows_done(4);
                    } else {
                        rows_done(3);
                    }
                } else {
                    rows_done(2);
                }
            } else {
                rows_done(1);
            }
        }
        while (pend > 15u) {
            store_row<0>(lds);
            rows_done(1);
        }
    }
    __device__ __forceinline__ u64 finish(char *lds) {  // returns total bytes
        while (pend >= 4u) {
            store_row<0>(lds);
            rows_done(1);
        }
        const u32 cnt_bytes = cnt >> 3;
        const u64 total = (u64)((goff - goff0) >> 10) * 16u + pend * 4u + cnt_bytes;
        if (goff >= glimit && (pend || cnt_bytes)) overflow = 1;
        if (overflow) return total;
        u8 *piece = wg_out + goff;  // the piece that is still filling: pend < 4 words and < 4 bytes, i.e. at most 15 bytes
        u32 a = fa;
        for (u32 j = 0; j < pend; ++j) {
            *reinterpret_cast<u32 *>(piece + 4 * j) = *reinterpret_cast<const u32 *>(lds + a);
            a = (a + ROW) & (RGE_RING_BYTES - 1);
        }
        for (u32 j = 0; j < cnt_bytes; ++j) piece[4 * pend + j] = (u8)(acc >> (cnt - 8 * (j + 1)));
        return total;
    }
};

// range // M (shrink_range, :100): a shift for a power-of-two total (GEN = false, md.m_log2), else exactly as
// trunc((range + 0.5) * (1 / M)) in binary64 (range < 2^32, M <= 2^12: the error 2^-20 is far below the distance
// 0.5 / M of (range + 0.5) / M from an integer)
struct RgDivM {
    u32 m_log2;
    u32 k;
    u32 t;  // MODE 2: log2 of the common frequency
    double inv_m;
};
// MODE: 0 = power-of-two total, 1 = any total, 2 = power-of-two total shared evenly by 256 symbols (f = 2^t, c = s f:
// no table at all), 3 = the same with t = 0 as a compile-time fact (f = 1, M = 256: configs[2], uniform bytes -- two
// shifts by zero less per decoded symbol, one per encoded symbol)
#define RG_UNI(MODE) ((MODE) == 2 || (MODE) == 3)
// encoder modes 4 and 5 = 0 and 1 for totals >= 256: range // M < 2^24, so c r and r f are 24-bit multiplies (full rate)
// instead of v_mad_u64_u32 + v_mul_lo_u32 (quarter rate each)
#define RG_R24(MODE) ((MODE) == 4 || (MODE) == 5)
template <int MODE>
__device__ __forceinline__ u32 rg_range_over_m(u32 range, const RgDivM &md) {
    if (MODE == 1 || MODE == 5) return (u32)(((double)range + 0.5) * md.inv_m);
    if (MODE == 3) return range >> 8;  // a literal shift: half the cost of one whose amount sits in an SGPR
    return range >> md.m_log2;
}

__device__ __forceinline__ bool rg_needs_byte(u32 low, u32 &range) {
    const bool settled = ((low ^ (low + range)) < RG_TOP);
    if (!settled && range >= RG_BOTTOM) return false;
    if (!settled) range = (0u - low) & (RG_BOTTOM - 1);  // (MASK + 1 - low) & (BOTTOM - 1), :172
    return true;
}

// The common case of one symbol and nothing else (round 5): shrink_range, then the leading bytes low and low + range
// agree on.  No branch, no loop; returns the range left after them -- below BOTTOM means the carry-less reset is due and
// none of what this returned holds (the caller replays the symbol with rg_encode_symbol).  May be called on a state that is
// itself such a discarded result (the second symbol of a pair whose first one was rare): everything here is defined for
// any input (v_ffbh_u32 of 0 is -1, written as the instruction: __builtin_clz(0) is not).
template <int MODE>
__device__ __forceinline__ u32 rg_fast_symbol(u32 &low, u32 &range, const uint2 e, const RgDivM &md, u32 &bytes, u32 &nb) {
    const u32 r = rg_range_over_m<MODE>(range, md);
    u32 low0, range0;
    if (RG_UNI(MODE)) {
        range0 = (MODE == 3) ? r : (r << md.t);
        low0 = __umul24(e.x, range0) + low;
    } else if (RG_R24(MODE)) {
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(low0) : "v"(e.x), "v"(r), "v"(low));
        range0 = __umul24(r, e.y);
    } else {
        low0 = low + e.x * r;
        range0 = r * e.y;
    }
    u32 lz;
    asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(low0 ^ (low0 + range0)));
    const u32 sh = lz & 0x18u;
    const u64 l64 = ((u64)low0) << sh;
    bytes = (u32)(l64 >> 32);
    nb = sh;
    low = (u32)l64;
    range = range0 << sh;
    return range;
}

// shrink_range (:88-105) + normalize (:107-179) for one symbol; returns its released bytes (0..3 of them, as a big-endian
// number) and their number IN BITS.  The normalisation is a closed form: the loop first releases every leading byte on which low and
// low + range agree (low + range never carries out of 32 bits), nb1 = clz(low ^ (low + range)) / 8 of them (range > 0, so
// the two values differ and nb1 <= 3); it goes on only if the range left after that is below BOTTOM (the carry-less
// reset, :136-178) -- rare.  ONE branch per symbol covers everything rare: the reference's literal loop, a symbol that
// releases a whole word (the bytes (pb, pn) its pair partner still holds go out first, to keep the reference's order of
// events) and the bytes after a fourth.  Fast-path results are computed unconditionally and overwritten there (branches
// are what this kernel has too many of: each costs 0.4-1.3 ns per wave, tools/ubench/valu_rate.hip).
template <int MODE, typename OUT>
__device__ __forceinline__ void rg_encode_symbol(u32 &low, u32 &range, const uint2 e, const RgDivM &md, u32 &bytes,
                                                 u32 &nb, u32 &pb, u32 &pn, OUT &o, char *lds) {
    const u32 r = rg_range_over_m<MODE>(range, md);
    u32 low0, range0;
    if (RG_UNI(MODE)) {  // e.x = the symbol: c r = s (r f), r f < 2^24 (range < 2^32, M / f = 256): one 24-bit multiply-add
        range0 = (MODE == 3) ? r : (r << md.t);
        low0 = __umul24(e.x, range0) + low;
    } else if (RG_R24(MODE)) {
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(low0) : "v"(e.x), "v"(r), "v"(low));  // c, r < 2^24; c r <= range
        range0 = __umul24(r, e.y);
    } else {
        low0 = low + e.x * r;  // c * r <= range: no overflow past MASK (carry-less coder)
        range0 = r * e.y;
    }
    // 8 * (leading common bytes): clz & 0x18 (the two values differ, so clz <= 31 and the count <= 3 bytes)
    const u32 sh = (u32)__builtin_clz(low0 ^ (low0 + range0)) & 0x18u;
    const u32 range_s = range0 << sh;
    const u64 l64 = ((u64)low0) << sh;  // one v_lshlrev_b64: the released bytes (big-endian) : the new low
    bytes = (u32)(l64 >> 32);
    nb = sh;  // in BITS
    low = (u32)l64;
    range = range_s;
    if (__builtin_expect(range_s < RG_BOTTOM, 0)) {
        low = low0;
        range = range0;
        bytes = 0;
        nb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool settled = ((low ^ (low + range)) < RG_TOP);
            if (!settled && range >= RG_BOTTOM) break;
            if (!settled) range = (0u - low) & (RG_BOTTOM - 1);  // (MASK + 1 - low) & (BOTTOM - 1), :172
            bytes = (bytes << 8) | (low >> 24);
            nb += 8;
            low <<= 8;
            range <<= 8;
        }
        if (nb == 32) {  // a whole word at once: everything goes out now, in the reference's order
            o.put_bytes(lds, pb, pn);
            pb = 0;
            pn = 0;
            o.put_bytes(lds, bytes, 32);
            bytes = 0;
            nb = 0;
            while (rg_needs_byte(low, range)) {  // never taken for valid models; keeps the loop exact
                o.put_bytes(lds, low >> 24, 8);
                low <<= 8;
                range <<= 8;
            }
        }
    }
}

struct RgLine128 {
    uint4 v[8];
    __device__ __forceinline__ void load(const uint4 *p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[i];
    }
};

// table entry {c, f} of the symbol whose byte offset into the table is a (= symbol * 8); MODE 2 has no table
template <int MODE>
__device__ __forceinline__ uint2 rg_entry(const char *tab, u32 a) {
    if (RG_UNI(MODE)) return make_uint2(a, 0u);  // a IS the symbol here (rg_encode16)
    return *reinterpret_cast<const uint2 *>(tab + a);
}

template <int MODE, typename OUT>
__device__ __forceinline__ void rg_encode16(const uint4 v, u32 &low, u32 &range, OUT &o, u32 &bad, char *lds,
                                            const char *tab, const RgDivM &md) {
    const u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const u32 w = wv[d];
        // table offsets 8 * symbol; the table-free modes (256 symbols: nothing to check) take the symbol itself, one v_bfe
        const u32 a[4] = {RG_UNI(MODE) ? __builtin_amdgcn_ubfe(w, 0, 8) : (w << 3) & 0x7F8u,
                          RG_UNI(MODE) ? __builtin_amdgcn_ubfe(w, 8, 8) : (w >> 5) & 0x7F8u,
                          RG_UNI(MODE) ? __builtin_amdgcn_ubfe(w, 16, 8) : (w >> 13) & 0x7F8u,
                          RG_UNI(MODE) ? (w >> 24) : (w >> 21) & 0x7F8u};
        // two symbols per visit of the byte accumulator: their released bytes (0..3 each in the common case) are
        // merged first when they fit a word together -- "a word completed" is then tested, and its branch body
        // executed by the whole wave, once per pair instead of once per symbol
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            if (!RG_UNI(MODE)) bad = max(bad, max(a[j], a[j + 1]));
            u32 b0, n0, b1, n1;  // n0, n1: bits
            // Both symbols on the common path, unconditionally; ONE test and one branch per pair for everything else -- a
            // carry-less reset in either symbol, more than a word of bytes from the two together -- whose lanes replay the
            // pair from the state of before it, symbol by symbol, with the exact routine.  (Was: a branch per symbol, a
            // third for the byte count, 20 scalar instructions and three taken branches per pair.)
            const uint2 e0 = rg_entry<MODE>(tab, a[j]), e1 = rg_entry<MODE>(tab, a[j + 1]);
            const u32 low_s = low, range_s = range;
            const u32 r0 = rg_fast_symbol<MODE>(low, range, e0, md, b0, n0);
            const u32 r1 = rg_fast_symbol<MODE>(low, range, e1, md, b1, n1);
            const u32 nn = n0 + n1;
            if (__builtin_expect((min(r0, r1) < RG_BOTTOM) | (nn > 32), 0)) {
                low = low_s;
                range = range_s;
                u32 z0 = 0, zn = 0, z1 = 0, z1n = 0;
                rg_encode_symbol<MODE>(low, range, e0, md, b0, n0, z0, zn, o, lds);
                o.put_bytes(lds, b0, n0);
                rg_encode_symbol<MODE>(low, range, e1, md, b1, n1, z1, z1n, o, lds);
                o.put_bytes(lds, b1, n1);
            } else {
                o.put_bytes(lds, (b0 << n1) | b1, nn);
            }
        }
    }
    o.template maybe_flush<RG_UNI(MODE)>(lds);
}

// OUT: RgOut (linear slots) or RgOutT (wave-striped slots, ABI 8: `out` then holds round_up(n_chunks, 64) slots)
template <int MODE, typename OUT = RgOut>
__global__ void __launch_bounds__(RGE_THREADS, 4) range_encode_fast_kernel(RangeFastDev P, const u8 *__restrict__ sym,
                                                                          u64 sym_stride,
                                                                          const u32 *__restrict__ lens, u32 chunk_len,
                                                                          u64 n_chunks, u8 *__restrict__ out,
                                                                          u64 out_stride, u64 *__restrict__ out_bit_off,
                                                                          u32 *__restrict__ out_nbits,
                                                                          u32 *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) char s_lds[RGE_RING_BYTES + 256 * 8];
    char *lds = s_lds;
    const char *tab = s_lds + RGE_RING_BYTES;
    reinterpret_cast<uint2 *>(s_lds + RGE_RING_BYTES)[threadIdx.x & 255] = P.d_enc_tab[threadIdx.x & 255];
    __syncthreads();
    const u64 c = (u64)blockIdx.x * RGE_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 n = lens ? lens[c] : chunk_len;
    const u8 *src = sym + c * sym_stride;
    OUT o;
    if constexpr (OUT::STRIPED) {
        o.init(threadIdx.x, out + (u64)blockIdx.x * RGE_THREADS * out_stride, (u32)out_stride);
    } else {
        // whole wave, equally long chunks: every lane reaches every flush point, so the wave can store lines cooperatively
        const bool coop = __builtin_amdgcn_ballot_w64(n == (u32)__builtin_amdgcn_readfirstlane((int)n)) == ~0ull;
        o.init(threadIdx.x, out + c * out_stride, out_stride, coop);
    }
    o.put_bytes(lds, n, 32);  // DATA_BLOCK_SIZE_BITS = 32 header, :197
    u32 low = 0, range = 0xFFFFFFFFu, bad = 0;
    RgDivM md;
    md.m_log2 = P.m_log2;
    md.k = P.K;
    md.t = P.uni_t;
    md.inv_m = 1.0 / (double)P.M;

    const u32 n_lines = n >> 7;
    const uint4 *src16 = reinterpret_cast<const uint4 *>(src);
    RgLine128 cur;  // no second buffer: with one the kernel spilled 100 bytes per lane (4 waves per SIMD hide the load)
#pragma nounroll
    for (u32 t = 0; t < n_lines; ++t) {
        cur.load(src16 + 8 * t);
        if (RG_UNI(MODE)) {
            // half a line per iteration (round 4, the table-free modes: the two with a table spill in this form): the
            // two-block form moves the line down by two registers after every 32 symbols (24 copies, 0.75 vector
            // instructions per symbol); this one moves four registers once per line
#pragma nounroll
            for (int q = 0; q < 2; ++q) {
                rg_encode16<MODE>(cur.v[0], low, range, o, bad, lds, tab, md);
                rg_encode16<MODE>(cur.v[1], low, range, o, bad, lds, tab, md);
                rg_encode16<MODE>(cur.v[2], low, range, o, bad, lds, tab, md);
                rg_encode16<MODE>(cur.v[3], low, range, o, bad, lds, tab, md);
#pragma unroll
                for (int i = 0; i < 4; ++i) cur.v[i] = cur.v[i + 4];
            }
        } else {
#pragma nounroll
            for (int q = 0; q < 4; ++q) {
                rg_encode16<MODE>(cur.v[0], low, range, o, bad, lds, tab, md);
                rg_encode16<MODE>(cur.v[1], low, range, o, bad, lds, tab, md);
#pragma unroll
                for (int i = 0; i < 6; ++i) cur.v[i] = cur.v[i + 2];
            }
        }
    }
    u32 i = n_lines << 7;
    for (; i + 16 <= n; i += 16)
        rg_encode16<MODE>(*reinterpret_cast<const uint4 *>(src + i), low, range, o, bad, lds, tab, md);
    for (; i < n; ++i) {
        const u32 a = RG_UNI(MODE) ? (u32)src[i] : (u32)src[i] << 3;
        if (!RG_UNI(MODE)) bad = max(bad, a);
        u32 bytes, nb, z0 = 0, zn = 0;
        rg_encode_symbol<MODE>(low, range, rg_entry<MODE>(tab, a), md, bytes, nb, z0, zn, o, lds);
        o.put_bytes(lds, bytes, nb);
        if ((i & 15u) == 15u) o.template maybe_flush<RG_UNI(MODE)>(lds);
    }
    o.template maybe_flush<RG_UNI(MODE)>(lds);
    o.put_bytes(lds, low, 32);  // flush :181-186: the four bytes of low, most significant first
    const u64 total_bytes = o.finish(lds);
    out_bit_off[c] = c * out_stride * 8;
    out_nbits[c] = (u32)(total_bytes * 8);
    if (status) status[c] = ((bad >= (P.K << 3)) ? SCL_ST_SYMBOL : 0u) | (o.overflow ? SCL_ST_CAPACITY : 0u);
}

// ---------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------
struct RgIn {  // forward bit reader over a per-lane LDS word ring (same scheme as the rANS fast decoder)
    const uint4 *base;
    u64 n_blocks16, next64;
    uint4 pf[4];
    u32 ra, wa, nrd, nwr;
    u32 A, B;
    int sh;
    u32 bias;

    __device__ __forceinline__ void load64(u64 j) {
        // one bounds test for the whole 64-byte block (all but the buffer's last are readable whole): the four separately
        // tested loads were 36 instructions per refill, run by the whole wave whenever any lane refills
        if (j * 4 + 4 <= n_blocks16) {
            const uint4 *p = base + j * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = p[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u64 idx = j * 4 + i;
                pf[i] = (idx < n_blocks16) ? base[idx] : make_uint4(0, 0, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void push_pf(char *lds) {
        char *r = lds + wa;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32 *>(r + (4 * i + 0) * RGD_THREADS * 4) = __builtin_bswap32(pf[i].x);
            *reinterpret_cast<u32 *>(r + (4 * i + 1) * RGD_THREADS * 4) = __builtin_bswap32(pf[i].y);
            *reinterpret_cast<u32 *>(r + (4 * i + 2) * RGD_THREADS * 4) = __builtin_bswap32(pf[i].z);
            *reinterpret_cast<u32 *>(r + (4 * i + 3) * RGD_THREADS * 4) = __builtin_bswap32(pf[i].w);
        }
        wa ^= 16 * RGD_THREADS * 4;
        nwr += 16;
    }
    __device__ __forceinline__ u32 next_word(const char *lds) {
        const u32 v = *reinterpret_cast<const u32 *>(lds + ra);
        ra = (ra + RGD_THREADS * 4) & (RGD_RING_BYTES - 1);
        ++nrd;
        return v;
    }
    // call at least every 4 symbols (a symbol consumes <= 4 bytes: 4 words in between)
    __device__ __forceinline__ void maybe_refill(char *lds) {
        if (nwr - nrd <= 16) {
            push_pf(lds);
            load64(next64++);
        }
    }
    __device__ __forceinline__ void init(const u8 *in, u64 in_size_bytes, u64 bit_off, char *lds, u32 tid) {
        base = reinterpret_cast<const uint4 *>(in);
        n_blocks16 = in_size_bytes >> 4;
        const u64 j0 = bit_off >> 9;
        wa = tid * 4;
        nwr = 0;
        load64(j0);
        push_pf(lds);
        load64(j0 + 1);
        push_pf(lds);
        load64(j0 + 2);
        next64 = j0 + 3;
        const u32 w0 = (u32)(bit_off >> 5) & 15u;
        ra = tid * 4 + w0 * RGD_THREADS * 4;
        nrd = w0;
        const u32 pos = (u32)bit_off & 31u;
        const u32 first = next_word(lds);
        if (pos == 0) {
            A = 0;
            B = first;
            sh = 0;
        } else {
            A = first;
            B = next_word(lds);
            sh = 32 - (int)pos;
        }
        bias = 32 * nrd - (u32)sh;
    }
    __device__ __forceinline__ u32 consumed() const { return 32 * nrd - (u32)sh - bias; }
    __device__ __forceinline__ u32 look() const { return __builtin_amdgcn_alignbit(A, B, (u32)sh); }
    __device__ __forceinline__ u32 look(const char *) const { return look(); }
    __device__ __forceinline__ void advance(const char *lds, u32 nb) {  // nb <= 32
        sh -= (int)nb;
        if (sh < 0) {
            A = B;
            B = next_word(lds);
            sh += 32;
        }
    }
    __device__ __forceinline__ u32 get32(const char *lds) {
        const u32 v = look();
        advance(lds, 32);
        return v;
    }
};

// floor(d / r) for any d < 2^32 and r >= 1, exactly: trunc((d + 0.5) * x) in binary64 with x = 1 / r good to 2^-42
// (v_rcp_f32 of the rounded divisor, 2^-21, squared by one Newton step on the exact divisor).  (d + 0.5) / r is at
// least 0.5 / r away from an integer, the error of the product is below (d + 0.5) / r * 2^-41 -- less than that for
// every d < 2^32.  Ten full-rate instructions; the float-with-directed-rounding version this replaces compiled to
// 35 (the rounding-mode conversions and the IEEE reciprocal are emulated on gfx950).
__device__ __forceinline__ u32 rg_div(u32 d, u32 r) {
    const double rd = (double)r;
    const double x0 = (double)__builtin_amdgcn_rcpf((float)r);
    const double e = __builtin_fma(-rd, x0, 1.0);
    const double x = __builtin_fma(x0, e, x0);
    return (u32)(((double)d + 0.5) * x);
}

// The same quotient for totals >= 256 in binary32: then r = range // M < 2^24 converts exactly and, for every state a
// valid stream can produce, q = d // r < M <= 2^16, so fl(fl(d) * fl(1/r)) is within q 2^-22 < 2^-6 of d / r (three
// roundings of 2^-24, 2^-23, 2^-24); biased down by 2^-5 inside the FMA it lies in (d/r - 0.047, d/r - 0.015), i.e. its
// floor is q or q - 1, and one remainder test settles which: 9 instructions, none of them binary64 (the ten above cost
// ~24 ns per wave on a SIMD, these ~16: tools/ubench/valu_rate.hip).
// The conversion is the hardware's (v_cvt_u32_f32: negative -> 0, >= 2^32 -> 0xFFFFFFFF), written as an instruction
// because the C++ cast is undefined outside [0, 2^32) -- a negative value occurs for d = 0 (valid), a huge one only for
// CORRUPT input (the normalisation keeps range >= BOTTOM >= M, so r >= 1, and d / r >= 2^32 needs r = 1 and
// d >= 2^32 - 2^7): there q saturates, q + 1 wraps to 0 and this path decodes slot 0 where rg_div and the generic
// kernel decode the last slot -- both outputs are garbage the range coder has no means to detect.
__device__ __forceinline__ u32 rg_div32(u32 d, u32 r) {
    const float qf = __builtin_fmaf((float)d, __builtin_amdgcn_rcpf((float)r), -0.03125f);
    u32 q;  // q or q - 1
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(qf));
    const u32 rem = d - __umul24(q, r);   // q < 2^17, r < 2^24; rem in [0, 2r)
    u32 t, qn;  // q + 1 - [rem < r]: the borrow of rem - r IS the test
    asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_subb_co_u32 %1, vcc, %4, -1, vcc"
        : "=&v"(t), "=v"(qn) : "v"(rem), "v"(r), "v"(q) : "vcc");
    return qn;
}

// one symbol: search (:225-238), shrink_range, normalize (:240-267); returns the symbol
// LUT: totals up to 4096 find the symbol in a slot -> symbol table; larger ones (up to BOTTOM = 2^16) by an 8-step
// binary search on the cumulative counts (K = alphabet size rides in md.k)
// The common case of one decoded symbol and nothing else (round 5, RGD_PAIR): search, shrink_range, the leading bytes low
// and low + range agree on -- taken from `lk`, the next 32 bits of the stream, which are consumed from its top: ONE 64-bit
// shift of state : lk yields the new state and the look-ahead that is left.  No reader access, no branch; returns the
// symbol, `nb` bits consumed and `left` = the range after them (below BOTTOM: the carry-less reset is due and nothing
// returned here holds -- the caller replays the symbol with rg_decode_symbol).  Defined for any input, like
// rg_fast_symbol.
template <int MODE, bool LUT, bool DIV32>
__device__ __forceinline__ u32 rg_decode_fast(u32 &low, u32 &range, u32 &state, u32 &lk, u32 &nb, u32 &left,
                                              const char *tab, const RgDivM &md, u32 slot_max) {
    const u32 rr = rg_range_over_m<MODE>(range, md);
    u32 q = DIV32 ? rg_div32(state - low, rr) : rg_div(state - low, rr);
    q = min(q, slot_max);
    u32 s;
    uint2 e;
    if (RG_UNI(MODE)) {
        s = (MODE == 3) ? q : (q >> md.t);
        e = make_uint2(0u, 0u);
    } else if (LUT) {
        e = *reinterpret_cast<const uint2 *>(tab + q * 8);
        s = e.x >> 24;
        if (!DIV32) e.x &= 0xFFFFFFu;
    } else {
        s = 0;
#pragma unroll
        for (u32 b = 128; b > 0; b >>= 1) {
            const u32 t = s + b;
            const u32 ct = *reinterpret_cast<const u32 *>(tab + min(t, 255u) * 8);
            s = (t < md.k && ct <= q) ? t : s;
        }
        e = *reinterpret_cast<const uint2 *>(tab + s * 8);
    }
    u32 low0, range0;
    if (RG_UNI(MODE)) {
        range0 = (MODE == 3) ? rr : (rr << md.t);
        low0 = __umul24(s, range0) + low;
    } else if (DIV32) {
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(low0) : "v"(e.x), "v"(rr), "v"(low));
        range0 = __umul24(rr, e.y);
    } else {
        low0 = low + e.x * rr;
        range0 = rr * e.y;
    }
    u32 lz;
    asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(low0 ^ (low0 + range0)));
    const u32 sh = lz & 0x18u;
    const u64 t = ((((u64)state) << 32) | lk) << sh;
    state = (u32)(t >> 32);
    lk = (u32)t;
    // low and low + range agree on their top sh bits, so range0 < 2^(32 - sh): shifting the pair low0 : range0 left by sh
    // moves nothing of range0 into low0's word -- one 64-bit shift for both (the bytes leaving the top of low0 are not
    // needed here: the decoder reads them from the stream)
    const u64 lr = ((((u64)low0) << 32) | range0) << sh;
    low = (u32)(lr >> 32);
    range = (u32)lr;
    nb = sh;
    left = range;
    return s;
}

template <int MODE, bool LUT, bool DIV32, typename Reader>
__device__ __forceinline__ u32 rg_decode_symbol(u32 &low, u32 &range, u32 &state, Reader &r, char *lds, const char *tab,
                                                const RgDivM &md, u32 slot_max) {
    const u32 rr = rg_range_over_m<MODE>(range, md);
    u32 q = DIV32 ? rg_div32(state - low, rr) : rg_div(state - low, rr);
    q = min(q, slot_max);  // state in the slack above c[K-1] + f[K-1] maps to the last symbol
    u32 s;
    uint2 e;
    if (RG_UNI(MODE)) {  // the slot IS the symbol (times f): no table read on the serial chain
        s = (MODE == 3) ? q : (q >> md.t);
        e = make_uint2(0u, 0u);
    } else if (LUT) {  // one read: slot -> {c | s << 24, f}
        e = *reinterpret_cast<const uint2 *>(tab + q * 8);
        s = e.x >> 24;
        if (!DIV32) e.x &= 0xFFFFFFu;  // (the 24-bit multiply-add below does not see the symbol byte)
    } else {
        s = 0;
#pragma unroll
        for (u32 b = 128; b > 0; b >>= 1) {
            const u32 t = s + b;
            const u32 ct = *reinterpret_cast<const u32 *>(tab + min(t, 255u) * 8);
            s = (t < md.k && ct <= q) ? t : s;
        }
        e = *reinterpret_cast<const uint2 *>(tab + s * 8);
    }
    if (RG_UNI(MODE)) {
        const u32 rf = (MODE == 3) ? rr : (rr << md.t);  // < 2^24, see rg_encode_symbol
        low = __umul24(s, rf) + low;
        range = rf;
    } else if (DIV32) {
        // totals >= 256: rr = range // M < 2^24, and c, f < 2^24 (the symbol byte on top of c is outside the 24 bits the
        // instruction reads): two full-rate 24-bit multiplies instead of v_mad_u64_u32 + v_mul_lo_u32 (quarter rate each).
        // c rr <= range and f rr <= range: no overflow (carry-less coder)
        u32 nl;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(nl) : "v"(e.x), "v"(rr), "v"(low));
        low = nl;
        range = __umul24(rr, e.y);
    } else {
        low += e.x * rr;
        range = rr * e.y;
    }
    const u32 lk = r.look(lds);
    // closed form of the common case computed unconditionally, ONE branch for everything rare (see rg_encode_symbol)
    const u32 low0 = low, range0 = range, state0 = state;
    const u32 nb1 = (u32)__builtin_clz(low0 ^ (low0 + range0)) >> 3;
    const u32 sh = 8 * nb1;
    const u32 range_s = range0 << sh;
    state = (u32)(((((u64)state0) << 32) | lk) << sh >> 32);  // the next nb1 bytes (sh may be 0): one 64-bit shift
    low = low0 << sh;
    range = range_s;
    u32 adv = sh;
    if (__builtin_expect(range_s < RG_BOTTOM, 0)) {
        low = low0;
        range = range0;
        state = state0;
        u32 nb = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool settled = ((low ^ (low + range)) < RG_TOP);
            if (!settled && range >= RG_BOTTOM) break;
            if (!settled) range = (0u - low) & (RG_BOTTOM - 1);
            state = (state << 8) | ((lk >> (24 - 8 * j)) & 0xFFu);
            ++nb;
            low <<= 8;
            range <<= 8;
        }
        r.advance(lds, 8 * nb);
        adv = 0;
        while (nb == 4 && rg_needs_byte(low, range)) {  // mirror of the encoder's guard
            state = (state << 8) | (r.look(lds) >> 24);
            r.advance(lds, 8);
            low <<= 8;
            range <<= 8;
        }
    }
    r.advance(lds, adv);
    return s;
}

// STRIPED: the input is in wave-striped slots (RgOutT / AnsBitReaderT); `in_size_bytes` then carries the slot stride and
// stream c lies in slot c
template <int MODE, bool LUT, bool DIV32, bool STRIPED = false>
__global__ void __launch_bounds__(RGD_THREADS) range_decode_fast_kernel(RangeFastDev P, const u8 *__restrict__ in,
                                                                       u64 in_size_bytes,
                                                                       const u64 *__restrict__ bit_off,
                                                                       const u32 *__restrict__ in_nbits, u64 n_chunks,
                                                                       u8 *__restrict__ out_sym, u64 out_stride,
                                                                       u32 out_cap, u32 *__restrict__ out_lens,
                                                                       u32 *__restrict__ consumed,
                                                                       u32 *__restrict__ status) {
    // [0, 128 KiB) word ring, then the table the symbol search reads: slot -> {c | s << 24, f} for totals up to 4096
    // (32 KiB, one read per symbol), else the 256 {c, f} pairs the binary search walks
    __shared__ __attribute__((aligned(16))) char s_lds[RGD_RING_BYTES + (LUT ? 4096 * 8 : 256 * 8)];
    char *lds = s_lds;
    const char *tab = s_lds + RGD_RING_BYTES;
    const u32 M = P.M;
    if (LUT) {
        for (u32 i = threadIdx.x; i < M; i += RGD_THREADS) {
            const u32 sy = P.d_slot2sym[i];
            const uint2 cf = P.d_enc_tab[sy];
            reinterpret_cast<uint2 *>(s_lds + RGD_RING_BYTES)[i] = make_uint2(cf.x | (sy << 24), cf.y);
        }
    } else if (threadIdx.x < 256) {
        reinterpret_cast<uint2 *>(s_lds + RGD_RING_BYTES)[threadIdx.x] = P.d_enc_tab[threadIdx.x];
    }
    __syncthreads();
    const u64 c = (u64)blockIdx.x * RGD_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 avail = in_nbits[c];
    u32 st = 0;
    if (avail < 64) {  // size header + the PRECISION/8 state bytes do not fit
        out_lens[c] = 0;
        consumed[c] = 64;
        if (status) status[c] = SCL_ST_TRUNCATED;
        return;
    }
    // the rANS decoder's windowless reader (scl_ans_fast_io.h): the 32 bits at the position come straight out of the ring,
    // once per PAIR of symbols; nothing is advanced under a branch
    typename std::conditional<STRIPED, AnsBitReaderT<RGD_THREADS>, AnsBitReaderW<RGD_THREADS>>::type r;
    if constexpr (STRIPED)
        r.init(in, in_size_bytes, c, bit_off[c], lds, threadIdx.x);
    else
        r.init(in, in_size_bytes, bit_off[c], lds, threadIdx.x);
    u32 n = r.get(lds, 32);
    u32 state = r.get(lds, 32);  // the first four bytes of the body (:289-291)
    out_lens[c] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    u8 *dst = out_sym + c * out_stride;
    u32 low = 0, range = 0xFFFFFFFFu;
    const u32 slot_max = M - 1;
    RgDivM md;
    md.m_log2 = P.m_log2;
    md.k = P.K;
    md.t = P.uni_t;
    md.inv_m = 1.0 / (double)P.M;

    u32 i = 0;
    // 128 symbols per iteration: eight registers, one burst of eight 16-byte stores (a whole line).  The b loop must be
    // unrolled in full (plain "#pragma unroll" gives up on a body of 16 symbols: a[] then lives in scratch memory, eight
    // scratch stores and loads per line -- 1.44 instead of 1.35 ms).  (The wave-cooperative store of the rANS / tANS
    // decoders, CoopLineStore, made this instruction-bound kernel 8 % slower.)
#pragma nounroll
    for (; i + 128 <= n; i += 128) {
        uint4 a[8];
#pragma clang loop unroll(full)
        for (int b = 0; b < 8; ++b) {
            u32 ow[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                u32 o = 0;
                // Both symbols of a pair on the common path out of ONE 32-bit look-ahead, unconditionally; one test and one
                // branch per pair for everything else (a carry-less reset in either symbol, more than 32 bits for the two):
                // those lanes replay the pair from the state of before it with the exact routine.
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    u32 lk = r.look(lds);
                    const u32 low_s = low, range_s = range, state_s = state;
                    u32 n0, n1, l0, l1;
                    u32 s0 = rg_decode_fast<MODE, LUT, DIV32>(low, range, state, lk, n0, l0, tab, md, slot_max);
                    u32 s1 = rg_decode_fast<MODE, LUT, DIV32>(low, range, state, lk, n1, l1, tab, md, slot_max);
                    const u32 nn = n0 + n1;
                    if (__builtin_expect((min(l0, l1) < RG_BOTTOM) | (nn > 32), 0)) {
                        low = low_s;
                        range = range_s;
                        state = state_s;
                        s0 = rg_decode_symbol<MODE, LUT, DIV32>(low, range, state, r, lds, tab, md, slot_max);
                        s1 = rg_decode_symbol<MODE, LUT, DIV32>(low, range, state, r, lds, tab, md, slot_max);
                    } else {
                        r.advance(lds, nn);
                    }
                    o |= (s0 << (8 * j)) | (s1 << (8 * j + 8));
                }
                if (d & 1) r.maybe_refill(lds);  // every eight symbols: <= 8 words even if each took the four-byte path
                asm volatile("" : "+v"(o) : : "memory");
                ow[d] = o;
            }
            a[b] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        uint4 *p = reinterpret_cast<uint4 *>(dst + i);
#pragma unroll
        for (int b = 0; b < 8; ++b) p[b] = a[b];
    }
    for (; i < n; ++i) {  // ragged tail
        dst[i] = (u8)rg_decode_symbol<MODE, LUT, DIV32>(low, range, state, r, lds, tab, md, slot_max);
        if ((i & 3u) == 3u) r.maybe_refill(lds);
    }
    const u32 used_bits = r.consumed();
    if (used_bits > avail) st |= SCL_ST_TRUNCATED;
    consumed[c] = used_bits;
    if (status) status[c] = st;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
int range_fast_build_tables(scl_range_model *m, const u32 *h_freq, const u32 *h_cum) {
    const RangeDev &D = m->dev;
    m->fast = 0;
    // any total the coder allows (M <= BOTTOM = 2^16, range_coder.py:85); the decoder's slot -> symbol table sits in
    // LDS for totals up to 4096, larger ones search the cumulative counts
    if (D.P != 32 || D.size_bits != 32 || D.M < 1 || D.M > 65536 || !m->d_slot2sym) return SCL_OK;
    std::vector<uint2> tab(256);
    for (u32 s = 0; s < 256; ++s) {
        const u32 src = s < D.K ? s : 0;
        tab[s] = make_uint2(h_cum[src], h_freq[src]);
    }
    hipError_t e = hipMalloc((void **)&m->d_enc_tab, 256 * sizeof(uint2));
    if (e == hipSuccess) e = hipMemcpy(m->d_enc_tab, tab.data(), 256 * sizeof(uint2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("range_model_create: fast-path table upload failed: %s", hipGetErrorString(e));
        return SCL_E_HIP;
    }
    m->fdev.K = D.K;
    m->fdev.m_log2 = D.m_log2;
    m->fdev.M = D.M;
    m->fdev.uni_t = 0xFFFFFFFFu;
    if (D.K == 256 && D.m_log2 != 0xFFFFFFFFu && D.m_log2 >= 8) {
        bool same = true;
        for (u32 s = 1; s < 256; ++s) same = same && h_freq[s] == h_freq[0];
        if (same) m->fdev.uni_t = D.m_log2 - 8;  // f = M / 256
    }
    m->fdev.d_enc_tab = m->d_enc_tab;
    m->fdev.d_slot2sym = m->d_slot2sym;
    m->fast = 1;
    return SCL_OK;
}

// the models the striped kernels are instantiated for: uniform bytes (f = 1, M = 256: configs[2]) and tables with totals
// 256..4096 (a 256-symbol table like T256 among them) -- the modes the headline-sized batches use
bool range_fast_striped_ok(const scl_range_model *m) {
    if (!m->fast) return false;
    if (m->fdev.uni_t == 0) return true;
    return m->fdev.uni_t == 0xFFFFFFFFu && m->fdev.M >= 256 && m->fdev.M <= 4096;
}

void range_fast_encode_launch(const scl_range_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                              u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                              u32 *d_status, hipStream_t st, bool striped) {
    const u32 blocks = (u32)((n_chunks + RGE_THREADS - 1) / RGE_THREADS);
    if (striped) {  // (range_fast_striped_ok holds: the caller checked)
#define RG_LAUNCH_ENC_T(MODE)                                                                                         \
    hipLaunchKernelGGL((range_encode_fast_kernel<MODE, RgOutT>), dim3(blocks), dim3(RGE_THREADS), 0, st, m->fdev, d_sym, \
                       sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_bit_off, d_nbits, d_status)
        if (m->fdev.uni_t == 0) RG_LAUNCH_ENC_T(3);
        else if (m->fdev.m_log2 != 0xFFFFFFFFu) RG_LAUNCH_ENC_T(4);
        else RG_LAUNCH_ENC_T(5);
#undef RG_LAUNCH_ENC_T
        return;
    }
#define RG_LAUNCH_ENC(MODE)                                                                                      \
    hipLaunchKernelGGL(range_encode_fast_kernel<MODE>, dim3(blocks), dim3(RGE_THREADS), 0, st, m->fdev, d_sym,       \
                       sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_bit_off, d_nbits, d_status)
    if (m->fdev.uni_t == 0) RG_LAUNCH_ENC(3);
    else if (m->fdev.uni_t != 0xFFFFFFFFu) RG_LAUNCH_ENC(2);
    else if (m->fdev.m_log2 != 0xFFFFFFFFu) {
        if (m->fdev.M >= 256) RG_LAUNCH_ENC(4); else RG_LAUNCH_ENC(0);
    } else {
        if (m->fdev.M >= 256) RG_LAUNCH_ENC(5); else RG_LAUNCH_ENC(1);
    }
#undef RG_LAUNCH_ENC
}

void range_fast_decode_launch(const scl_range_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                              const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                              u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st, bool striped) {
    const u32 blocks = (u32)((n_chunks + RGD_THREADS - 1) / RGD_THREADS);
    if (striped) {  // totals 256..4096: slot table + binary32 quotient (range_fast_striped_ok)
#define RG_LAUNCH_DEC_T(MODE)                                                                                          \
    hipLaunchKernelGGL((range_decode_fast_kernel<MODE, true, true, true>), dim3(blocks), dim3(RGD_THREADS), 0, st, m->fdev, \
                       d_in, in_size_bytes, d_bit_off, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,   \
                       d_consumed, d_status)
        if (m->fdev.uni_t == 0) RG_LAUNCH_DEC_T(3);
        else if (m->fdev.m_log2 != 0xFFFFFFFFu) RG_LAUNCH_DEC_T(0);
        else RG_LAUNCH_DEC_T(1);
#undef RG_LAUNCH_DEC_T
        return;
    }
#define RG_LAUNCH_DEC2(MODE, LUT, DIV32)                                                                          \
    hipLaunchKernelGGL((range_decode_fast_kernel<MODE, LUT, DIV32>), dim3(blocks), dim3(RGD_THREADS), 0, st, m->fdev, \
                       d_in,                                                                                      \
                       in_size_bytes, d_bit_off, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, d_out_lens,   \
                       d_consumed, d_status)
    // totals of 256 and more divide in binary32 (rg_div32)
#define RG_LAUNCH_DEC(MODE, LUT)                                       \
    do {                                                               \
        if (m->fdev.M >= 256) RG_LAUNCH_DEC2(MODE, LUT, true);         \
        else RG_LAUNCH_DEC2(MODE, LUT, false);                         \
    } while (0)
    const bool pow2 = m->fdev.m_log2 != 0xFFFFFFFFu, lut = m->fdev.M <= 4096;
    if (m->fdev.uni_t == 0) RG_LAUNCH_DEC2(3, true, true);
    else if (m->fdev.uni_t != 0xFFFFFFFFu) RG_LAUNCH_DEC2(2, true, true);
    else if (pow2 && lut) RG_LAUNCH_DEC(0, true);
    else if (pow2) RG_LAUNCH_DEC(0, false);
    else if (lut) RG_LAUNCH_DEC(1, true);
    else RG_LAUNCH_DEC(1, false);
#undef RG_LAUNCH_DEC
#undef RG_LAUNCH_DEC2
}
