// scl_aec_lane_io.h -- per-lane stream I/O and packed-u16 helpers of the arithmetic-coder kernels that keep a
// private model per lane in LDS (scl_aec_fast.hip: order-k rows, scl_aec_iid.hip: two-level i.i.d. table).
// These kernels run one wave per SIMD (LDS-bound occupancy), so few lines are open per CU and plain 4-byte
// accesses are merged by L2; the high-occupancy kernels use the line-granular I/O of scl_ans_fast_io.h instead.
// Internal to csrc/.
#pragma once
#include "scl_common.h"

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// LDS rows are read as 2-byte elements and written as 16-byte halves: the accesses must not be reordered by
// type-based alias analysis
typedef u16 __attribute__((may_alias)) u16_lds;
typedef u32 __attribute__((may_alias)) u32_lds;
typedef uint4 __attribute__((may_alias)) uint4_lds;

__device__ __forceinline__ u32 af_pk_add(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
// acc += (e > t) ? -1 : 0 per 16-bit half (all values < 2^15)
__device__ __forceinline__ u32 af_pk_count_gt(u32 acc, u32 t, u32 e) {
    s16x2 d = __builtin_bit_cast(s16x2, t) - __builtin_bit_cast(s16x2, e);
    d = d >> (s16x2)(15);
    return __builtin_bit_cast(u32, (s16x2)(__builtin_bit_cast(s16x2, acc) + d));
}

// decoder search over one row of 16 nondecreasing u16 (8 packed registers): m[r] = (e > t) ? 0xFFFF : 0 per half, and the
// number of entries <= t.  The row is sorted, so e[j] > t  <=>  j >= s: the masks ARE update_model's increment
// (Y[j] += 1 for j >= s is Y - m), no mask table to read; the count comes out of v_dot2c_i32_i16 without a combine step.
__device__ __forceinline__ u32 af_pk_search16(const u32 (&e)[8], u32 t, u32 (&m)[8]) {
    const s16x2 tp = {(short)t, (short)t};  // a splat: the compare reads the low half twice (op_sel), no v_lshl_or
    const s16x2 one = {1, 1};
    int acc;
#pragma unroll
    for (u32 r = 0; r < 8; ++r) {
        const s16x2 d = (tp - __builtin_bit_cast(s16x2, e[r])) >> (s16x2)(15);
        m[r] = __builtin_bit_cast(u32, d);
        if (r == 0)  // the three-address form with the constant 16 as accumulator: no v_mov to start the chain
            asm("v_dot2_i32_i16 %0, %1, %2, 16" : "=v"(acc) : "v"(d), "s"(0x10001));
        else
            acc = __builtin_amdgcn_sdot2(d, one, acc, false);
    }
    return (u32)acc;
}
__device__ __forceinline__ u32 af_pk_sub(u32 a, u32 b) {
    return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}

// ---- forward bit writer: completed big-endian words go straight to the slot (4-byte stores; the stream is a
// third of the input and L2 merges them -- a register FIFO costs ~50 phi copies per symbol, an LDS ring does not fit)
// STAGED (round 4): completed words collect in 64 bytes of LDS per lane ([thread][64 bytes] at `lds_stage`) and leave as
// whole 64-byte sectors, four back-to-back 16-byte stores -- for kernels that have the 16 KiB to spare.
template <bool STAGED = false>
struct AfWriterT {
    u32 hi;    // pending bits, right-aligned (the oldest is the most significant), < 32 of them
    u32 nacc;  // number of pending bits
    u32 *dst;
    u32 nwords;
    char *stage;

    __device__ __forceinline__ void init(u8 *slot, char *lds_stage = nullptr, u32 tid = 0) {
        hi = 0;
        nacc = 0;
        nwords = 0;
        dst = reinterpret_cast<u32 *>(slot);
        stage = STAGED ? lds_stage + tid * 64 : nullptr;
    }
    __device__ __forceinline__ void emit(u32 word_be) {
        if (!STAGED) {
            dst[nwords++] = word_be;
            return;
        }
        *reinterpret_cast<u32_lds *>(stage + (nwords & 15u) * 4) = word_be;
        ++nwords;
        if ((nwords & 15u) == 0) {
            const uint4 q0 = *reinterpret_cast<const uint4_lds *>(stage);
            const uint4 q1 = *reinterpret_cast<const uint4_lds *>(stage + 16);
            const uint4 q2 = *reinterpret_cast<const uint4_lds *>(stage + 32);
            const uint4 q3 = *reinterpret_cast<const uint4_lds *>(stage + 48);
            uint4 *p = reinterpret_cast<uint4 *>(dst + (nwords - 16));
            p[0] = q0;
            p[1] = q1;
            p[2] = q2;
            p[3] = q3;
        }
    }
    __device__ __forceinline__ void put(u32 v, u32 nb) {  // v < 2^nb, nb <= 32
        // 32-bit arithmetic on purpose: see AnsFwdWriter::put (scl_ans_fast_io.h)
        const u32 tot = nacc + nb;
        if (tot >= 32) {
            const u32 r = tot - 32;  // <= 31; nb - r = 32 - nacc
            const u32 word = (r == 0) ? ((hi << (nb & 31)) | v) : ((hi << (nb - r)) | (v >> r));
            emit(__builtin_bswap32(word));
            hi = v & ((1u << r) - 1u);
            nacc = r;
        } else {
            hi = (hi << nb) | v;
            nacc = tot;
        }
    }
    // put for the per-symbol field of the encoders (nb <= 32, nb = 0 allowed with v = 0): the pending bits and the field
    // side by side in 64 bits, so that only the completed word is conditional -- no "which half" case split, no values
    // merged after a branch (put above costs the lone wave ~8 register copies per symbol for those).
    __device__ __forceinline__ void put_field(u32 v, u32 nb) {
        const u32 tot = nacc + nb;                    // <= 63
        const u64 wide = ((u64)hi << nb) | v;         // tot valid bits, right-aligned
        nacc = tot & 31u;
        if (tot >= 32) emit(__builtin_bswap32((u32)(wide >> nacc)));  // tot - 32 = tot & 31 here
        hi = (u32)wide & ((1u << nacc) - 1u);
    }
    __device__ __forceinline__ void put_run(u32 bit, u32 count) {
        while (count >= 32) {
            put(bit ? 0xFFFFFFFFu : 0u, 32);
            count -= 32;
        }
        if (count) put(bit ? ((1u << count) - 1u) : 0u, count);
    }
    __device__ __forceinline__ u64 finish() {
        const u64 total = (u64)nwords * 32 + nacc;
        if (STAGED)  // the words of the last, incomplete sector
            for (u32 w = nwords & ~15u; w < nwords; ++w) dst[w] = *reinterpret_cast<const u32_lds *>(stage + (w & 15u) * 4);
        if (nacc) dst[nwords] = __builtin_bswap32(hi << (32 - nacc));
        return total;
    }
};
typedef AfWriterT<false> AfWriter;

// ---- forward bit reader: 4-byte loads, one word ahead; bits past the end of the stream read as 0 ---------------
struct AfReader {
    // Two 32-bit words of the stream in registers: the unread bits are the low `r` bits of `a` (0 <= r <= 31) followed by `b`,
    // so the next 32 bits are one v_alignbit(a, b, r) -- no 64-bit window to shift (round 4; the window cost two 64-bit
    // shifts, a counter and an "nb == 0" branch per symbol).
    // A refill is taken by SOME lane of the wave at nearly every symbol (the lanes' word boundaries are out of phase), so
    // its instruction count is paid per symbol: one 32-bit counter does both the bounds check and the zero fill.
    const u32 *ptr;  // address of the word held in `ahead`
    u32 ahead;       // raw (memory-order) word, loaded one refill early
    u32 a, b;
    int r;
    u32 left;        // words of the stream not yet moved into a/b, the partial last one included, clipped to the buffer
    u32 tail;        // mask of the stream's bits in its last word (all ones if it ends on a word boundary); 0 once used
    const u32 *ptr0; // position(): ptr at init,
    u32 past;        //             refills that did not advance ptr (at and past the end of the stream),
    u32 skip;        //             bits in front of the stream in its first word

    // The word in `ahead` becomes part of the window; bits past the end of the stream read as zero (the arithmetic
    // decoder looks PRECISION bits ahead, arithmetic_coding.py:222-229), and no load goes past the stream's last word.
    // UNCHECKED: for a stretch of symbols over which the caller has established that no lane reaches the last word of its
    // stream (safe_symbols below): no mask, no counter -- six instructions instead of twelve.  The caller settles `left`
    // afterwards (settle).
    template <bool UNCHECKED = false>
    __device__ __forceinline__ u32 next_word() {
        if (UNCHECKED) {
            const u32 v = __builtin_bswap32(ahead);
            ptr += 1;
            ahead = *ptr;
            return v;
        }
        const bool more = left > 1u;
        const u32 v = __builtin_bswap32(ahead) & (more ? 0xFFFFFFFFu : tail);
        tail = more ? tail : 0u;
        past += more ? 0u : 1u;
        left = left ? left - 1u : 0u;
        ptr = reinterpret_cast<const u32 *>(reinterpret_cast<const char *>(ptr) + (more ? 4 : 0));
        ahead = *ptr;
        return v;
    }
    // number of symbols this lane can decode with the unchecked refill: a symbol takes at most two words (closed form: <= 31
    // bits; the literal loops of a corner: <= 63), and the last word of the stream must not be reached
    __device__ __forceinline__ u32 safe_symbols() const { return left ? (left - 1u) >> 1 : 0u; }
    __device__ __forceinline__ void settle(const u32 *ptr_before) { left -= (u32)(ptr - ptr_before); }
    __device__ __forceinline__ void init(const u8 *in, u64 in_size_bytes, u64 bit_off, u32 nbits) {
        const u64 nwords = in_size_bytes >> 2;  // readable 32-bit words
        const u64 wi = min(bit_off >> 5, nwords - 1);
        const u32 skipb = (u32)bit_off & 31u;
        const u64 sbits = (u64)nbits + skipb;                       // from the start of word wi
        const u64 swords = (sbits + 31) >> 5;
        const u64 avail = (bit_off >> 5) < nwords ? nwords - wi : 0;  // a stream that starts past the buffer reads as zeros
        left = (u32)min(min(swords, avail), (u64)0xFFFFFFFFu);
        const u32 tb = (u32)sbits & 31u;
        tail = (left < swords) ? 0xFFFFFFFFu : (tb ? ~(0xFFFFFFFFu >> tb) : 0xFFFFFFFFu);
        if (left == 0) tail = 0;
        ptr = reinterpret_cast<const u32 *>(in) + wi;
        ptr0 = ptr;
        past = 0;
        skip = skipb;
        ahead = *ptr;
        a = 0;
        b = next_word();
        r = 0;
        consume(skipb);  // drop the bits in front of the stream
    }
    // bits of the stream consumed so far: every refill moved 32 bits in (ptr advanced, or `past` counted it), r + 32 are unread.
    // The decoders derive num_bits_consumed from this at the end instead of adding up k + m symbol by symbol.
    __device__ __forceinline__ u32 position() const { return 32u * ((u32)(ptr - ptr0) + past) - 32u - (u32)r - skip; }
    __device__ __forceinline__ u32 look() const { return __builtin_amdgcn_alignbit(a, b, (u32)r); }  // the next 32 bits
    template <bool UNCHECKED = false>
    __device__ __forceinline__ void consume(u32 nb) {  // nb <= 31
        r -= (int)nb;
        if (r < 0) {
            a = b;
            b = next_word<UNCHECKED>();
            r += 32;
        }
    }
    template <bool UNCHECKED = false>
    __device__ __forceinline__ u32 get(u32 nb) {  // nb <= 32; headers, the first state and the literal loops
        const u32 l = look();
        if (nb == 32) {
            a = b;
            b = next_word<UNCHECKED>();
            return l;
        }
        consume<UNCHECKED>(nb);
        return nb ? l >> (32 - nb) : 0u;
    }
};

// the closed-form renormalisation's state update: kt = k + m <= 31 bits come in from the stream, the bit k steps below the
// top is kept on top (the E3 steps), arithmetic_coding.py:245-275.  {state, look} << kt is one 64-bit shift and kt = 0 needs
// no special case.
template <bool UNCHECKED = false>
__device__ __forceinline__ u32 af_state_shift_in(AfReader &rd, u32 state, u32 k, u32 kt) {
    u32 l = rd.look();
    asm volatile("" : "+v"(l));  // taken before the refill branch, so that a/b/r are updated in place there
    const u64 both = (((u64)state << 32) | l) << kt;
    const u32 keep = (state << k) & 0x80000000u;
    rd.consume<UNCHECKED>(kt);
    return ((u32)(both >> 32) & 0x7FFFFFFFu) | keep;
}

// minimum of v over the lanes of the wave that execute this.  The decode kernels call it after lanes have returned
// (chunk >= n_chunks, a refused header): a butterfly would then read registers of disabled lanes -- ds_bpermute happens
// to return 0 for those today, which made the count 0 and the caller fall back to its checked loop, but that is not a
// contract, and a DPP lowering of the same shuffle would return garbage, i.e. an undersized minimum and unchecked reads
// past a stream.  So: the butterfly only when all 64 lanes are here (the batch case); any other wave reads the live lanes'
// values one by one through the scalar unit (v_readlane over the exec mask: ~6 scalar instructions per live lane, on a
// path only partial waves and waves with a damaged chunk take) -- and still gets its unchecked stretches.
__device__ __forceinline__ u32 af_wave_min(u32 v) {
    u64 live = __builtin_amdgcn_ballot_w64(true);
    if (live == ~0ull) {
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) v = min(v, (u32)__shfl_xor((int)v, sft, 64));
        return v;
    }
    u32 r = 0xFFFFFFFFu;
    while (live) {
        const int l = __builtin_ctzll(live);
        r = min(r, (u32)__builtin_amdgcn_readlane((int)v, l));
        live &= live - 1;
    }
    return r;
}

// ---- decoded symbols: four to a word, sixteen words to a 64-byte sector staged in LDS ([thread][64 bytes]), stored as four
// back-to-back 16-byte stores.  One 4-byte store per four symbols -- what these kernels did until round 4 -- is not merged by
// the memory system at 262 144 open output lines: 12.6 GB of HBM writes for 1 GiB of symbols (profiles/traffic.json, r03).
// `i` is the symbol's index in its chunk: the same for every live lane of the wave (the decode loops run in lockstep), so
// the sector branch is wave-uniform.
struct AfSymOut {
    char *stage;
    u8 *dst;
    u32 oword;
    __device__ __forceinline__ void init(char *lds_stage, u32 tid, u8 *row) {
        stage = lds_stage + tid * 64;
        dst = row;
        oword = 0;
    }
    __device__ __forceinline__ void put(u32 s, u32 i) {
        oword |= s << (8 * (i & 3));
        if ((i & 3) == 3) {
            *reinterpret_cast<u32_lds *>(stage + ((i >> 2) & 15) * 4) = oword;
            oword = 0;
            if ((i & 63) == 63) {
                const uint4 q0 = *reinterpret_cast<const uint4_lds *>(stage);
                const uint4 q1 = *reinterpret_cast<const uint4_lds *>(stage + 16);
                const uint4 q2 = *reinterpret_cast<const uint4_lds *>(stage + 32);
                const uint4 q3 = *reinterpret_cast<const uint4_lds *>(stage + 48);
                uint4 *p = reinterpret_cast<uint4 *>(dst + (i - 63));
                p[0] = q0;
                p[1] = q1;
                p[2] = q2;
                p[3] = q3;
            }
        }
    }
    // the words of the last, incomplete sector and the partial word (zero-padded inside the row); n = symbols put
    __device__ __forceinline__ void finish(u32 n) {
        const u32 done = n & ~63u;
        u32 *d32 = reinterpret_cast<u32 *>(dst + done);
        const u32 full = (n - done) >> 2;
        for (u32 w = 0; w < full; ++w) d32[w] = *reinterpret_cast<const u32_lds *>(stage + w * 4);
        if ((n & 3) != 0) d32[full] = oword;
    }
};
