// scl_rans_fast.hip -- the gfx950 fast path of batched rANS (BASELINE.json configs[1] / headline):
// u32 state, any total 2 <= M <= 4096 (a power of two gets shifts and masks, any other total an exact binary64
// division in the decoder), NUM_BITS_OUT = 1, RANGE_FACTOR = 2^r <= 2^23, H < 2^31, any alphabet 2..256.
// Same bit stream as the generic kernels in scl_rans.hip (and as reference rANS.py:186-210 / :270-297);
// what changes is how a lane spends its instructions:
//
//  encode, per symbol (one 16-byte LDS table read, issued one word = four symbols ahead of its use):
//    k     = k_lo[s] + (x >= thresh[s])                  closed form of shrink_state's while-loop
//                                                         (rANS.py:149-161; tANS.py:74-86 gives the same rule)
//    field = low k bits of x;  xs = x >> k
//    q     = floor(xs / f) = mulhi(x << (32-nsb), rcp[s]) >> (s[s] + k),  s[s] + k_lo[s] == m
//            with rcp = ceil(2^(nsb+s) / f), s = ceil(log2 f): exact for every x < 2^nsb
//            (error term x*e/(f*2^(nsb+s+k)) < 2^-(s+k) <= 1/(f*2^k))
//            -- and the threshold test is read off the same product: q0 = floor(x / (f 2^k_lo)) >= 2 RF  <=>  x >= thresh
//    x     = xs + c[s] + q*(M - f[s])                     == (xs//f)*M + c + xs%f  (rANS.py:138-147)
//    The field is never extracted: v_alignbit shifts the low k bits of x straight into a 64-bit bit window; completed
//    big-endian words go to a per-lane LDS ring and leave as whole 128-byte lines, back to front, stored by the four
//    lanes of the source lane's quad (AnsBackWriterL: 64-word rings, two workgroups per CU; AnsBackWriterS: 48-word
//    rings in 64-byte slots, three workgroups per CU, taken when that saves a round of workgroups; scl_ans_fast_io.h).
//  decode, per symbol (one 8-byte LDS table read, slot -> {f | sym << 24, slot - c}):
//    x  = (x >> m)*f + (slot - c)                          rans_base_decode_step (rANS.py:234-249)
//    nb = clz(x) - (32 - nsb);  x = (x << nb) | next nb bits   closed form of expand_state (:251-260),
//         done as one v_alignbit on a normalised copy.
//    The stream arrives as whole 128-byte lines through a per-lane LDS word ring (AnsBitReader); symbols leave
//    back to front, 128 per burst of eight 16-byte stores.
//
// Every lane only ever moves whole, aligned 128-byte lines of global memory (input symbols: eight 16-byte loads
// into registers, next line prefetched); DESIGN.md section 3.1 has the measurements that led there.
#include <type_traits>
#include <vector>

#include <stdlib.h>

#include "scl_ans_fast_io.h"
#include "scl_rans_internal.h"

#define RF_THREADS 256

__device__ __forceinline__ u32 rf_umulhi(u32 a, u32 b) { return __umulhi(a, b); }

// ---------------------------------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------------------------------
// Memory granularity (profiles/r01_v2_pmc_summary.txt): with one lane per 4 KiB chunk a CU owns a thousand open cache
// lines for input and as many for output -- more than L1 and, per XCD, the whole L2 -- so 16-byte lane accesses each
// became their own fabric request (3.6x / 2.3x traffic).  Every lane therefore moves whole lines:
//   input : 128 bytes (one line) per lane as 8 back-to-back 16-byte loads into registers, next line prefetched while
//           the current one is encoded (ping-pong register buffers).  (Loading the wave's 64 lines cooperatively, eight
//           lanes per line + an 8 x 8 register transpose, was built and measured: 3 % slower -- the read shape is not what
//           bounds the kernel, tools/ubench/linecopy3.hip.)
//   output: AnsBackWriterL (scl_ans_fast_io.h): bit window -> 64-word LDS ring per lane -> whole lines stored by quads.
// LDS per workgroup: 64 KiB ring + 4 KiB table = two workgroups per CU (2 waves per SIMD; 4 measured the same).
// Two writers (scl_ans_fast_io.h): AnsBackWriterL, 256-byte rings, two workgroups per CU; AnsBackWriterS, 192-byte
// rings in 64-byte slots, three workgroups per CU -- for batches that can populate a third wave per SIMD
// (rans_fast_encode_launch).
#define RF_RING_OFF 4096u  // the rings sit behind the 4 KiB symbol table in the workgroup's LDS block
typedef AnsBackWriterL<RF_THREADS> EncOutL;
typedef AnsBackWriterS<RF_THREADS> EncOutS;
typedef AnsBackWriterT<RF_THREADS> EncOutT;  // wave-striped slots, four workgroups per CU (round 6)

// v_mad_u32_u24 d, a, b, c: the compiler splits __umul24(a, b) + (c1 + c2) into v_mul_u32_u24 + v_add3_u32 (two
// half-rate instructions); one full-rate add feeding the multiply-add is cheaper
__device__ __forceinline__ u32 rf_mad24(u32 a, u32 b, u32 c) {
    u32 d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// One symbol.  Table entry {rcp, M-f, c, k_lo}.
// MSH = m - (32 - nsb) + 1 as before; q0 = mulhi(x, rcp) >> (MSH - 1) = floor(x / (f 2^k_lo)) is exact for EVERY
// x < 2^nsb (rans_fast_build_tables), so the test "x >= thresh = 2 RF f 2^k_lo" is simply q0 >= 2 RF: no threshold in
// the table, no subtraction -- pos = q0 >> (r + 1) (q0 < 4 RF), k = k_lo + pos, q = floor((x >> k) / f) = q0 >> pos
// (nested floors).  The k released bits are never extracted: push() takes the low k bits of x by itself.
struct __attribute__((aligned(16))) EncEntry {
    u32 rcp, mf, c, k_lo;  // all four words in use: the read stays a ds_read_b128 (a ds_read_b96 is far slower here)
};
template <int MSH_T, int R_T, typename EncOut>
__device__ __forceinline__ u32 rf_encode_entry(u32 &x, const EncEntry e, u32 msh_rt, EncOut &o) {  // returns k
    // run-time form: msh_rt = MSH | pre << 8 | r << 16.  Tables so small that m < 32 - nsb would need a negative MSH;
    // they shift x left by pre = (32 - nsb) - m first (x << pre < 2^(32 - m)) and use MSH = 1.
    // MSH_T = -1 (round 4): such a table with the pre-shift folded into its reciprocals -- mulhi(x << pre, rcp) = mulhi(x,
    // rcp << pre) when rcp << pre fits 32 bits, which it does for every table seen (rans_fast_build_tables checks) -- so
    // neither the pre-shift nor the quotient shift by MSH - 1 = 0 is executed: two instructions less per symbol than the
    // run-time form (tANS at its default RANGE_FACTOR = 1 is such a table).
    const u32 mh = rf_umulhi((MSH_T != 0) ? x : (x << ((msh_rt >> 8) & 0xFFu)), e.rcp);
    u32 q0, pos;
    if (MSH_T < 0) {
        q0 = mh;
        pos = mh >> ((msh_rt >> 16) + 1u);
    } else if (MSH_T) {
        static_assert(MSH_T <= 0 || MSH_T + R_T < 32, "shift out of range");
        q0 = mh >> (MSH_T - 1);
        pos = mh >> (MSH_T + R_T);  // = q0 >> (r + 1), straight from the product
    } else {
        q0 = mh >> ((msh_rt & 0xFFu) - 1u);
        pos = q0 >> ((msh_rt >> 16) + 1u);
    }
    const u32 k = e.k_lo + pos;  // a plain add: a byte of another word would cost an SDWA form, 1.75 instead of 1.05 ns
                // 2.5 ns where v_lshrrev + v_alignbit cost 2.8 and two issue slots: 0.5568 -> 0.5468 ms, nine alternations)
    const u64 t = ((((u64)x) << 32) | o.hi) >> k;
    o.hi = (u32)t;
    x = rf_mad24(q0 >> pos, e.mf, (u32)(t >> 32) + e.c);
    return k;
}

// NUM_BITS_OUT = b > 1 (round 4; b in {4, 8, 16}, at most 16 bits released per symbol).  shrink_state releases GROUPS of b
// bits: k_lo groups at x = L, one more from x = thresh on (scl_rans_fast_b.hip has the derivation), so
//   q0  = floor(x / (f 2^(b k_lo))) = mulhi(x, rcp) >> sh      exact for every x < 2^nsb with rcp = ceil(2^E / f),
//                                                              E = nsb + ceil(log2 f), sh = E + b k_lo - 32 -- PER SYMBOL:
//         k_lo is too coarse in b for one shift to serve every symbol, so the shift rides in the entry (a shift
//         instruction reads the low five bits of its operand: no extraction)
//   posb = b if q0 >= RF 2^b (one more group) else 0;  s = b k_lo + posb bits released;  q = q0 >> posb
//   x   = (x >> s) + c + q (M - f)
// Entry {rcp, M - f, c, sh | (b k_lo) << 8}; msh_rt = . | . | (r + b) << 16 | b << 24.
template <typename EncOut>
__device__ __forceinline__ u32 rf_encode_entry_b(u32 &x, const EncEntry e, u32 msh_rt, EncOut &o) {  // returns bits released
    const u32 q0 = rf_umulhi(x, e.rcp) >> (e.k_lo & 31u);      // low five bits of the word = sh (the mask costs nothing:
                                                               // v_lshrrev_b32 reads five bits, the backend drops it)
    const u32 posb = (q0 >> ((msh_rt >> 16) & 0xFFu)) ? (msh_rt >> 24) : 0u;
    const u32 k = (e.k_lo >> 8) + posb;
    const u64 t = ((((u64)x) << 32) | o.hi) >> k;  // k <= 16
    o.hi = (u32)t;
    x = rf_mad24(q0 >> posb, e.mf, (u32)(t >> 32) + e.c);
    return k;
}

struct Entries4 {
    EncEntry e[4];
    __device__ __forceinline__ void load(u32 w, const char *tab) {
        // byte 0 like the other three: one SDWA shift (the compiler turns (w & 0xFF) << 4 into a shift and a mask)
        u32 a0;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
            : "=v"(a0)
            : "v"(4u), "v"(w));
#define RF_LOAD_ENTRY(dst, addr) dst = *reinterpret_cast<const EncEntry *>(tab + (addr))
        RF_LOAD_ENTRY(e[0], a0);
        RF_LOAD_ENTRY(e[1], (w >> 4) & 0xFF0u);
        RF_LOAD_ENTRY(e[2], (w >> 12) & 0xFF0u);
        RF_LOAD_ENTRY(e[3], (w >> 20) & 0xFF0u);
    }
};

// 16 symbols (one 16-byte register).  The table entries of four symbols are fetched from LDS together (the state
// chain is serial, the table reads are not).
// CHECK_SYM: 0 = alphabet of 256 symbols, nothing to check; 1 / 2 = flag bytes above n = K - 1 for n <= 127 /
// n >= 128 with one SWAR test per four symbols: bit 7 of ((b & 0x7F) + c) says (b & 0x7F) > n (mod 128); OR-ed
// with b (n <= 127: bytes >= 128 are above n anyway) resp. AND-ed with b (n >= 128: only bytes >= 128 can be).
// `bad` collects those bits; chk_c = 0x01010101 * (127 - n) resp. 0x01010101 * (255 - n).
// `pre` holds the entries of the first four symbols on entry and those of `next_w` (the first word of the NEXT block)
// on exit: the table reads of a word are issued one word ahead of their use, in program order, so that their LDS latency
// (100+ clocks with the bank conflicts of a random symbol mix) runs under the arithmetic of the current word instead of
// in front of it -- with two waves per SIMD there is nobody else to hide it.
template <int CHECK_SYM, int MSH_T, int R_T, int NB_T = 1, typename EncOut>
__device__ __forceinline__ void rf_encode16(const uint4 v, u32 next_w, Entries4 &pre, u32 &x, EncOut &o, u32 &bad, u32 chk_c,
                                            char *lds, const char *tab, u32 msh_rt) {
    const u32 wv[5] = {v.x, v.y, v.z, v.w, next_w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const Entries4 cur = pre;
        pre.load(wv[d + 1], tab);
        asm volatile("" ::: "memory");  // keep the reads here: the compiler would sink them to their first use
        // the four entries of this word were issued back to back a whole word ago: touching the LAST of them first makes the
        // compiler wait once (lgkmcnt(4): only the reads just issued may still be in flight) instead of once per symbol
        asm volatile("" : : "v"(cur.e[3].k_lo));
        if (CHECK_SYM) {
            const u32 w = wv[d];
            const u32 t = (w & 0x7F7F7F7Fu) + chk_c;
            bad |= (CHECK_SYM == 1) ? (t | w) : (t & w);
        }
        // (a pair releases at most 26 bits for NUM_BITS_OUT = 1 and at most 32 for b > 1: the window holds 32 + 32)
        // the window's low word is brought up to date once per pair (NUM_BITS_OUT = 1: a pair releases < 32 bits) -- the
        // hi of before the pair funnel-shifted by k0 + k1 is what the two pushes moved; b > 1 (up to 32 bits) folds per symbol
        const u32 h0 = o.hi;
        const u32 k0 = NB_T == 1 ? rf_encode_entry<MSH_T, R_T>(x, cur.e[0], msh_rt, o) : rf_encode_entry_b(x, cur.e[0], msh_rt, o);
        const u32 h1 = o.hi;
        if (NB_T != 1) o.fold_lo(h0, k0);
        const u32 k1 = NB_T == 1 ? rf_encode_entry<MSH_T, R_T>(x, cur.e[1], msh_rt, o) : rf_encode_entry_b(x, cur.e[1], msh_rt, o);
        if (NB_T != 1) o.fold_lo(h1, k1); else o.fold_lo(h0, k0 + k1);
        o.template check<RF_RING_OFF>(lds, k0 + k1);
        const u32 h2 = o.hi;
        const u32 k2 = NB_T == 1 ? rf_encode_entry<MSH_T, R_T>(x, cur.e[2], msh_rt, o) : rf_encode_entry_b(x, cur.e[2], msh_rt, o);
        const u32 h3 = o.hi;
        if (NB_T != 1) o.fold_lo(h2, k2);
        const u32 k3 = NB_T == 1 ? rf_encode_entry<MSH_T, R_T>(x, cur.e[3], msh_rt, o) : rf_encode_entry_b(x, cur.e[3], msh_rt, o);
        if (NB_T != 1) o.fold_lo(h3, k3); else o.fold_lo(h2, k2 + k3);
        o.template check<RF_RING_OFF>(lds, k2 + k3);
    }
}

template <typename EncOut, int CHECK_SYM, int MSH_T, int R_T, int NB_T = 1>
// (alphabets below 256 symbols carry the SWAR symbol check: a few registers more than the 128 that four workgroups per CU
// leave -- the striped writer then runs three)
__global__ void __launch_bounds__(RF_THREADS, (EncOut::STRIPED && CHECK_SYM) ? 3 : EncOut::WG_PER_CU) rans_encode_fast_kernel(RansFastDev P, const u8 *__restrict__ sym,
                                                                        u64 sym_stride,
                                                                        const u32 *__restrict__ lens, u32 chunk_len,
                                                                        u64 n_chunks, u8 *__restrict__ out,
                                                                        u64 out_stride, u64 *__restrict__ out_bit_off,
                                                                        u32 *__restrict__ out_nbits,
                                                                        u32 *__restrict__ status) {
    // one LDS block: the 4 KiB symbol table first (its offsets then fit the 16-bit offset field of the DS instructions;
    // behind the rings, every table address cost an extra VALU instruction), then 64 KiB of word rings
    __shared__ __attribute__((aligned(16))) char s_lds[RF_RING_OFF + EncOut::RING_BYTES];
    char *lds = s_lds + RF_RING_OFF;
    const char *tab = s_lds;
    static_assert(RF_RING_OFF == 256 * sizeof(uint4), "the rings start right behind the symbol table");
    reinterpret_cast<uint4 *>(s_lds)[threadIdx.x & 255] = P.d_enc_tab[threadIdx.x & 255];
    __syncthreads();
    const u64 c = (u64)blockIdx.x * RF_THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 n = lens ? lens[c] : chunk_len;
    const u8 *src = sym + c * sym_stride;
    const u32 msh_rt = P.enc_msh;  // MSH | pre << 8 | r << 16, see rans_fast_build_tables
    // offsets from the workgroup's first slot (the launch guarantees RF_THREADS * out_stride < 2^32): the store address
    // is then a uniform base + one 32-bit register, which is also all a helper lane needs to know of its source
    u8 *wg_out = out + (u64)blockIdx.x * RF_THREADS * out_stride;
    EncOut o;
    if constexpr (EncOut::STRIPED)
        o.init(threadIdx.x, (u32)out_stride);  // the 64 slots of a wave interleaved in 16-byte pieces (AnsBackWriterT)
    else
        o.init(threadIdx.x, (threadIdx.x + 1) * (u32)out_stride);
    // whole wave, equally long chunks: every lane reaches every flush point, so the quads can store cooperatively
    const bool coop_out = __builtin_amdgcn_ballot_w64(n == (u32)__builtin_amdgcn_readfirstlane((int)n)) == ~0ull;
#define RF_FLUSH()                                   \
    do {                                             \
        if constexpr (EncOut::STRIPED) {             \
            o.flush(lds, wg_out, NB_T == 1 ? 72u : 60u); \
        } else {                                     \
            if (coop_out)                            \
                o.flush_quad(lds, wg_out, threadIdx.x); \
            else                                     \
                o.flush_lane(lds, wg_out);           \
        }                                            \
    } while (0)
    u32 x = P.L;
    u32 bad = 0;
    const u32 chk_c = 0x01010101u * ((CHECK_SYM == 2 ? 255u : 127u) - (P.K - 1));  // unused when K = 256

    // One 128-byte line per tile.
    const u32 n_lines = n >> 7;
    const uint4 *src16 = reinterpret_cast<const uint4 *>(src);
    Line128 cur, nxt;
    Entries4 pre;
    if (n_lines) {
        cur.load(src16);
        pre.load(cur.v[0].x, tab);
    }
    // one line = eight blocks of 16 symbols, straight-line code: an inner loop holding only stores would make the compiler
    // drain vmcnt in its preheader (SIInsertWaitcnts::shouldFlushVmCnt), i.e. wait for the prefetch at once.
    // Flush points every 64 symbols (<= 26 new words on top of <= 31 pending, ring of 64) -- after the symbols 32 and 96 of
    // the line, NOT 64 and 128: the wait for the prefetched line at the end of the iteration is an s_waitcnt vmcnt(0)
    // (loads and stores share one in-order counter and the stores sit in conditional code, so the compiler cannot count
    // them), i.e. it also waits for every store issued so far to COMPLETE; stores issued just before it cost their whole
    // round trip.  (The first word of the next line is only known once the prefetch has landed: its entries are read after
    // the line.)
#define RF_ENCODE_LINE(L)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                    \
        rf_encode16<CHECK_SYM, MSH_T, R_T, NB_T>(L.v[i], i < 7 ? L.v[i + 1].x : 0u, pre, x, o, bad, chk_c, lds, tab, msh_rt); \
        if ((i & EncOut::FLUSH_MASK) == EncOut::FLUSH_PHASE) RF_FLUSH();                                               \
    }
    // prefetch the next line while this one is encoded; unconditional (the last line is simply loaded again): a load
    // under a lane-dependent condition is merged with the old value, i.e. waited for, at once
#define RF_LOAD_LINE(L, T) L.load(src16 + 8 * min((T), n_lines - 1))
    u32 t = 0;
#pragma nounroll
    for (; t < n_lines; ++t) {
        RF_LOAD_LINE(nxt, t + 1);
        RF_ENCODE_LINE(cur)
        cur = nxt;
        pre.load(cur.v[0].x, tab);
    }
#undef RF_ENCODE_LINE
#undef RF_LOAD_LINE
    u32 i = n_lines << 7;
    for (; i + 16 <= n; i += 16) {  // ragged tail: whole 16-byte blocks, then single symbols
        const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
        pre.load(v.x, tab);
        rf_encode16<CHECK_SYM, MSH_T, R_T, NB_T>(v, 0u, pre, x, o, bad, chk_c, lds, tab, msh_rt);
        RF_FLUSH();
    }
    for (; i < n; ++i) {
        const u32 a = (u32)src[i] << 4;
        if (CHECK_SYM && (a >> 4) >= P.K) bad |= 0x80u;
        const EncEntry e1 = *reinterpret_cast<const EncEntry *>(tab + a);
        const u32 h_before = o.hi;
        const u32 k_one = NB_T == 1 ? rf_encode_entry<MSH_T, R_T>(x, e1, msh_rt, o) : rf_encode_entry_b(x, e1, msh_rt, o);
        o.fold_lo(h_before, k_one);
        o.template check<RF_RING_OFF>(lds, k_one);
        if ((i & 15u) == 15u) RF_FLUSH();
    }
    RF_FLUSH();
#undef RF_FLUSH
    o.template put32<RF_RING_OFF>(lds, x, P.nsb);
    u32 st = (CHECK_SYM && (bad & 0x80808080u)) ? SCL_ST_SYMBOL : 0u;
    if (P.size_bits < 32 && (n >> P.size_bits)) st |= SCL_ST_SIZE;
    o.template put32<RF_RING_OFF>(lds, n, P.size_bits);
    const u64 total = o.finish(lds, wg_out);
    out_bit_off[c] = (c + 1) * out_stride * 8 - total;
    out_nbits[c] = (u32)total;
    if (status) status[c] = st;
}

// ---------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------
// Same line-granular rule as the encoder:
//   input : the lane's stream arrives in 64-byte blocks (4 back-to-back 16-byte loads, one block prefetched
//           in registers), is byte-swapped once and parked in a per-lane LDS ring of 32 words ([w][t] layout,
//           conflict-free); the bit window refills one word at a time from the ring;
//   output: 64 decoded symbols are assembled in 4 registers and leave as one 64-byte sector
//           (4 back-to-back 16-byte stores), end of the chunk first.
// One workgroup of 1024 lanes per CU: [0, 128 KiB) word ring, [128 KiB, 160 KiB) slot table of 8-byte entries
// {f | sym << 24, slot - c}: v_mad_u32_u24 ignores the symbol byte, so no field needs extracting.
// Workgroup size: 1024 lanes (one workgroup per CU, 4 waves per SIMD) for batches that fill the chip that way
// (>= 256 workgroups); smaller batches take 256-lane workgroups so that they spread over all CUs instead of running 4
// waves per SIMD on a quarter of them (65 536 chunks: decode 0.48 -> see DESIGN.md).
#define RD_THREADS 1024
#define RD_THREADS_SMALL 256

// decode one symbol: state update + renormalisation from the 32-bit lookahead `lk` (bits are consumed from
// its top); returns the first table word (symbol in byte 3) and the number of bits used.
// ML / CB: compile-time m_log2 and 32 - nsb when non-zero, else the run-time values.
// With compile-time constants and CB == 3 the state is carried TOP-ALIGNED (X = x << 3, what the renormalising
// v_alignbit produces anyway): the table offset 8 * (x mod M) is X & (8 * (M - 1)) -- one v_and_or with the table
// base instead of shift + v_and_or -- and the "x = y >> CB" step disappears: two instructions less per symbol.
struct RfGenM {  // run-time constants of the any-total decoder
    double inv_m;
    u32 m, l;
};
// `lk` is updated to the lookahead that remains after the symbol (bits are consumed from its top).
template <int ML_T, int CB_T, int NB_T = 1>
__device__ __forceinline__ u32 rf_decode_symbol(u32 &x, u32 &lk, u32 &used, const char *tab, u32 ml_rt, u32 cb_rt,
                                                const RfGenM &rf_gen) {
    if (NB_T != 1) {
        // NUM_BITS_OUT = b in {4, 8, 16} (round 4): expand_state (rANS.py:251-260) reads GROUPS of b bits while x < L.  The
        // state is carried as X = x << 3 | three don't-care bits, like the literal b = 1 form below (nsb <= 29); d =
        // bit_width(L) - bit_width(xn) = clz(xn) - cbl bits are missing (d >= 1 - b: xn < 2^nsb), i.e. sh = (d + b - 1) &
        // ~(b - 1) bits are read, and the 64-bit shift that renormalises AND re-aligns is by sh + 3 = (that) | 3 (b >= 4:
        // the low two bits of sh are zero): v_add + v_and_or.  rf_gen.m = b - 1 - cbl, rf_gen.l = ~(b - 1), ml_rt = m.
        const uint2 e = *reinterpret_cast<const uint2 *>(tab + (x & (((1u << ml_rt) - 1u) << 3)));
        const u32 xn = __umul24(x >> (ml_rt + 3), e.x) + e.y;
        const u32 cl = (((u32)__builtin_clz(xn) + rf_gen.m) & rf_gen.l) | 3u;
        const u64 t = ((((u64)xn) << 32) | lk) << cl;
        x = (u32)(t >> 32);
        lk = __builtin_amdgcn_alignbit(x, (u32)t, 3);
        used = cl;
        return e.x;
    }
    if (ML_T > 0 && CB_T == 3) {
        // Top-aligned state X = x << 3 | three bits of lookahead (don't care): one 64-bit shift of xn:lookahead by
        // cl = clz(xn) renormalises; its high word is the new X, its low word the lookahead shifted by cl -- three bits
        // more than were read (cl - 3) -- so the lookahead that REMAINS is alignbit(high, low, 3): one instruction
        // instead of computing cl - 3 and shifting again.  `used` is cl here (the caller subtracts the 3 per symbol once
        // per pair).
        const uint2 e = *reinterpret_cast<const uint2 *>(tab + (x & (((1u << ML_T) - 1u) << 3)));
        const u32 xn = __umul24(x >> (ML_T + 3), e.x) + e.y;
        const u32 cl = (u32)__builtin_clz(xn);
        const u64 t = ((((u64)xn) << 32) | lk) << cl;  // v_lshlrev_b64
        x = (u32)(t >> 32);
        lk = __builtin_amdgcn_alignbit(x, (u32)t, 3);
        used = cl;
        return e.x;
    }
    if (ML_T == 0 && CB_T == 3) {
        // The same with run-time constants, for every power-of-two total and every RANGE_FACTOR with NUM_STATE_BITS <= 29
        // (tANS at its default RANGE_FACTOR = 1 among them): the state is carried as X = x << 3 (not top-aligned any more,
        // but the table offset is still X & (8 (M - 1))), and the renormalising shift that also re-aligns is by
        // (clz(xn) - cbl) + 3 -- cb_rt carries 3 - cbl here: one v_add on top of the literal form above, two instructions
        // less than the form below.
        const uint2 e = *reinterpret_cast<const uint2 *>(tab + (x & (((1u << ml_rt) - 1u) << 3)));
        const u32 xn = __umul24(x >> (ml_rt + 3), e.x) + e.y;
        const u32 cl = (u32)__builtin_clz(xn) + cb_rt;  // 3 <= cl <= 31
        const u64 t = ((((u64)xn) << 32) | lk) << cl;
        x = (u32)(t >> 32);
        lk = __builtin_amdgcn_alignbit(x, (u32)t, 3);
        used = cl;
        return e.x;
    }
    if (ML_T < 0) {
        // any total M <= 4096 (ml_rt carries nothing here; the run-time values come through rf_gen):
        //   x // M exactly as trunc((x + 0.5) * (1 / M)) in binary64 (x < 2^31, M <= 2^12: the error 2^-20 is far
        //   below the distance 0.5 / M of (x + 0.5) / M from an integer), slot = x - q M;
        //   expand_state: L = RANGE_FACTOR * M has NUM_STATE_BITS - 1 bits, so the renormalised state has either
        //   that width (if it is already >= L) or one bit more: one comparison picks between the two.
        const u32 qd = (u32)(((double)x + 0.5) * rf_gen.inv_m);
        const u32 slot = x - qd * rf_gen.m;
        const uint2 e = *reinterpret_cast<const uint2 *>(tab + slot * 8);
        const u32 xn = __umul24(qd, e.x) + e.y;
        const u32 cl = (u32)__builtin_clz(xn);
        const u32 y = __builtin_amdgcn_alignbit(xn, lk, 32 - cl);
        const u32 lt = ((y >> (cb_rt + 1)) - rf_gen.l) >> 31;  // 1 iff the narrower candidate is below L
        x = y >> (cb_rt + 1 - lt);
        used = cl - cb_rt - 1 + lt;
        lk <<= used;
        return e.x;
    }
    const u32 ML = ML_T > 0 ? (u32)ML_T : ml_rt, CB = ML_T > 0 ? (u32)CB_T : cb_rt;
    const uint2 e = *reinterpret_cast<const uint2 *>(tab + ((x << 3) & (((1u << ML) - 1u) << 3)));  // v_and_or with the table base
    x = __umul24(x >> ML, e.x) + e.y;                         // v_mad_u32_u24 reads the low 24 bits (f) of e.x
    const u32 cl = (u32)__builtin_clz(x);                     // x >= 2^r > 0
    const u32 y = __builtin_amdgcn_alignbit(x, lk, 32 - cl);  // (x << cl) | (lk >> (32 - cl)), cl in [1,31]
    x = y >> CB;                                              // keep nb = cl - CB new bits
    used = cl - CB;
    lk <<= used;
    return e.x;
}

// 16 symbols, last first, into one 16-byte register (byte i of the result = symbol i of the block)
// REFILL: top the input ring up afterwards.  The ring wants that at least every 32 symbols (32 x 13 bits = 13 words:
// a check that found 17 words ahead and did nothing still leaves 4); the line loop asks after every second block,
// because a check costs the wave BOTH refill bodies (some lane is always at either stage: 2 x 16 byte swaps and ring
// writes, plus the loads) whether or not a given lane needed one.
template <int ML_T, int CB_T, bool REFILL = true, int NB_T = 1, typename DecIn>
__device__ __forceinline__ uint4 rf_decode16(u32 &x, DecIn &r, char *lds, const char *tab, u32 ml_rt, u32 cb_rt,
                                             const RfGenM &rf_gen) {
    u32 ow[4];
#pragma unroll
    for (int d = 3; d >= 0; --d) {
        u32 pr[2] = {0u, 0u};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32 lk = r.look(lds);
            u32 ua, ub;
            const u32 ea = rf_decode_symbol<ML_T, CB_T, NB_T>(x, lk, ua, tab, ml_rt, cb_rt, rf_gen);
            const u32 eb = rf_decode_symbol<ML_T, CB_T, NB_T>(x, lk, ub, tab, ml_rt, cb_rt, rf_gen);
            // two symbols use at most 2*m <= 24 bits of the 32-bit lookahead (2 x 16 for NUM_BITS_OUT > 1: the second
            // symbol's three don't-care bits are whatever follows); the top-aligned variants report their shift, 3 more
            // per symbol than they read
            r.advance(lds, ((ML_T >= 0 && CB_T == 3) || NB_T != 1) ? ua + ub - 6u : ua + ub);
            // the symbol bytes of a pair with ONE v_perm (the earlier symbol is the more significant byte), the two pairs
            // of a word with another: three instead of four per four symbols
            u32 p = __builtin_amdgcn_perm(ea, eb, 0x0c0c0703u);  // 0 : 0 : ea >> 24 : eb >> 24
            // the chain through x is serial anyway; without this fence the compiler sinks all the byte inserts
            // of a 64-symbol iteration to its end, keeps every table word alive until then and spills
            asm volatile("" : "+v"(p) : : "memory");
            pr[h] = p;
        }
        ow[d] = __builtin_amdgcn_perm(pr[0], pr[1], 0x05040100u);
    }
    if (REFILL) r.maybe_refill(lds);
    return make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

// STRIPED: the input is in wave-striped slots (AnsBackWriterT / AnsBitReaderT, scl_ans_fast_io.h); `in_size_bytes` then
// carries the slot stride and stream c lies in slot c.
template <int ML_T, int CB_T, int THREADS, int NB_T = 1, bool STRIPED = false>
__global__ void __launch_bounds__(THREADS) rans_decode_fast_kernel(RansFastDev P, const u8 *__restrict__ in,
                                                                     u64 in_size_bytes,
                                                                     const u64 *__restrict__ bit_off,
                                                                     const u32 *__restrict__ in_nbits, u64 n_chunks,
                                                                     u8 *__restrict__ out_sym, u64 out_stride,
                                                                     u32 out_cap, u32 *__restrict__ out_lens,
                                                                     u32 *__restrict__ consumed,
                                                                     u32 *__restrict__ status) {
    // slot table first: its offsets (< 32 KiB) then need no base added (a DS offset field reaches 64 KiB); the ring
    // works on addresses relative to its own base, which the DS offset field supplies
    typedef typename std::conditional<STRIPED, AnsBitReaderT<THREADS>, AnsBitReaderW<THREADS>>::type DecIn;
    __shared__ __attribute__((aligned(16))) char s_lds[4096 * 8 + DecIn::RING_BYTES];
    char *lds = s_lds + 4096 * 8;
    const char *tab = s_lds;
    const u32 M = P.M;
    constexpr u32 XSH = ((ML_T >= 0 && CB_T == 3) || NB_T != 1) ? 3u : 0u;  // top-aligned state, see rf_decode_symbol
    for (u32 i = threadIdx.x; i < M; i += THREADS) {
        const uint2 v = P.d_dec_tab[i];
        reinterpret_cast<uint2 *>(s_lds)[i] = v;
    }
    __syncthreads();
    const u64 c = (u64)blockIdx.x * THREADS + threadIdx.x;
    if (c >= n_chunks) return;
    const u32 avail = in_nbits[c];
    u32 st = 0;
    if (avail < P.size_bits + P.nsb) {  // header does not fit
        out_lens[c] = 0;
        consumed[c] = P.size_bits + P.nsb;
        if (status) status[c] = SCL_ST_TRUNCATED;
        return;
    }
    DecIn r;
    if constexpr (STRIPED)
        r.init(in, in_size_bytes, c, bit_off[c], lds, threadIdx.x);
    else
        r.init(in, in_size_bytes, bit_off[c], lds, threadIdx.x);
    u32 n = r.get(lds, P.size_bits);
    u32 x = r.get(lds, P.nsb);
    x <<= XSH;
    out_lens[c] = n;
    if (n > out_cap) {
        st |= SCL_ST_CAPACITY;
        n = 0;
    }
    const u32 st_header = st;
    const u32 ml_rt = P.m_log2, cb_rt = (ML_T == 0 && CB_T == 3) ? 3u - (32 - P.nsb) : 32 - P.nsb;  // see rf_decode_symbol
    RfGenM rf_gen;
    rf_gen.inv_m = 1.0 / (double)P.M;
    rf_gen.m = NB_T != 1 ? P.dec_sadd : P.M;   // NUM_BITS_OUT > 1: b - 1 - cbl and ~(b - 1), see rf_decode_symbol
    rf_gen.l = NB_T != 1 ? P.dec_notb : P.L;
    u8 *dst = out_sym + c * out_stride;

    // symbols come out last-first (rANS.py:291): the ragged head of the last 16-byte block ...
    u32 i = n;
    while (i & 15u) {
        u32 used, lk1 = r.look(lds);
        const u32 e = rf_decode_symbol<ML_T, CB_T, NB_T>(x, lk1, used, tab, ml_rt, cb_rt, rf_gen);
        r.advance(lds, used - XSH);
        dst[--i] = (u8)(e >> 24);
        if ((i & 3u) == 0) r.maybe_refill(lds);
    }
    // ... whole 16-byte blocks up to a line boundary ...
    while (i & 127u) {
        const uint4 v = rf_decode16<ML_T, CB_T, true, NB_T>(x, r, lds, tab, ml_rt, cb_rt, rf_gen);
        i -= 16;
        *reinterpret_cast<uint4 *>(dst + i) = v;
    }
    // ... then one full 128-byte line per iteration, eight registers: stored cooperatively by the wave when every
    // lane is here with the same position (CoopLineStore, scl_ans_fast_io.h), else as a burst of eight 16-byte stores
    CoopLineStore cs;
    cs.init(out_sym, c, out_stride, i);
#pragma nounroll
    while (i) {
        uint4 a[8];
#pragma unroll
        for (int b = 7; b >= 0; --b)
            // ring check every 32 symbols (<= 13 bits each), every 16 for NUM_BITS_OUT > 1 (<= 16 bits each)
            a[b] = ((b & 1) && NB_T == 1) ? rf_decode16<ML_T, CB_T, false, NB_T>(x, r, lds, tab, ml_rt, cb_rt, rf_gen)
                                          : rf_decode16<ML_T, CB_T, true, NB_T>(x, r, lds, tab, ml_rt, cb_rt, rf_gen);
        i -= 128;
        if (cs.on) {
            cs.store(a, i);
        } else {
            uint4 *p = reinterpret_cast<uint4 *>(dst + i);
#pragma unroll
            for (int b = 0; b < 8; ++b) p[b] = a[b];
        }
    }
    const u32 used_bits = r.consumed();
    if (used_bits > avail) st |= SCL_ST_TRUNCATED;
    else if (st_header == 0 && (x >> XSH) != P.L) st |= SCL_ST_STATE;  // assert state == INITIAL_STATE (rANS.py:295)
    consumed[c] = used_bits;
    if (status) status[c] = st;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static u32 ceil_log2_u32(u32 v) {
    u32 s = 0;
    while ((1ull << s) < v) ++s;
    return s;
}

// Decides whether the model qualifies and uploads the two tables.  Returns SCL_OK also when the model
// simply does not qualify (m->fast stays 0).
// NUM_BITS_OUT = b in {4, 8, 16} on the same kernels (round 4).  Qualifies: total a power of two <= 4096, RANGE_FACTOR =
// 2^r, NUM_STATE_BITS = r + m + b <= 29 (state carried as x << 3), r + b <= 24 (quotients fit v_mad_u32_u24), at most 16
// bits released per symbol (a pair of symbols fits the encoder's 32 + 32-bit window; 64 symbols fit its ring).
static int rans_fast_build_tables_b(scl_rans_model *m, const u32 *h_freq, const u32 *h_cum) {
    const RansDev &D = m->dev;
    const u32 b = D.b;
    if (b != 4 && b != 8 && b != 16) return SCL_OK;
    if (D.m_log2 == 0xFFFFFFFFu || D.M < 2 || D.M > 4096 || D.K < 2) return SCL_OK;
    if ((D.RF & (D.RF - 1)) != 0 || D.nsb > 29 || m->max_bits_per_symbol > 16) return SCL_OK;
    u32 r = 0;
    while ((1ull << r) < D.RF) ++r;
    if (r + b > 24 || D.nsb != r + D.m_log2 + b) return SCL_OK;
    const u32 M = (u32)D.M, nsb = D.nsb;
    std::vector<uint4> enc(256);
    std::vector<uint2> dec(M);
    for (u32 s = 0; s < 256; ++s) {
        const u32 src = s < D.K ? s : 0;  // out-of-alphabet symbols are flagged, entry 0 keeps the lane sane
        const u32 f = h_freq[src], c = h_cum[src];
        const u64 a1 = ((u64)D.RF * f) << b;  // max_shrunk_state + 1 (rANS.py:112)
        u32 k_lo = 0, k_hi = 0;
        while ((D.L >> (k_lo * b)) >= a1) ++k_lo;
        while ((m->H >> (k_hi * b)) >= a1) ++k_hi;
        if (k_hi > k_lo + 1 || (k_lo + 1) * b > 16) return SCL_OK;
        // floor(x / (f 2^(b k_lo))) = (x rcp) >> (E + b k_lo) for every x < 2^nsb: rcp = ceil(2^E / f) with E = nsb +
        // ceil(log2 f) (the error x (rcp f - 2^E) / (f 2^(E + b k_lo)) stays below 1 / (f 2^(b k_lo)))
        const u32 E = nsb + ceil_log2_u32(f);
        const u64 rcp = ((1ull << E) + f - 1) / f;
        if ((rcp >> 32) != 0 || E + b * k_lo < 32 || E + b * k_lo - 32 > 31) return SCL_OK;
        enc[s] = make_uint4((u32)rcp, M - f, c, (E + b * k_lo - 32) | ((b * k_lo) << 8));
    }
    for (u32 s = 0; s < D.K; ++s)
        for (u32 j = 0; j < h_freq[s]; ++j) dec[h_cum[s] + j] = make_uint2(h_freq[s] | (s << 24), j);
    hipError_t e = hipMalloc((void **)&m->d_enc_tab, 256 * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_dec_tab, M * sizeof(uint2));
    if (e == hipSuccess) e = hipMemcpy(m->d_enc_tab, enc.data(), 256 * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_dec_tab, dec.data(), M * sizeof(uint2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("rans_model_create: fast-path table upload failed: %s", hipGetErrorString(e));
        return SCL_E_HIP;
    }
    m->enc_lockstep = 0;
    m->fdev.K = D.K;
    m->fdev.nsb = nsb;
    m->fdev.size_bits = D.size_bits;
    m->fdev.m_log2 = D.m_log2;
    m->fdev.L = (u32)D.L;
    m->fdev.M = M;
    m->fdev.enc_msh = ((r + b) << 16) | (b << 24);
    m->fdev.b = b;
    const u32 cbl = 32 - scl_bit_width_u64(D.L);
    m->fdev.dec_sadd = b - 1 - cbl;  // (mod 2^32: added to a count of leading zeros >= cbl - b + 1)
    m->fdev.dec_notb = ~(b - 1);
    m->fdev.d_enc_tab = m->d_enc_tab;
    m->fdev.d_dec_tab = m->d_dec_tab;
    m->fast = 1;
    return SCL_OK;
}

int rans_fast_build_tables(scl_rans_model *m, const u32 *h_freq, const u32 *h_cum) {
    const RansDev &D = m->dev;
    m->fast = 0;
    m->fdev.b = 1;
    m->fdev.enc_folded = 0;
    m->fdev.dec_sadd = m->fdev.dec_notb = 0;
    if (D.b > 1) return rans_fast_build_tables_b(m, h_freq, h_cum);
    // any total 2 <= M <= 4096 (a power of two or not), NUM_BITS_OUT = 1, RANGE_FACTOR = 2^r <= 2^23, H < 2^31
    if (D.b != 1 || D.M < 2 || D.M > 4096) return SCL_OK;
    if ((D.RF & (D.RF - 1)) != 0 || D.RF > (1u << 23) || D.nsb > 30 || D.K < 2) return SCL_OK;
    if (m->max_bits_per_symbol > 13) return SCL_OK;
    const u32 M = (u32)D.M, nsb = D.nsb;
    u32 r = 0;
    while ((1ull << r) < D.RF) ++r;
    const u32 mp = ceil_log2_u32(M);
    if (nsb != r + 1 + mp) return SCL_OK;  // cannot happen: bit_width(2 RF M - 1)
    // Exact division by f through one 32-bit multiply-high for every symbol and BOTH shift counts k in {k1-1, k1}:
    //   floor((x >> k) / f) = (x * rcp) >> (E + k),  rcp = ceil(2^E / f),  needs E >= nsb + ceil(log2 f);
    //   E = C - k1 with C = r + 2 mp + 2 makes E + k = C - [x < thresh] the same for all symbols (k1 + ceil(log2 f)
    //   <= mp + 1), so the shift after the multiply-high is MSH - [x < thresh] with MSH = C - 32 -- or, when C <= 32,
    //   MSH = 1 after x was shifted left by pre = 33 - C.  For a power-of-two total this is rcp = ceil(2^(nsb+s)/f).
    const u32 C = r + 2 * mp + 2;
    const u32 enc_msh = ((C >= 33) ? (C - 32) : (1u | ((33 - C) << 8))) | (r << 16);
    std::vector<uint4> enc(256);
    std::vector<uint2> dec(M);
    bool foldable = true;
    u64 fixed_k_mass[32] = {0};  // total frequency of the symbols that ALWAYS release k bits (k_hi == k_lo), per k
    for (u32 s = 0; s < 256; ++s) {
        const u32 src = s < D.K ? s : 0;  // out-of-alphabet symbols are flagged, entry 0 keeps the lane sane
        const u32 f = h_freq[src], c = h_cum[src];
        const u64 a1 = 2ull * D.RF * f;  // max_shrunk_state + 1 (rANS.py:112)
        // shrink_state (rANS.py:149-161) shifts x in [L, 2L) right by the smallest k with (x >> k) < a1: that is
        // k_lo for x = L and at most k_lo + 1 for x = 2L - 1, the switch being at thresh = a1 << k_lo
        u32 k_lo = 0, k_hi = 0;
        while ((D.L >> k_lo) >= a1) ++k_lo;
        while (((2 * D.L - 1) >> k_hi) >= a1) ++k_hi;
        const u64 thresh = a1 << k_lo;
        const u32 k1 = k_lo + 1;
        if (k_hi > k1 || thresh > (1ull << 31) || C < k1 + 1) return SCL_OK;
        const u32 E = C - k1;
        if (E < nsb + ceil_log2_u32(f) || E > 62) return SCL_OK;  // cannot happen (see above)
        const u64 rcp = ((1ull << E) + f - 1) / f;
        if (rcp >> 32) return SCL_OK;
        if (C < 33 && ((rcp << (33 - C)) >> 32) != 0) foldable = false;
        enc[s] = make_uint4((u32)rcp, M - f, c, k_lo);
        if (s < D.K && k_hi == k_lo && k_lo < 32) fixed_k_mass[k_lo] += f;
    }
    // small tables (C <= 32: MSH = 1 after a pre-shift of x): fold the pre-shift into the reciprocals when they all fit
    u32 enc_msh_final = enc_msh;
    m->fdev.enc_folded = 0;
    if (C < 33 && foldable) {
        for (u32 s = 0; s < 256; ++s) enc[s].x <<= (33 - C);
        enc_msh_final = 1u | (r << 16);
        m->fdev.enc_folded = 1;
    }
    // lanes whose symbols nearly all release the same number of bits complete their words at the same steps
    // (a table of equal frequencies does so exactly): AnsBackWriterS has no defence against that, AnsBackWriterL has
    u64 top_mass = 0;
    for (u32 k = 0; k < 32; ++k) top_mass = fixed_k_mass[k] > top_mass ? fixed_k_mass[k] : top_mass;
    m->enc_lockstep = (4 * top_mass >= 3 * (u64)M) ? 1u : 0u;
    for (u32 s = 0; s < D.K; ++s)
        for (u32 j = 0; j < h_freq[s]; ++j) dec[h_cum[s] + j] = make_uint2(h_freq[s] | (s << 24), j);
    hipError_t e = hipMalloc((void **)&m->d_enc_tab, 256 * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_dec_tab, M * sizeof(uint2));
    if (e == hipSuccess) e = hipMemcpy(m->d_enc_tab, enc.data(), 256 * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_dec_tab, dec.data(), M * sizeof(uint2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        scl_set_error("rans_model_create: fast-path table upload failed: %s", hipGetErrorString(e));
        return SCL_E_HIP;
    }
    m->fdev.K = D.K;
    m->fdev.nsb = nsb;
    m->fdev.size_bits = D.size_bits;
    m->fdev.m_log2 = D.m_log2;
    m->fdev.L = (u32)D.L;
    m->fdev.M = M;
    m->fdev.enc_msh = enc_msh_final;
    m->fdev.d_enc_tab = m->d_enc_tab;
    m->fdev.d_dec_tab = m->d_dec_tab;
    m->fast = 1;
    return SCL_OK;
}

// Which writer.  At equal occupancy the two run within noise of each other (the third wave per SIMD that AnsBackWriterS
// makes room for is worth ~5 %, its longer word-emission block and doubled flush points cost as much), so the choice is
// a matter of ROUNDS: a CU holds two workgroups of the L kernel (a round = 512 workgroups on this 256-CU chip, t) or
// three of the S kernel (768 workgroups, ~1.5 t).  S is taken when it needs strictly less time by that count --
// e.g. 513..768 workgroups (131 073..196 608 chunks: 0.467 -> 0.427 ms) -- and never for tables that drive the lanes in
// lockstep.  1 024 workgroups (the headline batch) are two full rounds of L.  SCL_RANS_ENC_WRITER=L|S overrides (tests,
// tools/ab_s.sh).
bool rf_use_slot_writer(const scl_rans_model *m, u64 n_chunks) {
    const char *force = getenv("SCL_RANS_ENC_WRITER");  // read at every call: a test may switch it inside one process
    if (force && (force[0] == 'L' || force[0] == 'l')) return false;
    if (force && (force[0] == 'S' || force[0] == 's')) return true;
    if (m->enc_lockstep) return false;
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    const u64 w = (n_chunks + RF_THREADS - 1) / RF_THREADS;
    const u64 rounds_l = (w + 2 * cus - 1) / (2 * cus), rounds_s = (w + 3 * cus - 1) / (3 * cus);
    return 3 * rounds_s < 2 * rounds_l;
}

// What a batch of n_chunks runs: ONE place decides (the launch below switches on it, rans_fast_kernel_names prints it the
// way rocprofv3 prints the instantiation -- bench.py's "kernel" fields come from there, so they can be matched
// mechanically against profiles/*_kernel_trace_summary.txt).
struct RfEncChoice {
    bool slots;  // AnsBackWriterS (three workgroups per CU) instead of AnsBackWriterL
    bool striped;  // AnsBackWriterT: wave-striped slots (four workgroups per CU)
    int check;   // CHECK_SYM: 0 = 256 symbols, 1 = K <= 128, 2 = 129..255
    int msh;     // MSH_T: 10 = the literal (MSH, r) = (10, 16) form, -1 = pre-shift folded into the reciprocals, 0 = run time
    int r;       // R_T (16 with msh = 10, else 0)
    int nb;      // NB_T: 1 = NUM_BITS_OUT 1, 0 = NUM_BITS_OUT in {4, 8, 16}
};
static RfEncChoice rf_encode_choice(const scl_rans_model *m, u64 n_chunks, bool striped) {
    RfEncChoice c;
    c.slots = !striped && rf_use_slot_writer(m, n_chunks);
    c.striped = striped;
    c.check = m->fdev.K == 256 ? 0 : (m->fdev.K <= 128 ? 1 : 2);
    if (m->fdev.b != 1) {  // NUM_BITS_OUT in {4, 8, 16}: run-time constants
        c.msh = 0, c.r = 0, c.nb = 0;
        return c;
    }
    // literal form: only without a pre-shift, and for the one (MSH, r) pair that is instantiated -- the reference
    // defaults with a 4096-total table (m = 12, nsb = 29)
    c.msh = (m->fdev.enc_msh == (10u | (16u << 16))) ? 10 : (m->fdev.enc_folded ? -1 : 0);
    c.r = c.msh > 0 ? 16 : 0;
    c.nb = 1;
    return c;
}

void rans_fast_encode_launch(const scl_rans_model *m, const u8 *d_sym, u64 sym_stride, const u32 *d_lens,
                             u32 chunk_len, u64 n_chunks, u8 *d_out, u64 out_stride, u64 *d_bit_off, u32 *d_nbits,
                             u32 *d_status, hipStream_t st, bool striped) {
    const u32 blocks = (u32)((n_chunks + RF_THREADS - 1) / RF_THREADS);
    const RfEncChoice ch = rf_encode_choice(m, n_chunks, striped);
#define RF_LAUNCH_ENC_K(OUT, CHECK, MSH, R, NB)                                                                       \
    hipLaunchKernelGGL((rans_encode_fast_kernel<OUT, CHECK, MSH, R, NB>), dim3(blocks), dim3(RF_THREADS), 0, st, m->fdev, \
                       d_sym, sym_stride, d_lens, chunk_len, n_chunks, d_out, out_stride, d_bit_off, d_nbits, d_status)
#define RF_LAUNCH_ENC_W(CHECK, MSH, R, NB)                  \
    do {                                                    \
        if (ch.striped)                                     \
            RF_LAUNCH_ENC_K(EncOutT, CHECK, MSH, R, NB);    \
        else if (ch.slots)                                  \
            RF_LAUNCH_ENC_K(EncOutS, CHECK, MSH, R, NB);    \
        else                                                \
            RF_LAUNCH_ENC_K(EncOutL, CHECK, MSH, R, NB);    \
    } while (0)
#define RF_LAUNCH_ENC_C(MSH, R, NB)                         \
    do {                                                    \
        if (ch.check == 0)                                  \
            RF_LAUNCH_ENC_W(0, MSH, R, NB);                 \
        else if (ch.check == 1)                             \
            RF_LAUNCH_ENC_W(1, MSH, R, NB);                 \
        else                                                \
            RF_LAUNCH_ENC_W(2, MSH, R, NB);                 \
    } while (0)
    if (ch.nb == 0)
        RF_LAUNCH_ENC_C(0, 0, 0);
    else if (ch.msh == 10)
        RF_LAUNCH_ENC_C(10, 16, 1);
    else if (ch.msh < 0)
        RF_LAUNCH_ENC_C(-1, 0, 1);
    else
        RF_LAUNCH_ENC_C(0, 0, 1);
#undef RF_LAUNCH_ENC_C
#undef RF_LAUNCH_ENC_W
#undef RF_LAUNCH_ENC_K
}

struct RfDecChoice {
    int ml, cb, threads, nb;  // ML_T, CB_T, THREADS, NB_T of rans_decode_fast_kernel
};
static RfDecChoice rf_decode_choice(const scl_rans_model *m, u64 n_chunks) {
    RfDecChoice c;
    // up to 2 x 256 small workgroups are resident at once (64 KiB of LDS each): beyond that the 1024-lane form wins
    c.threads = n_chunks > 2ull * 256 * RD_THREADS_SMALL ? RD_THREADS : RD_THREADS_SMALL;
    c.nb = m->fdev.b != 1 ? 0 : 1;
    if (m->fdev.b != 1)  // NUM_BITS_OUT in {4, 8, 16}
        c.ml = 0, c.cb = 0;
    else if (m->fdev.m_log2 == 0xFFFFFFFFu)  // total is not a power of two
        c.ml = -1, c.cb = 0;
    else if (m->fdev.m_log2 == 12 && m->fdev.nsb == 29)  // the reference defaults with a 4096-total table: literal constants
        c.ml = 12, c.cb = 3;
    else if (m->fdev.nsb <= 29)  // the state fits shifted left by three (rf_decode_symbol)
        c.ml = 0, c.cb = 3;
    else
        c.ml = 0, c.cb = 0;
    return c;
}

// the two kernels of a batch of n_chunks aligned, equally long rows, as rocprofv3 prints them
void rans_fast_kernel_names(const scl_rans_model *m, u64 n_chunks, char *enc, char *dec, size_t cap, bool striped) {
    const RfEncChoice e = rf_encode_choice(m, n_chunks, striped);
    const RfDecChoice d = rf_decode_choice(m, n_chunks);
    if (enc)
        snprintf(enc, cap, "rans_encode_fast_kernel<AnsBackWriter%c<256>, %d, %d, %d, %d>", e.striped ? 'T' : (e.slots ? 'S' : 'L'), e.check, e.msh,
                 e.r, e.nb);
    if (dec)
        snprintf(dec, cap, "rans_decode_fast_kernel<%d, %d, %d, %d, %s>", d.ml, d.cb, d.threads, d.nb,
                 striped ? "true" : "false");
}

void rans_fast_decode_launch(const scl_rans_model *m, const u8 *d_in, u64 in_size_bytes, const u64 *d_bit_off,
                             const u32 *d_in_nbits, u64 n_chunks, u8 *d_out_sym, u64 out_stride, u32 out_cap,
                             u32 *d_out_lens, u32 *d_consumed, u32 *d_status, hipStream_t st, bool striped) {
    const RfDecChoice ch = rf_decode_choice(m, n_chunks);
#define RF_LAUNCH_DEC_K(ML, CB, TH, NB, ST)                                                                           \
    hipLaunchKernelGGL((rans_decode_fast_kernel<ML, CB, TH, NB, ST>), dim3((u32)((n_chunks + TH - 1) / TH)), dim3(TH), 0, \
                       st, m->fdev, d_in, in_size_bytes, d_bit_off, d_in_nbits, n_chunks, d_out_sym, out_stride, out_cap, \
                       d_out_lens, d_consumed, d_status)
#define RF_LAUNCH_DEC_T(ML, CB, TH, NB)                     \
    do {                                                    \
        if (striped)                                        \
            RF_LAUNCH_DEC_K(ML, CB, TH, NB, true);          \
        else                                                \
            RF_LAUNCH_DEC_K(ML, CB, TH, NB, false);         \
    } while (0)
#define RF_LAUNCH_DEC(ML, CB, NB)                           \
    do {                                                    \
        if (ch.threads == RD_THREADS)                       \
            RF_LAUNCH_DEC_T(ML, CB, RD_THREADS, NB);        \
        else                                                \
            RF_LAUNCH_DEC_T(ML, CB, RD_THREADS_SMALL, NB);  \
    } while (0)
    if (ch.nb == 0)
        RF_LAUNCH_DEC(0, 0, 0);
    else if (ch.ml < 0)
        RF_LAUNCH_DEC(-1, 0, 1);
    else if (ch.ml == 12)
        RF_LAUNCH_DEC(12, 3, 1);
    else if (ch.cb == 3)
        RF_LAUNCH_DEC(0, 3, 1);
    else
        RF_LAUNCH_DEC(0, 0, 1);
#undef RF_LAUNCH_DEC
#undef RF_LAUNCH_DEC_T
#undef RF_LAUNCH_DEC_K
}
