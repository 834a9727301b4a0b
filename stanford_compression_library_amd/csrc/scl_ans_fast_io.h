// scl_ans_fast_io.h -- line-granular stream I/O shared by the rANS and tANS fast kernels (same stream layout:
// produced back to front, read front to back).  Internal to csrc/.
//
// Memory granularity rule (profiles/r01_v2_pmc_summary.txt -> r01_v3): with one lane per 4 KiB chunk a CU owns
// 1024 open cache lines for input and 1024 for output -- more than L1 and, per XCD, the whole L2 -- so a lane
// must only ever move whole lines / 64-byte sectors:
//   symbols in : 128 bytes per lane as 8 back-to-back 16-byte loads into registers (Line128);
//   stream out : completed big-endian words go to a per-lane ring in LDS (word w of thread t at [w][t]: bank =
//                t mod 32 for every w, so the scattered ds_write_b32 never conflict) and leave as 64 contiguous
//                bytes (4 back-to-back 16-byte stores) once 16 words are pending (AnsBackWriter);
//   stream in  : whole 128-byte lines prefetched into registers, byte-swapped and moved into the same kind of ring
//                one 64-byte half at a time; the 64-bit bit window refills one word at a time from it (AnsBitReader).
// The ring of a workgroup must start at LDS offset 0 (addresses wrap with a single AND).
#pragma once
#include "scl_common.h"

// Instruction selection follows profiles/r01_ubench_valu_issue_cost.txt: on gfx950 only
// v_add/v_sub/v_lshrrev/v_ashrrev/v_and/v_or/v_xor/v_mov (VGPR or literal operands) issue at 32 lanes/clk;
// compares, carries, left shifts, SDWA, every VOP3 form, the multiplies and any SGPR operand cost twice that.
// Hence: sign-bit arithmetic instead of compare+carry, model constants as template literals, fields laid out
// so they need no extraction (k1 rides in the byte v_mad_u32_u24 ignores).
// Measured on the 1 GiB batch: the branch-free form of put() below is slower at 4 waves per SIMD (encode 0.706 vs
// 0.675 ms) and only wins when a lone wave per SIMD is latency bound (65 536 chunks: 0.240 vs 0.257 ms): off.
template <int THREADS, bool HOLD_HALF_LINE = true>
struct AnsBackWriter {
    static constexpr u32 RING_BYTES = 32u * THREADS * 4u;  // placed at LDS offset 0 of the workgroup
    u32 lo;    // pending bits (right-aligned; newest bits are the high ones), < 32 of them
    u32 nacc;  // number of pending bits
    // The stream grows towards lower addresses, so the ring is filled from its top row down: ascending rows are then
    // ascending memory addresses and a flush reads its 16-byte pieces in register order (filled upwards, every
    // ds_read2 pair arrived swapped and cost two copies).
    u32 ra;    // LDS byte address of the ring word that completes next (thread column, moves DOWN a row per word, wraps)
    u32 fa;    // LDS byte address of the lowest row of the oldest unflushed group of 16 words
    u32 nfl;   // words already stored to memory
    u8 *slot_end;

    // completed words not yet stored to memory: the oldest of them sits in the top row of the group at fa, the next
    // word goes to ra, one row lower per word (never 32 pending: see maybe_flush) -- not kept as a counter, which
    // would cost an instruction per completed word where this costs three per flush check
    __device__ __forceinline__ u32 pend() const { return ((fa + 15 * THREADS * 4 - ra) & (RING_BYTES - 1)) / (THREADS * 4); }

    __device__ __forceinline__ void init(u32 tid, u8 *slot_end_) {
        lo = 0;
        nacc = 0;
        ra = tid * 4 + 31 * THREADS * 4;
        fa = tid * 4 + 16 * THREADS * 4;
        nfl = 0;
        slot_end = slot_end_;
        have_held = 0;
        held[0] = held[1] = held[2] = held[3] = make_uint4(0, 0, 0, 0);
    }
    static __device__ __forceinline__ u32 *ring_at(char *lds, u32 byte_addr) {
        return reinterpret_cast<u32 *>(lds + byte_addr);
    }
    // append `w` bits (v < 2^w, w < 32) in front of the stream
    __device__ __forceinline__ void put(char *lds, u32 v, u32 w) {
        const u32 lo2 = (v << nacc) | lo;
        const u32 nacc2 = nacc + w;
        if (nacc2 >= 32) {  // a word completes only if bits were pending, so 32 - nacc is a valid shift
            *ring_at(lds, ra) = __builtin_bswap32(lo2);
            ra = (ra - THREADS * 4) & (RING_BYTES - 1);
            lo = v >> (32 - nacc);
            nacc = nacc2 - 32;
        } else {
            lo = lo2;
            nacc = nacc2;
        }
    }
    __device__ __forceinline__ void put32(char *lds, u32 v, u32 w) {  // any w <= 32 (header fields)
        if (w > 16) {
            put(lds, v & 0xFFFFu, 16);
            put(lds, v >> 16, w - 16);
        } else {
            put(lds, v, w);
        }
    }
    // 16 pending words leave the ring at a time; the first 64-byte half of a cache line waits in registers
    // (`held`) until the second half is ready, then the whole 128-byte line is stored as one burst of eight
    // 16-byte stores (half-line bursts cost ~20 % more time: profiles, decode 0.98 -> 0.81 ms).
    // Call at least every 32 symbols (<= 12 new words on top of <= 15 pending, ring of 32).
    uint4 held[4];
    u32 have_held;

    __device__ __forceinline__ void maybe_flush(char *lds) {
        if (pend() >= 16) {
            const char *r = lds + fa;
            // word j of this group (completion order) sits at -4*(nfl + j + 1) in memory and in row 15 - j of the
            // group: the 16-byte piece i (ascending addresses) is the rows 4i .. 4i+3
#define SCL_RING_W(j) (*reinterpret_cast<const u32 *>(r + (j) * THREADS * 4))
#define SCL_RING_Q(i) make_uint4(SCL_RING_W(4 * (i)), SCL_RING_W(4 * (i) + 1), SCL_RING_W(4 * (i) + 2), SCL_RING_W(4 * (i) + 3))
            if (HOLD_HALF_LINE && !have_held) {
                // upper-address half of a line: read straight into the registers that keep it until the lower half
                // is ready (read before the branch, the words took 16 extra copies to get there)
                held[0] = SCL_RING_Q(0);
                held[1] = SCL_RING_Q(1);
                held[2] = SCL_RING_Q(2);
                held[3] = SCL_RING_Q(3);
                have_held = 1;
            } else {
                const uint4 q0 = SCL_RING_Q(0), q1 = SCL_RING_Q(1), q2 = SCL_RING_Q(2), q3 = SCL_RING_Q(3);
                uint4 *p = reinterpret_cast<uint4 *>(slot_end - 4 * (u64)(nfl + 16));
                p[0] = q0;
                p[1] = q1;
                p[2] = q2;
                p[3] = q3;
                if (HOLD_HALF_LINE) {  // ... and the whole line leaves as one burst
                    p[4] = held[0];
                    p[5] = held[1];
                    p[6] = held[2];
                    p[7] = held[3];
                    have_held = 0;
                }
            }
#undef SCL_RING_Q
#undef SCL_RING_W
            nfl += 16;
            fa ^= 16 * THREADS * 4;  // the ring has two halves of 16 words
        }
    }
    __device__ __forceinline__ u64 finish(char *lds) {
        maybe_flush(lds);
        if (have_held) {  // nfl counts the held words as flushed: they belong at -4*nfl .. -4*(nfl-16)
            uint4 *p = reinterpret_cast<uint4 *>(slot_end - 4 * (u64)nfl);
            p[0] = held[0];
            p[1] = held[1];
            p[2] = held[2];
            p[3] = held[3];
        }
        u32 *end32 = reinterpret_cast<u32 *>(slot_end);
        u32 a = fa + 15 * THREADS * 4;  // the oldest word of the group sits in its top row (fewer than 16 pending here)
        const u32 np = pend();
        for (u32 j = 0; j < np; ++j) {
            end32[-(i64)(nfl + j) - 1] = *ring_at(lds, a);
            a -= THREADS * 4;
        }
        const u32 words = nfl + np;
        if (nacc) end32[-(i64)words - 1] = __builtin_bswap32(lo);  // zero bits in front of the stream
        return (u64)words * 32 + nacc;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Round-2 back writer: 64-bit bit window -> per-lane LDS ring of 64 words -> whole 128-byte lines stored by quads.
//
// (1) Bits.  The pending bits are the TOP n bits of a 64-bit window hi:lo, newest on top:
//       push(x, k):  lo = alignbit(hi, lo, k);  hi = alignbit(x, hi, k);  n += k
//     v_alignbit takes the low k bits of its first operand by itself: no field extraction (v_bfe), no merging of two
//     fields, no shift of the field to its place in an accumulator.  check() every two symbols (n <= 32 + 2*13 < 64):
//     n >= 33 means the oldest 32 bits, alignbit(hi, lo, 64 - n), are a complete word; the rest of the window stays
//     where it is (top-aligned), n -= 32.  The counter is room = 32 - n, unsigned: the subtraction that accounts for a
//     pair of symbols borrows exactly when n reaches 33 (no compare instruction), and the wrapped value's low 5 bits are
//     the word's shift amount (v_alignbit reads 5 bits: (32 - n) mod 32 = 64 - n for 33 <= n <= 63).
// (2) Ring.  256 bytes per lane, [thread][word]: the two halves are two 128-byte line buffers holding words in MEMORY
//     order (the stream grows downwards, so the write offset walks 124, 120, .. 0, 252, .. 128, 124, ..).  The ring is
//     256-byte aligned, so the write ADDRESS is the only state: wa = ((wa - 4) & 255) | base.  A lane's scattered 4-byte
//     writes now collide in the banks (bank = word index mod 32, whatever the lane), which ds_write_b32 mostly hides
//     (measured: no slower than the conflict-free [word][thread] layout of round 1); what the layout buys is that a
//     16-byte piece of a line is ONE aligned ds_read_b128 -- for any lane.
// (3) Stores.  What bounds a lane-per-chunk coder on this chip is the SHAPE of its write requests: with the same
//     lines, the same bytes and no arithmetic at all, lanes storing their own lines (8 x 16 bytes) take 0.58 ms per GiB
//     batch, four lanes per 64-byte half line 0.52, eight lanes per whole line 0.42 (tools/ubench/linecopy3.hip; the read
//     shape makes no difference).  The round-1 encoder (0.59 ms) sat exactly on the first figure.  So a line is stored by
//     the four lanes of the source lane's quad, 2 x 16 bytes each, as two back-to-back 64-byte requests -- without
//     synchronising the lanes, which complete their lines at data-dependent times: at a flush point (wave-uniform code,
//     every 64 symbols) four rounds r = 0..3; in round r the quads whose lane r has a complete line store it; lane j
//     of the quad reads the pieces j and j + 4 of the SOURCE lane's line buffer (two ds_read_b128; the source's
//     addresses come by DPP quad broadcast) and stores them at the source's position + 16 j (+ 64).
//     That needs a whole line buffered while the next one fills (64-word ring = 2 workgroups per CU instead of 4: measured
//     neutral, the kernel issues as fast with 2 waves per SIMD) and every lane of the wave alive until the last flush
//     point: whole waves of equally long chunks; otherwise flush_lane(), the lane's own eight 16-byte stores.
template <int R>
__device__ __forceinline__ u32 scl_quad_bcast(u32 v) {  // value of lane R of this lane's quad (quad_perm:[R,R,R,R])
    return (u32)__builtin_amdgcn_mov_dpp((int)v, R * 0x55, 0xF, 0xF, true);
}
template <int THREADS>
struct AnsBackWriterL {
    static constexpr u32 FLUSH_MASK = 3;   // the encoder's flush points: every (FLUSH_MASK + 1) x 16 symbols
    static constexpr u32 FLUSH_PHASE = 1;  // ... after block 1 (mod 4) of a line
    static constexpr u32 WG_PER_CU = 2;    // 64 KiB of rings + the 4 KiB table per 256 lanes
    static constexpr bool STRIPED = false;
    static constexpr u32 LANE_BYTES = 256;                   // one 256-byte ring per lane, 256-byte aligned
    static constexpr u32 RING_BYTES = THREADS * LANE_BYTES;  // placed at LDS offset 0 of the workgroup
    u32 hi, lo;   // the window
    u32 room;     // 32 - (number of pending bits), as an unsigned counter: a push that BORROWS completed a word
    u32 wa;       // LDS byte address (from the ring base) of the word that completes next: base | offset 0..252
    u32 th4;      // ring offset of the FIRST word (highest address) of the oldest unflushed line
    u32 base;     // tid * LANE_BYTES (low 8 bits zero)
    u32 goff;     // byte offset (from the workgroup's output base) of the END of the next line to store
    u32 goff0;    // ... of the slot end

    // Lanes that run in LOCKSTEP (equal symbol costs, e.g. a uniform table: every lane completes its words at the same
    // steps) would all write the same ring offset -- the same bank, 32 ways.  Each lane's ring is therefore rotated by
    // 16 * (lane mod 16) bytes: whole pieces, so a piece stays one aligned 16-byte read; lockstep lanes then spread over
    // 8 bank groups (4-way, which ds_write_b32 half hides) and a line may wrap around the end of its ring.
    static __device__ __forceinline__ u32 rot_of(u32 tid) { return 16u * (tid & 15u); }

    __device__ __forceinline__ u32 pend4() const { return (th4 - wa) & 255u; }  // 4 * completed words not yet stored (< 256)

    __device__ __forceinline__ void init(u32 tid, u32 slot_end_off) {
        hi = lo = 0;
        room = 32;
        base = tid * LANE_BYTES;
        th4 = (124u + rot_of(tid)) & 255u;
        wa = base | th4;
        goff = goff0 = slot_end_off;
    }
    // the low k bits of v go in front of the stream; k < 32, bits of v above bit k are ignored
    __device__ __forceinline__ void push(u32 v, u32 k) {
        lo = __builtin_amdgcn_alignbit(hi, lo, k);
        hi = __builtin_amdgcn_alignbit(v, hi, k);
    }
    // push() in two halves (round 5): what two pushes of k0 and k1 bits move from hi into lo is what ONE funnel shift of the
    // hi of before by k0 + k1 moves (k0 + k1 < 32) -- so a pair of symbols updates hi twice and lo once: three v_alignbit
    // instead of four
    __device__ __forceinline__ void push_hi(u32 v, u32 k) { hi = __builtin_amdgcn_alignbit(v, hi, k); }
    __device__ __forceinline__ void fold_lo(u32 hi_before, u32 bits) { lo = __builtin_amdgcn_alignbit(hi_before, lo, bits); }
    // account for the `bits` (<= 26) pushed since the last call.  The subtraction's borrow IS the test "33 or more bits
    // pending" (v_sub_co_u32 + a branch on VCC: no separate compare); the wrapped counter's low 5 bits are the shift that
    // brings the oldest 32 bits down: (32 - n) mod 32 = 64 - n for 33 <= n <= 63.
    // RING_OFF: byte offset of the rings in the workgroup's LDS block (an immediate of the ds_write).  The block is
    // written by hand: the compiler's version of the same seven instructions carries an s_cbranch_execz that skips them
    // when no lane completed a word -- practically never, with 64 lanes -- and every instruction on this path, which the
    // whole wave runs for every pair of symbols, is ~2.5 % of the kernel (0.575 -> 0.556 ms).  LDS instructions of a wave
    // execute in order, so the flush rounds' reads see these writes; the compiler, which does not know of them, can only
    // wait too long for its own LDS operations, never too short.
    template <u32 RING_OFF>
    __device__ __forceinline__ void check(char *lds, u32 bits) {
        u32 w, t;
        u64 sv;
        asm volatile(
            "v_sub_co_u32 %[room], vcc, %[room], %[bits]\n\t"
            "s_and_saveexec_b64 %[sv], vcc\n\t"
            "v_alignbit_b32 %[w], %[hi], %[lo], %[room]\n\t"
            "v_add_u32 %[t], 0xfc, %[wa]\n\t"
            "v_perm_b32 %[w], 0, %[w], %[sel]\n\t"
            "v_add_u32 %[room], 32, %[room]\n\t"
            "ds_write_b32 %[wa], %[w] offset:%[off]\n\t"
            "v_and_or_b32 %[wa], %[t], %[m255], %[base]\n\t"
            "s_or_b64 exec, exec, %[sv]"
            : [room] "+v"(room), [wa] "+v"(wa), [w] "=&v"(w), [t] "=&v"(t), [sv] "=&s"(sv)
            : [bits] "v"(bits), [hi] "v"(hi), [lo] "v"(lo), [sel] "s"(0x00010203u), [m255] "s"(255u), [base] "v"(base),
              [off] "i"(RING_OFF)
            : "vcc", "memory");
        (void)lds;
    }
    template <u32 RING_OFF>
    __device__ __forceinline__ void put32(char *lds, u32 v, u32 w) {  // any w <= 32 (header fields)
        if (w > 16) {
            push(v, 16);
            check<RING_OFF>(lds, 16);
            push(v >> 16, w - 16);
            check<RING_OFF>(lds, w - 16);
        } else {
            push(v, w);
            check<RING_OFF>(lds, w);
        }
    }
    template <int R>
    __device__ __forceinline__ void quad_round(const char *lds, u8 *wg_out, u32 f, u32 qbase, u32 j16) const {
        if (scl_quad_bcast<R>(f)) {  // all four lanes of the quads whose lane R has a complete line
            // lowest address of the source's line inside its ring: 124 bytes below its first word, modulo the ring
            const u32 l0 = scl_quad_bcast<R>(th4) + (132u + j16);  // (th4 - 124 + 16 j) mod 256, the +256 keeps it positive
            const u32 go_s = scl_quad_bcast<R>(goff);
            const char *r = lds + qbase + R * LANE_BYTES;
            const uint4 q0 = *reinterpret_cast<const uint4 *>(r + (l0 & 255u));
            const uint4 q1 = *reinterpret_cast<const uint4 *>(r + ((l0 + 64u) & 255u));
            u8 *p = wg_out + (go_s - 128u + j16);
            *reinterpret_cast<uint4 *>(p) = q0;
            *reinterpret_cast<uint4 *>(p + 64) = q1;
        }
    }
    // WAVE-UNIFORM call (all 64 lanes), at least every 64 symbols: <= 26 new words on top of <= 31 pending
    __device__ __forceinline__ void flush_quad(char *lds, u8 *wg_out, u32 tid) {
        const u32 f = pend4() >= 128u ? 1u : 0u;
        if (__builtin_amdgcn_ballot_w64(f != 0)) {
            const u32 qbase = (tid & ~3u) * LANE_BYTES, j16 = 16u * (tid & 3u);
            quad_round<0>(lds, wg_out, f, qbase, j16);
            quad_round<1>(lds, wg_out, f, qbase, j16);
            quad_round<2>(lds, wg_out, f, qbase, j16);
            quad_round<3>(lds, wg_out, f, qbase, j16);
            if (f) {
                goff -= 128u;
                th4 ^= 128u;
            }
        }
    }
    // the lane's own stores (ragged batches, partial waves): one line = eight 16-byte stores
    __device__ __forceinline__ void flush_lane(char *lds, u8 *wg_out) {
        if (pend4() >= 128u) {
            const char *r = lds + base;
            const u32 l0 = th4 + 132u;
            uint4 *p = reinterpret_cast<uint4 *>(wg_out + (goff - 128u));
            uint4 q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = *reinterpret_cast<const uint4 *>(r + ((l0 + 16u * i) & 255u));
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = q[i];
            goff -= 128u;
            th4 ^= 128u;
        }
    }
    __device__ __forceinline__ u64 finish(char *lds, u8 *wg_out) {  // per lane; returns the stream length in bits
        flush_lane(lds, wg_out);
        flush_lane(lds, wg_out);
        u32 *end32 = reinterpret_cast<u32 *>(wg_out + goff);  // words go below this, newest at the lowest address
        const u32 np = pend4() >> 2;                         // < 32 now
        u32 a = th4;
        for (u32 j = 0; j < np; ++j) {
            end32[-(i64)j - 1] = *reinterpret_cast<const u32 *>(lds + base + a);
            a = (a - 4u) & 252u;
        }
        const u32 n = 32u - room;  // <= 32 after the last check(): the top n bits of hi, zero bits in front of them
        if (n) end32[-(i64)np - 1] = __builtin_bswap32(n == 32 ? hi : (hi >> (32 - n)));
        return (u64)(((goff0 - goff) >> 2) + np) * 32 + n;
    }
};

// Round-2 back writer, three-workgroups-per-CU form: the same bit window and the same quad-cooperative line stores
// over rings of 192 bytes = three 64-byte SLOTS per lane, so that three workgroups of 256 lanes fit a CU
// (3 x (4 KiB table + 48 KiB)) -- a SIMD with two waves cannot cover their stalls (a wave issues at most one VALU
// instruction per ~5 clocks: tools/ubench/valu_rate.hip), a third is worth 9 % of the rANS encoder's time.
//   * A line is two slots, upper half then lower half (the stream grows downwards); slots never wrap, so a helper
//     lane's two piece addresses are two DPP adds (source's slot offset + its own quad base + 16 j) -- cheaper than
//     the 256-byte ring's modulo arithmetic.  No per-lane rotation: see the LOCKSTEP note below.
//   * 48 words hold a line waiting for its flush point (<= 31 words) plus what 32 symbols add (<= 13): flush points
//     are 32 symbols apart (FLUSH_MASK = 1).
//   * The word pointer wraps at a non-power-of-two: v_cmp + v_cndmask + v_add instead of v_add + v_and_or.
// LOCKSTEP: lanes whose symbols all cost the same number of bits (a table of equal frequencies) complete their words
// at the same steps and write the same ring offset; with a lane stride of 192 bytes and no rotation that is 2 of the
// 32 banks.  The launch code keeps such tables on AnsBackWriterL (whose rings are rotated per lane).
template <int THREADS>
struct AnsBackWriterS {
    static constexpr u32 FLUSH_MASK = 1;
    static constexpr u32 FLUSH_PHASE = 0;  // after blocks 0, 2, 4, 6 of a line: none just before its end (see the kernel)
    static constexpr u32 WG_PER_CU = 3;
    static constexpr bool STRIPED = false;
    static constexpr u32 LANE_BYTES = 192;
    static constexpr u32 RING_BYTES = THREADS * LANE_BYTES;
    u32 hi, lo;   // the window
    u32 room;     // 32 - (number of pending bits): a push that BORROWS completed a word
    u32 wo;       // ring offset (0..188) of the word that completes next; its LDS address is base + wo
    u32 rhi, rlo; // ring offsets (0, 64 or 128) of the slots holding the upper / lower half of the oldest unflushed line
    u32 base;     // tid * LANE_BYTES
    u32 goff;     // byte offset (from the workgroup's output base) of the END of the next line to store
    u32 goff0;    // ... of the slot end

    // 4 * completed words not yet stored (<= 176): the line's first word is the top word of its upper slot
    __device__ __forceinline__ u32 pend4() const {
        const u32 t = rhi + 60u - wo;  // negative when the write offset is above the line's first word
        return t + ((u32)((int)t >> 31) & LANE_BYTES);
    }
    __device__ __forceinline__ void init(u32 tid, u32 slot_end_off) {
        hi = lo = 0;
        room = 32;
        base = tid * LANE_BYTES;
        rhi = 128;
        rlo = 64;
        wo = 188u;
        goff = goff0 = slot_end_off;
    }
    __device__ __forceinline__ void push(u32 v, u32 k) {  // the low k bits of v go in front of the stream; k < 32
        lo = __builtin_amdgcn_alignbit(hi, lo, k);
        hi = __builtin_amdgcn_alignbit(v, hi, k);
    }
    __device__ __forceinline__ void push_hi(u32 v, u32 k) { hi = __builtin_amdgcn_alignbit(v, hi, k); }  // see AnsBackWriterL
    __device__ __forceinline__ void fold_lo(u32 hi_before, u32 bits) { lo = __builtin_amdgcn_alignbit(hi_before, lo, bits); }
    template <u32 RING_OFF>
    __device__ __forceinline__ void check(char *lds, u32 bits) {  // as AnsBackWriterL::check, by hand for the same reason
        u32 w, t;
        u64 sv;
        asm volatile(
            "v_sub_co_u32 %[room], vcc, %[room], %[bits]\n\t"
            "s_and_saveexec_b64 %[sv], vcc\n\t"
            "v_alignbit_b32 %[w], %[hi], %[lo], %[room]\n\t"
            "v_add_u32 %[t], %[base], %[wo]\n\t"
            "v_perm_b32 %[w], 0, %[w], %[sel]\n\t"
            "v_add_u32 %[wo], -4, %[wo]\n\t"
            "ds_write_b32 %[t], %[w] offset:%[off]\n\t"
            "v_min_u32 %[wo], 0xbc, %[wo]\n\t"  // (wo - 4) mod 192: only wo == 0 wraps, to a huge unsigned value
            "v_add_u32 %[room], 32, %[room]\n\t"
            "s_or_b64 exec, exec, %[sv]"
            : [room] "+v"(room), [wo] "+v"(wo), [w] "=&v"(w), [t] "=&v"(t), [sv] "=&s"(sv)
            : [bits] "v"(bits), [hi] "v"(hi), [lo] "v"(lo), [sel] "s"(0x00010203u), [base] "v"(base), [off] "i"(RING_OFF)
            : "vcc", "memory");
        (void)lds;
    }
    template <u32 RING_OFF>
    __device__ __forceinline__ void put32(char *lds, u32 v, u32 w) {  // any w <= 32 (header fields)
        if (w > 16) {
            push(v, 16);
            check<RING_OFF>(lds, 16);
            push(v >> 16, w - 16);
            check<RING_OFF>(lds, w - 16);
        } else {
            push(v, w);
            check<RING_OFF>(lds, w);
        }
    }
    template <int R>
    __device__ __forceinline__ void quad_round(const char *lds, u8 *wg_out, u32 f, u32 qj, u32 j16) const {
        if (scl_quad_bcast<R>(f)) {  // all four lanes of the quads whose lane R has a complete line
            const char *r = lds + R * LANE_BYTES;
            const uint4 q0 = *reinterpret_cast<const uint4 *>(r + (scl_quad_bcast<R>(rlo) + qj));
            const uint4 q1 = *reinterpret_cast<const uint4 *>(r + (scl_quad_bcast<R>(rhi) + qj));
            u8 *p = wg_out + (scl_quad_bcast<R>(goff) - 128u + j16);
            *reinterpret_cast<uint4 *>(p) = q0;
            *reinterpret_cast<uint4 *>(p + 64) = q1;
        }
    }
    __device__ __forceinline__ void line_done() {
        goff -= 128u;
        const u32 l = rlo;
        rhi = min(l + 128u, l - 64u);   // (l - 64) mod 192
        rlo = min(l + 64u, l - 128u);   // (l - 128) mod 192
    }
    // WAVE-UNIFORM call (all 64 lanes), at least every 32 symbols: <= 13 new words on top of <= 31 pending
    __device__ __forceinline__ void flush_quad(char *lds, u8 *wg_out, u32 tid) {
        const u32 f = pend4() >= 128u ? 1u : 0u;
        if (__builtin_amdgcn_ballot_w64(f != 0)) {
            const u32 j16 = 16u * (tid & 3u), qj = (tid & ~3u) * LANE_BYTES + j16;
            quad_round<0>(lds, wg_out, f, qj, j16);
            quad_round<1>(lds, wg_out, f, qj, j16);
            quad_round<2>(lds, wg_out, f, qj, j16);
            quad_round<3>(lds, wg_out, f, qj, j16);
            if (f) line_done();
        }
    }
    // the lane's own stores (ragged batches, partial waves): one line = eight 16-byte stores
    __device__ __forceinline__ void flush_lane(char *lds, u8 *wg_out) {
        if (pend4() >= 128u) {
            const uint4 *a = reinterpret_cast<const uint4 *>(lds + base + rlo);
            const uint4 *b = reinterpret_cast<const uint4 *>(lds + base + rhi);
            uint4 *p = reinterpret_cast<uint4 *>(wg_out + (goff - 128u));
            uint4 q[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = a[i], q[4 + i] = b[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = q[i];
            line_done();
        }
    }
    __device__ __forceinline__ u64 finish(char *lds, u8 *wg_out) {  // per lane; returns the stream length in bits
        flush_lane(lds, wg_out);
        flush_lane(lds, wg_out);
        u32 *end32 = reinterpret_cast<u32 *>(wg_out + goff);  // words go below this, newest at the lowest address
        const u32 np = pend4() >> 2;                         // < 32 now
        u32 a = rhi + 60u;
        for (u32 j = 0; j < np; ++j) {
            end32[-(i64)j - 1] = *reinterpret_cast<const u32 *>(lds + base + a);
            a = min(a - 4u, LANE_BYTES - 4u);
        }
        const u32 n = 32u - room;  // <= 32 after the last check(): the top n bits of hi, zero bits in front of them
        if (n) end32[-(i64)np - 1] = __builtin_bswap32(n == 32 ? hi : (hi >> (32 - n)));
        return (u64)(((goff0 - goff) >> 2) + np) * 32 + n;
    }
};

// Round-6 back writer for WAVE-STRIPED slots: four workgroups per CU.
//
// What held the L / S writers at two or three waves per SIMD is the line buffer: a lane that owns a contiguous slot must
// store whole 128-byte lines (anything smaller reaches memory as its own partial-line request: rounds 1-3), so it buffers
// a whole line while the next one fills -- 192 to 256 bytes of LDS per lane.  In a STRIPED slot the 64 streams of a wave
// are interleaved at 16-byte granularity:
//     byte b of the logical slot of lane l (stride S, stream ending at S)  ->  wave_slot + (b / 16) * 1024 + 16 l + b % 16
// i.e. the 16-byte piece q of every lane forms ONE contiguous kilobyte ("row" q): eight neighbouring lanes share a
// 128-byte line.  A lane then owes memory 16-byte pieces, not 128-byte lines: the ring shrinks to 128 bytes per lane, 32 KiB
// + the 4 KiB table per workgroup = four workgroups per CU, four waves per SIMD -- and the WAVE stores a row (64 adjacent
// pieces, eight whole lines, one instruction) when every lane holds its piece of it (flush()).  No lane depends on
// another for correctness, so ragged batches and partial waves take the same path.  The logical position of every bit is what it was -- `bit_offset` /
// `nbits` keep their meaning -- only the mapping to memory changes (scl_stripe_byte below; the decoder's reader and
// scl_streams_compact undo it).
//   * bit window, check(): as AnsBackWriterL (the asm block differs in the ring constants only);
//   * ring: 32 words per lane, [thread][word], words in MEMORY order inside 16-byte pieces, rotated by 16 * (lane mod 8)
//     bytes against lockstep tables; flush points 32 symbols apart (<= 13 new words on top of the <= 18 a lane may keep);
//   * flush(): the wave stores the next row when every lane holds a complete piece (see the comment there).
__device__ __forceinline__ u64 scl_stripe_byte(u64 logical_byte, u64 stride) {
    // logical byte address in a batch of slots of `stride` bytes (a multiple of 16) -> physical byte address
    const u64 slot = logical_byte / stride, b = logical_byte - slot * stride;
    return (slot >> 6) * (stride << 6) + (b >> 4) * 1024u + ((slot & 63u) << 4) + (b & 15u);
}
template <int THREADS>
struct AnsBackWriterT {
    static constexpr u32 FLUSH_MASK = 1;   // the encoder's flush points: every 32 symbols
    static constexpr u32 FLUSH_PHASE = 0;  // after blocks 0, 2, 4, 6 of a line: none just before its end (see the kernel)
    static constexpr u32 WG_PER_CU = 4;    // 32 KiB of rings + the 4 KiB table per 256 lanes
    static constexpr u32 LANE_BYTES = 128;
    static constexpr u32 RING_BYTES = THREADS * LANE_BYTES;
    static constexpr bool STRIPED = true;
    u32 hi, lo;   // the window
    u32 room;     // 32 - (number of pending bits): a push that BORROWS completed a word
    u32 wa;       // LDS byte address (from the ring base) of the word that completes next: base | offset 0..124
    u32 th4;      // ring offset of the FIRST word (highest address) of the oldest unflushed piece (= 12 mod 16)
    u32 base;     // tid * LANE_BYTES
    u32 goff;     // byte offset (from the workgroup's output base) of this lane's next piece
    u32 goff0;    // ... of its first piece (the last 16 bytes of its logical slot)

    __device__ __forceinline__ u32 pend4() const { return (th4 - wa) & 127u; }  // 4 * completed words not yet stored

    // out_stride: bytes per logical slot (a multiple of 16; 256 lanes * out_stride < 2^32)
    __device__ __forceinline__ void init(u32 tid, u32 out_stride) {
        hi = lo = 0;
        room = 32;
        base = tid * LANE_BYTES;
        th4 = (124u + 16u * (tid & 7u)) & 127u;
        wa = base | th4;
        goff = goff0 = ((tid >> 6) + 1u) * (out_stride << 6) - 1024u + ((tid & 63u) << 4);
    }
    __device__ __forceinline__ void push(u32 v, u32 k) {  // the low k bits of v go in front of the stream; k < 32
        lo = __builtin_amdgcn_alignbit(hi, lo, k);
        hi = __builtin_amdgcn_alignbit(v, hi, k);
    }
    __device__ __forceinline__ void push_hi(u32 v, u32 k) { hi = __builtin_amdgcn_alignbit(v, hi, k); }  // see AnsBackWriterL
    __device__ __forceinline__ void fold_lo(u32 hi_before, u32 bits) { lo = __builtin_amdgcn_alignbit(hi_before, lo, bits); }
    template <u32 RING_OFF>
    __device__ __forceinline__ void check(char *lds, u32 bits) {  // as AnsBackWriterL::check, by hand for the same reason
        u32 w, t;
        u64 sv;
        asm volatile(
            "v_sub_co_u32 %[room], vcc, %[room], %[bits]\n\t"
            "s_and_saveexec_b64 %[sv], vcc\n\t"
            "v_alignbit_b32 %[w], %[hi], %[lo], %[room]\n\t"
            "v_add_u32 %[t], 0x7c, %[wa]\n\t"
            "v_perm_b32 %[w], 0, %[w], %[sel]\n\t"
            "v_add_u32 %[room], 32, %[room]\n\t"
            "ds_write_b32 %[wa], %[w] offset:%[off]\n\t"
            "v_and_or_b32 %[wa], %[t], %[m127], %[base]\n\t"
            "s_or_b64 exec, exec, %[sv]"
            : [room] "+v"(room), [wa] "+v"(wa), [w] "=&v"(w), [t] "=&v"(t), [sv] "=&s"(sv)
            : [bits] "v"(bits), [hi] "v"(hi), [lo] "v"(lo), [sel] "s"(0x00010203u), [m127] "s"(127u), [base] "v"(base),
              [off] "i"(RING_OFF)
            : "vcc", "memory");
        (void)lds;
    }
    template <u32 RING_OFF>
    __device__ __forceinline__ void put32(char *lds, u32 v, u32 w) {  // any w <= 32 (header fields)
        if (w > 16) {
            push(v, 16);
            check<RING_OFF>(lds, 16);
            push(v >> 16, w - 16);
            check<RING_OFF>(lds, w - 16);
        } else {
            push(v, w);
            check<RING_OFF>(lds, w);
        }
    }
    // piece r (0 = oldest) of the complete ones this lane holds -> its row; nothing is updated (see rows_done)
    template <u32 R>
    __device__ __forceinline__ void store_row(char *lds, u8 *wg_out) const {
        const uint4 q = *reinterpret_cast<const uint4 *>(lds + base + ((th4 - 12u - 16u * R) & 127u));
        typedef u32 u32x4_nt __attribute__((ext_vector_type(4)));
        const u32x4_nt t = {q.x, q.y, q.z, q.w};
        // non-temporal: a row is written once, whole, and read next by another kernel (same-box alternation: the round trip
        // is 1 % faster with the hint; AnsBackWriterL's quad stores were neutral with it)
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4_nt *>(wg_out + goff - 1024u * R));
    }
    __device__ __forceinline__ void rows_done(u32 m) {
        goff -= 1024u * m;
        th4 = (th4 - 16u * m) & 127u;
    }
    // Flush point (any subset of the wave may call; the lanes present act together).  Lanes complete their pieces at
    // data-dependent times; a piece stored the moment it is complete makes every 128-byte line arrive in two or three
    // partial writes (measured: +0.13 ms per GiB batch, worse than the L writer).  So the wave stores ROW BY ROW: the next
    // row goes out when EVERY lane present holds a complete piece -- one instruction, 64 adjacent pieces, eight whole lines
    // when the lanes are where i.i.d. data keeps them: in step to within a piece or two, which the ring absorbs (a lane may
    // leave a flush point holding up to `guard` bytes of complete words).  A lane beyond that stores its own oldest pieces
    // -- partial lines; lanes that drift apart for good (very different statistics from chunk to chunk) all end up there,
    // at the cost of the unsynchronised form (+3 %), never of correctness: the ring cannot reach 32 words.
    // guard: 4 * (31 - the most words 32 symbols can complete): 72 for <= 13 bits per symbol, 60 for <= 16
    __device__ __forceinline__ void flush(char *lds, u8 *wg_out, u32 guard) {
        const u32 p = pend4();
        const u64 all = __builtin_amdgcn_ballot_w64(true);
        if (__builtin_amdgcn_ballot_w64(p >= 16u) == all) {
            store_row<0>(lds, wg_out);
            if (__builtin_amdgcn_ballot_w64(p >= 32u) == all) {
                store_row<1>(lds, wg_out);
                if (__builtin_amdgcn_ballot_w64(p >= 48u) == all) {
                    store_row<2>(lds, wg_out);
                    if (__builtin_amdgcn_ballot_w64(p >= 64u) == all) {
                        store_row<3>(lds, wg_out);
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
        while (pend4() > guard) {
            store_row<0>(lds, wg_out);
            rows_done(1);
        }
    }
    __device__ __forceinline__ void flush_all(char *lds, u8 *wg_out) {  // per lane
        while (pend4() >= 16u) {
            store_row<0>(lds, wg_out);
            rows_done(1);
        }
    }
    __device__ __forceinline__ u64 finish(char *lds, u8 *wg_out) {  // per lane; returns the stream length in bits
        flush_all(lds, wg_out);
        u8 *piece = wg_out + goff;           // the piece that is still filling: its words go in from the top
        const u32 nw = pend4() >> 2;         // < 4 now
        u32 a = th4;
        for (u32 j = 0; j < nw; ++j) {
            *reinterpret_cast<u32 *>(piece + 12u - 4u * j) = *reinterpret_cast<const u32 *>(lds + base + a);
            a = (a - 4u) & 127u;
        }
        const u32 n = 32u - room;  // <= 32 after the last check(): the top n bits of hi, zero bits in front of them
        // (nw <= 3; with n != 0 and nw == 3 the word lands at piece + 0)
        if (n) *reinterpret_cast<u32 *>(piece + 12u - 4u * nw) = __builtin_bswap32(n == 32 ? hi : (hi >> (32 - n)));
        return (u64)(((goff0 - goff) >> 10) * 4u + nw) * 32 + n;
    }
};

// Wave-cooperative store of one decoded 128-byte line per lane.
// A lane that stores its own line issues eight 16-byte stores; the 64 lanes of a store instruction hit 64 different
// lines, and every 16-byte piece travels to L2 as its own write request (6.7e7 of them per GiB: TCP_TCC_WRITE_REQ in
// profiles/r01_v6_pmc_summary.txt).  When all 64 lanes of the wave are at the same position of equally long chunks
// (the batch case) the eight registers of a line are transposed across the lanes l, l+8, ..., l+56 -- three butterfly
// stages: lanes l ^ 8 by DPP (row_ror:8 under a bank mask), l ^ 16 and l ^ 32 by v_permlane16_swap /
// v_permlane32_swap, which exchange the odd 16-lane rows (32-lane halves) of their first operand with the even ones
// of their second: one instruction per pair of dwords -- after which those eight lanes hold the eight 16-byte pieces
// of ONE line and every store instruction writes eight whole lines.  rANS decode of 1 GiB: 0.75 -> 0.60 ms, 1.68e7
// write requests (profiles/r01_coop_store_note.txt).  Ragged batches and partial waves keep the per-lane bursts.
__device__ __forceinline__ void scl_swap8(u32 &p, u32 &q) {
    // banks 0-1 = lanes 0..7 of each 16-lane row, banks 2-3 = lanes 8..15; row_ror:8 reads lane ^ 8
    const u32 q2 = (u32)__builtin_amdgcn_update_dpp((int)q, (int)p, 0x128, 0xF, 0x3, false);  // lanes 0..7: q = p of lane ^ 8
    p = (u32)__builtin_amdgcn_update_dpp((int)p, (int)q, 0x128, 0xF, 0xC, false);            // lanes 8..15: p = q of lane ^ 8
    q = q2;
}
__device__ __forceinline__ void scl_swap16(u32 &p, u32 &q) {
    const auto t = __builtin_amdgcn_permlane16_swap(p, q, false, false);
    p = t[0];
    q = t[1];
}
__device__ __forceinline__ void scl_swap32(u32 &p, u32 &q) {
    const auto t = __builtin_amdgcn_permlane32_swap(p, q, false, false);
    p = t[0];
    q = t[1];
}
// 8x8 transpose of 16-byte elements across the lanes l, l+8, ..., l+56 (all 64 lanes active): register j of lane
// (l, k) gets what register k of lane (l, j) held
__device__ __forceinline__ void scl_transpose8(uint4 *a) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        scl_swap8(a[r].x, a[r + 1].x);
        scl_swap8(a[r].y, a[r + 1].y);
        scl_swap8(a[r].z, a[r + 1].z);
        scl_swap8(a[r].w, a[r + 1].w);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r & 2) continue;
        scl_swap16(a[r].x, a[r + 2].x);
        scl_swap16(a[r].y, a[r + 2].y);
        scl_swap16(a[r].z, a[r + 2].z);
        scl_swap16(a[r].w, a[r + 2].w);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        scl_swap32(a[r].x, a[r + 4].x);
        scl_swap32(a[r].y, a[r + 4].y);
        scl_swap32(a[r].z, a[r + 4].z);
        scl_swap32(a[r].w, a[r + 4].w);
    }
}
struct CoopLineStore {
    u8 *base;  // lane (l0, k) = (lane & 7, lane >> 3) writes piece k of the lines of the lanes l0 + 8 j
    u64 stride;
    bool on;
    // `key`: any per-lane value that decides how many lines the lane will store and when (the caller's loop counter):
    // the wave cooperates iff all 64 lanes are present with the same non-zero key.  Rows are out_stride apart.
    __device__ __forceinline__ void init(u8 *out_sym, u64 c, u64 out_stride, u32 key) {
        const u32 lane = threadIdx.x & 63u;
        on = __builtin_amdgcn_ballot_w64(key != 0 && key == (u32)__builtin_amdgcn_readfirstlane((int)key)) == ~0ull;
        base = out_sym + (c - lane + (lane & 7u)) * out_stride + 16u * (lane >> 3);
        stride = out_stride;
    }
    // a[b] = bytes [16 b, 16 b + 16) of this lane's line, which starts at byte `pos` of its row
    __device__ __forceinline__ void store(uint4 *a, u32 pos) const {
        scl_transpose8(a);
#pragma unroll
        // non-temporal: a decoded line is written once, whole, and never read here.  (With per-lane 16-byte pieces the
        // same hint was a disaster -- they then reach memory unmerged; with whole lines it leaves decode unchanged and
        // makes the kernel that runs NEXT 2.5 % faster: less dirty data in L2 when it starts.)
        for (int j = 0; j < 8; ++j) {
            typedef u32 u32x4_nt __attribute__((ext_vector_type(4)));
            const u32x4_nt t = {a[j].x, a[j].y, a[j].z, a[j].w};
            __builtin_nontemporal_store(t, reinterpret_cast<u32x4_nt *>(base + (u64)(8 * j) * stride + pos));
        }
    }
};

struct Line128 {
    uint4 v[8];
    __device__ __forceinline__ void load(const uint4 *p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[i];
    }
};

// ZERO_PAST_END: bits after the last bit of the stream read as 0 whatever the buffer holds there (the arithmetic
// decoder's state register looks 32 bits ahead, arithmetic_coding.py:222-229, :258-261); the ANS decoders never
// look at what they do not consume and leave it off.
template <int THREADS, bool ZERO_PAST_END = false>
struct AnsBitReader {
    static constexpr u32 RING_BYTES = 32u * THREADS * 4u;
    u32 wabs;      // ZERO_PAST_END: index (from the buffer start) of the next word that enters the ring
    u32 end_word;  //                index of the word holding the first bit after the stream
    u32 end_mask;  //                bits of that word which still belong to the stream
    const uint4 *base;
    u64 n_blocks16;  // readable 16-byte blocks
    u64 next_line;   // index of the next 128-byte line to prefetch
    uint4 pf[8];     // prefetched line: its two 64-byte halves enter the ring one at a time
    u32 stage;       // 0: the lower half of pf is next, 1: the upper half
    u32 ra;          // LDS byte address of the next ring word to read (thread column, wraps inside the ring)
    u32 wa;          // LDS byte address of the ring half that is filled next
    u32 nwr;         // words written to the ring (the words READ follow from it and the two ring positions: nrd())
    u32 A, B;        // 64-bit window, big-endian words
    u32 sh;          // lookahead = low32((A:B) >> sh); sh in [0,31].  Unsigned on purpose: advance() takes the BORROW of
                     // its subtraction as "the window crossed a word" (v_sub_co + branch on VCC, no compare)
    u32 bias;        // consumed bits = 32*nrd() - sh - bias

    __device__ __forceinline__ void load_line(u64 j) {  // whole 128-byte line: one HBM burst instead of two
        // Every line but the buffer's last is readable as a whole: ONE comparison and eight loads off one address.  (Eight
        // separately bounds-checked loads -- compare, zero default, branch, each -- were 70 instructions per line, a fifth
        // of the rANS decoder's vector instructions.)
        // (not for ZERO_PAST_END users: the static-model arithmetic decoder sits at its register limit and spills with it)
        if (!ZERO_PAST_END && j * 8 + 8 <= n_blocks16) {
            const uint4 *p = base + j * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) pf[i] = p[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u64 idx = j * 8 + i;
                pf[i] = (idx < n_blocks16) ? base[idx] : make_uint4(0, 0, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void push_half(char *lds, const uint4 &q0, const uint4 &q1, const uint4 &q2,
                                              const uint4 &q3) {
        char *r = lds + wa;
        const uint4 blk[4] = {q0, q1, q2, q3};
        if (ZERO_PAST_END && wabs + 16 > end_word) {  // the half that holds the end of the stream, or lies past it
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 w4[4] = {blk[i].x, blk[i].y, blk[i].z, blk[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32 wi = wabs + 4 * i + j;
                    const u32 keep = (wi < end_word) ? 0xFFFFFFFFu : (wi == end_word ? end_mask : 0u);
                    *reinterpret_cast<u32 *>(r + (4 * i + j) * THREADS * 4) = __builtin_bswap32(w4[j]) & keep;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32 *>(r + (4 * i + 0) * THREADS * 4) = __builtin_bswap32(blk[i].x);
                *reinterpret_cast<u32 *>(r + (4 * i + 1) * THREADS * 4) = __builtin_bswap32(blk[i].y);
                *reinterpret_cast<u32 *>(r + (4 * i + 2) * THREADS * 4) = __builtin_bswap32(blk[i].z);
                *reinterpret_cast<u32 *>(r + (4 * i + 3) * THREADS * 4) = __builtin_bswap32(blk[i].w);
            }
        }
        if (ZERO_PAST_END) wabs += 16;
        wa ^= 16 * THREADS * 4;
        nwr += 16;
    }
    __device__ __forceinline__ u32 next_word(const char *lds) {
        const u32 v = *reinterpret_cast<const u32 *>(lds + ra);
        ra = (ra + THREADS * 4) & (RING_BYTES - 1);
        return v;
    }
    // words still ahead of the reader, minus one, in ring-address units: `wa` is the half that is filled next, i.e.
    // the row after the newest word; the reader never runs dry, so a distance of 0 rows means 32
    __device__ __forceinline__ u32 ahead_m1() const { return (wa - ra - THREADS * 4) & (RING_BYTES - 1); }
    __device__ __forceinline__ u32 nrd() const { return nwr - (ahead_m1() / (THREADS * 4) + 1); }
    // call at least every 16 symbols (<= 6 words consumed in between)
    __device__ __forceinline__ void maybe_refill(char *lds) {
        if (ahead_m1() < 16 * THREADS * 4) {  // 16 words or fewer ahead: the reader has left the older half
            if (stage == 0) {
                push_half(lds, pf[0], pf[1], pf[2], pf[3]);
                stage = 1;
            } else {
                push_half(lds, pf[4], pf[5], pf[6], pf[7]);
                stage = 0;
                load_line(next_line++);
            }
        }
    }
    __device__ __forceinline__ void init(const u8 *in, u64 in_size_bytes, u64 bit_off, char *lds, u32 tid,
                                         u32 nbits = 0) {
        base = reinterpret_cast<const uint4 *>(in);
        n_blocks16 = in_size_bytes >> 4;
        const u64 j0 = bit_off >> 10;
        if (ZERO_PAST_END) {  // (streams and buffers are below 2^37 bits: word indices fit 32 bits)
            const u64 end = bit_off + nbits;
            wabs = (u32)(j0 * 32);
            end_word = (u32)(end >> 5);
            end_mask = ~(0xFFFFFFFFu >> (end & 31u)) & (0u - (u32)((end & 31u) != 0));
        }
        wa = tid * 4;
        nwr = 0;
        load_line(j0);
        push_half(lds, pf[0], pf[1], pf[2], pf[3]);
        push_half(lds, pf[4], pf[5], pf[6], pf[7]);
        load_line(j0 + 1);
        next_line = j0 + 2;
        stage = 0;
        const u32 w0 = (u32)(bit_off >> 5) & 31u;
        ra = tid * 4 + w0 * THREADS * 4;
        // a stream that starts in the upper half of its line has fewer than 17 words ahead of it: top the ring
        // up before the first word is read (the lower half of the ring is already behind the read position)
        maybe_refill(lds);
        const u32 pos = (u32)bit_off & 31u;
        const u32 first = next_word(lds);
        if (pos == 0) {
            A = 0;
            B = first;
            sh = 0;
        } else {
            A = first;
            B = next_word(lds);
            sh = 32 - pos;
        }
        bias = 32 * nrd() - sh;  // consumed == 0 here
    }
    __device__ __forceinline__ u32 consumed() const { return 32 * nrd() - sh - bias; }
    __device__ __forceinline__ u32 look() const { return __builtin_amdgcn_alignbit(A, B, sh); }
    __device__ __forceinline__ u32 look(const char *) const { return look(); }
    __device__ __forceinline__ void advance(const char *lds, u32 nb) {  // nb <= 32
        if (__builtin_usub_overflow(sh, nb, &sh)) {
            A = B;
            B = next_word(lds);
            sh += 32;
        }
    }
    __device__ __forceinline__ u32 get(const char *lds, u32 w) {  // 1 <= w <= 32
        const u32 v = look() >> (32 - w);
        advance(lds, w);
        return v;
    }
};

// Round-4 reader: no bit window in registers.  AnsBitReader keeps two ring words (A, B) in registers and advances
// them under a branch that the whole wave runs for every pair of symbols (some lane always crosses a word: 5 VALU + 2 SALU
// + the branch per pair).  Here the lookahead of a pair is read straight out of the ring by BIT POSITION: the two words
// around the position (two conflict-free ds_read_b32 of the lane's column) and one v_alignbit -- no branch, no window to
// shift, two VALU instructions less per pair (the rANS decoder issues 12.8 VALU per symbol at 77-80 % of one per four
// clocks: instruction COUNT is what bounds it).
//   N = -(bits consumed since bit 0 of the lane's first 128-byte line): v_alignbit reads its low five bits, which is the
//       shift that brings the position's bit to the top -- for a position on a word boundary that shift is 0 and the
//       result is the SECOND word, so the first word read is the one holding the bit BEFORE the position:
//   P = (position - 1) << RSH: P & ROWMASK is that word's ring row (byte address without the thread column).
// Same ring, same line-granular refill and the same call cadence as AnsBitReader.
template <int THREADS, bool ZERO_PAST_END = false>
struct AnsBitReaderW {
    static constexpr u32 RING_BYTES = 32u * THREADS * 4u;
    static constexpr u32 ROW = THREADS * 4u;
    static constexpr u32 RSH = THREADS == 1024 ? 7u : (THREADS == 512 ? 6u : (THREADS == 256 ? 5u : 4u));  // log2(ROW) - 5
    static constexpr u32 ROWMASK = 31u * ROW;
    static_assert((32u << RSH) == ROW, "THREADS must be 128, 256, 512 or 1024");
    u32 wabs;      // ZERO_PAST_END: index (from the buffer start) of the next word that enters the ring
    u32 end_word;  //                index of the word holding the first bit after the stream
    u32 end_mask;  //                bits of that word which still belong to the stream
    const uint4 *base;
    u64 n_blocks16;  // readable 16-byte blocks
    u64 next_line;   // index of the next 128-byte line to prefetch
    uint4 pf[8];     // prefetched line: its two 64-byte halves enter the ring one at a time
    u32 stage;       // 0: the lower half of pf is next, 1: the upper half
    u32 wa;          // LDS byte address of the ring half that is filled next
    u32 N, P;        // see above
    u32 col;         // tid * 4
    u32 rowA;        // LDS byte address of the first word the last look() read: rows from here on are still needed
    u32 start;       // bit position at init

    __device__ __forceinline__ void load_line(u64 j) {  // whole 128-byte line: one HBM burst instead of two
        // Every line but the buffer's last is readable as a whole: ONE comparison and eight loads off one address.  (Eight
        // separately bounds-checked loads -- compare, zero default, branch, each -- were 70 instructions per line, a fifth
        // of the rANS decoder's vector instructions.)
        // (not for ZERO_PAST_END users: the static-model arithmetic decoder sits at its register limit and spills with it)
        if (!ZERO_PAST_END && j * 8 + 8 <= n_blocks16) {
            const uint4 *p = base + j * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) pf[i] = p[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u64 idx = j * 8 + i;
                pf[i] = (idx < n_blocks16) ? base[idx] : make_uint4(0, 0, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void push_half(char *lds, const uint4 &q0, const uint4 &q1, const uint4 &q2,
                                              const uint4 &q3) {
        char *r = lds + wa;
        const uint4 blk[4] = {q0, q1, q2, q3};
        if (ZERO_PAST_END && wabs + 16 > end_word) {  // the half that holds the end of the stream, or lies past it
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 w4[4] = {blk[i].x, blk[i].y, blk[i].z, blk[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32 wi = wabs + 4 * i + j;
                    const u32 keep = (wi < end_word) ? 0xFFFFFFFFu : (wi == end_word ? end_mask : 0u);
                    *reinterpret_cast<u32 *>(r + (4 * i + j) * ROW) = __builtin_bswap32(w4[j]) & keep;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32 *>(r + (4 * i + 0) * ROW) = __builtin_bswap32(blk[i].x);
                *reinterpret_cast<u32 *>(r + (4 * i + 1) * ROW) = __builtin_bswap32(blk[i].y);
                *reinterpret_cast<u32 *>(r + (4 * i + 2) * ROW) = __builtin_bswap32(blk[i].z);
                *reinterpret_cast<u32 *>(r + (4 * i + 3) * ROW) = __builtin_bswap32(blk[i].w);
            }
        }
        if (ZERO_PAST_END) wabs += 16;
        wa ^= 16 * ROW;
    }
    // rows from the first one still needed up to the newest, minus one, in ring-address units.  The writer may fill the
    // half at `wa` once at most 16 rows are left ahead of rowA: that half then lies wholly behind the reader.  A check
    // that finds 17 rows and does nothing still leaves 4 after the <= 13 words of the next 32 symbols -- the position's
    // two words and one to spare.
    __device__ __forceinline__ u32 ahead_m1() const { return (wa - rowA - ROW) & (RING_BYTES - 1); }
    // call at least every 32 symbols of <= 13 bits
    __device__ __forceinline__ void maybe_refill(char *lds) {
        if (ahead_m1() < 16 * ROW) {
            if (stage == 0) {
                push_half(lds, pf[0], pf[1], pf[2], pf[3]);
                stage = 1;
            } else {
                push_half(lds, pf[4], pf[5], pf[6], pf[7]);
                stage = 0;
                load_line(next_line++);
            }
        }
    }
    __device__ __forceinline__ void init(const u8 *in, u64 in_size_bytes, u64 bit_off, char *lds, u32 tid,
                                         u32 nbits = 0) {
        base = reinterpret_cast<const uint4 *>(in);
        n_blocks16 = in_size_bytes >> 4;
        const u64 j0 = bit_off >> 10;
        if (ZERO_PAST_END) {  // (streams and buffers are below 2^37 bits: word indices fit 32 bits)
            const u64 end = bit_off + nbits;
            wabs = (u32)(j0 * 32);
            end_word = (u32)(end >> 5);
            end_mask = ~(0xFFFFFFFFu >> (end & 31u)) & (0u - (u32)((end & 31u) != 0));
        }
        col = tid * 4;
        wa = col;
        load_line(j0);
        push_half(lds, pf[0], pf[1], pf[2], pf[3]);
        push_half(lds, pf[4], pf[5], pf[6], pf[7]);
        load_line(j0 + 1);
        next_line = j0 + 2;
        stage = 0;
        start = (u32)bit_off & 1023u;
        N = 0u - start;
        P = (start - 1u) << RSH;
        rowA = ((start >> 5) * ROW) | col;  // the word of the position itself (the one before it is never needed again)
        // a stream that starts in the upper half of its line has fewer than 17 words ahead of it: top the ring up before
        // the first word is read (the lower half of the ring is already behind the read position)
        maybe_refill(lds);
    }
    __device__ __forceinline__ u32 consumed() const { return (0u - N) - start; }
    // the 32 bits at the position (the ring always holds the two words around it)
    __device__ __forceinline__ u32 look(const char *lds) {
        const u32 a = (P & ROWMASK) | col;            // v_and_or
        const u32 b = ((P + ROW) & ROWMASK) | col;    // v_add + v_and_or
        rowA = a;
        const u32 A = *reinterpret_cast<const u32 *>(lds + a);
        const u32 B = *reinterpret_cast<const u32 *>(lds + b);
        return __builtin_amdgcn_alignbit(A, B, N);
    }
    __device__ __forceinline__ void advance(const char *, u32 nb) {  // any nb
        N -= nb;
        // one v_lshl_add; written out because the compiler re-associates P into (initial P) + (running sum << RSH), an add
        // more per pair
        asm("v_lshl_add_u32 %0, %1, %2, %0" : "+v"(P) : "v"(nb), "n"(RSH));
    }
    __device__ __forceinline__ u32 get(const char *lds, u32 w) {  // 1 <= w <= 32
        const u32 v = look(lds) >> (32 - w);
        advance(lds, w);
        return v;
    }
};

// Round-6 reader for WAVE-STRIPED slots (the layout AnsBackWriterT writes; scl_stripe_byte).  Same windowless look() /
// advance() as AnsBitReaderW over the same [word][thread] ring of 32 words -- what changes is the refill: the unit is the
// 16-byte PIECE, not the 128-byte line.  Piece q of lane l lives at wave_slot + 1024 q + 16 l, so when the lanes of a wave
// fetch "their next piece" the addresses of neighbouring lanes are adjacent -- lanes within the same row read ONE
// contiguous run -- where the linear layout made every lane's load touch a different line (64 lines per instruction).
// Refill point (every 32 symbols of <= 13 bits, like AnsBitReaderW): first the pieces requested at the PREVIOUS refill
// point enter the ring (they have had 32 symbols' time to arrive), then every lane requests as many pieces as its ring
// will have room for: free = 32 - ahead words -> free / 4 pieces, at most four.  Level: after the arrivals a lane holds
// >= 15 words (13 for the next 32 symbols + the two around the position) because at the previous point it requested
// floor(free / 4) pieces, i.e. ahead + 4 * requested >= 29.
// Loads never leave the lane's slot: the offset of the next piece is clamped to the last piece (a stream that claims more
// bits than it has re-reads its tail and is flagged TRUNCATED by the caller's length check).
template <int THREADS>
struct AnsBitReaderT {
    static constexpr u32 RING_BYTES = 32u * THREADS * 4u;
    static constexpr u32 ROW = THREADS * 4u;
    static constexpr u32 RSH = THREADS == 1024 ? 7u : (THREADS == 512 ? 6u : (THREADS == 256 ? 5u : 4u));  // log2(ROW) - 5
    static constexpr u32 ROWMASK = 31u * ROW;
    static_assert((32u << RSH) == ROW, "THREADS must be 128, 256, 512 or 1024");
    const u8 *wbase;  // the wave's striped slot + 16 * lane
    u32 goff;         // byte offset (from wbase) of the next piece to request
    u32 glast;        // ... of the lane's last piece
    uint4 pf[4];      // pieces requested at the previous refill point
    u32 npf;          // how many of them
    u32 wa;           // LDS byte address of the ring row filled next (a multiple of four rows | col)
    u32 N, P;         // as AnsBitReaderW
    u32 col;          // tid * 4
    u32 rowA;         // LDS byte address of the first word the last look() read
    u32 start;        // bit position at init (inside the first piece loaded: < 128)

    __device__ __forceinline__ uint4 fetch() {
        const uint4 q = *reinterpret_cast<const uint4 *>(wbase + goff);
        goff = min(goff + 1024u, glast);
        return q;
    }
    __device__ __forceinline__ void push_piece(char *lds, const uint4 &q) {
        char *r = lds + wa;
        *reinterpret_cast<u32 *>(r) = __builtin_bswap32(q.x);
        *reinterpret_cast<u32 *>(r + ROW) = __builtin_bswap32(q.y);
        *reinterpret_cast<u32 *>(r + 2 * ROW) = __builtin_bswap32(q.z);
        *reinterpret_cast<u32 *>(r + 3 * ROW) = __builtin_bswap32(q.w);
        wa = (wa + 4 * ROW) & (RING_BYTES - 1);
    }
    __device__ __forceinline__ u32 ahead_m1() const { return (wa - rowA - ROW) & (RING_BYTES - 1); }
    // call at least every 32 symbols of <= 13 bits
    __device__ __forceinline__ void maybe_refill(char *lds) {
#pragma unroll
        for (u32 r = 0; r < 4; ++r)
            if (npf > r) push_piece(lds, pf[r]);
        npf = min((31u - (ahead_m1() >> (RSH + 5u))) >> 2, 4u);  // free words / 4 (<= 4 anyway: >= 15 words are ahead here)
#pragma unroll
        for (u32 r = 0; r < 4; ++r)
            if (npf > r) pf[r] = fetch();
    }
    // in: the batch's output buffer; stride: bytes per logical slot (a multiple of 16); c: this lane's chunk; bit_off: the
    // stream's LOGICAL bit offset (inside slot c)
    __device__ __forceinline__ void init(const u8 *in, u64 stride, u64 c, u64 bit_off, char *lds, u32 tid) {
        const u32 rel = (u32)(bit_off - c * stride * 8);  // < 2^32: slots are below 512 MiB
        wbase = in + (c >> 6) * (stride << 6) + ((c & 63u) << 4);
        glast = (u32)stride * 64u - 1024u;
        goff = min((rel >> 7) * 1024u, glast);
        col = tid * 4;
        wa = col;
        uint4 q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = fetch();
#pragma unroll
        for (int i = 0; i < 8; ++i) push_piece(lds, q[i]);  // wa is back at row 0: the ring is full
        npf = 0;
        start = rel & 127u;
        N = 0u - start;
        P = (start - 1u) << RSH;
        rowA = ((start >> 5) * ROW) | col;
    }
    __device__ __forceinline__ u32 consumed() const { return (0u - N) - start; }
    __device__ __forceinline__ u32 look(const char *lds) {
        const u32 a = (P & ROWMASK) | col;
        const u32 b = ((P + ROW) & ROWMASK) | col;
        rowA = a;
        const u32 A = *reinterpret_cast<const u32 *>(lds + a);
        const u32 B = *reinterpret_cast<const u32 *>(lds + b);
        return __builtin_amdgcn_alignbit(A, B, N);
    }
    __device__ __forceinline__ void advance(const char *, u32 nb) {
        N -= nb;
        asm("v_lshl_add_u32 %0, %1, %2, %0" : "+v"(P) : "v"(nb), "n"(RSH));
    }
    __device__ __forceinline__ u32 get(const char *lds, u32 w) {  // 1 <= w <= 32
        const u32 v = look(lds) >> (32 - w);
        advance(lds, w);
        return v;
    }
};

// Forward twin of AnsBackWriter for streams that grow front to back (arithmetic coder): completed big-endian
// words go to the per-lane LDS ring ([word][thread], at LDS offset 0) and leave for memory as whole 128-byte lines
// (the first 64-byte half waits in registers), so that a lane only ever stores whole, aligned lines.
template <int THREADS>
struct AnsFwdWriter {
    static constexpr u32 RING_BYTES = 32u * THREADS * 4u;
    u32 hi;    // pending bits, right-aligned (the oldest is the most significant), < 32 of them
    u32 nacc;  // number of pending bits
    u32 ra;    // LDS byte address of the ring word that completes next
    u32 fa;    // LDS byte address of the oldest unflushed word
    u32 pend;  // completed words not yet stored to memory
    u32 nfl;   // words already stored to memory (a held half line counts as stored)
    u8 *slot;
    uint4 held[4];
    u32 have_held;

    __device__ __forceinline__ void init(u32 tid, u8 *slot_) {
        hi = 0;
        nacc = 0;
        ra = tid * 4;
        fa = tid * 4;
        pend = 0;
        nfl = 0;
        slot = slot_;
        have_held = 0;
        held[0] = held[1] = held[2] = held[3] = make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ void put(char *lds, u32 v, u32 w) {  // v < 2^w, w <= 32
        const u32 tot = nacc + w;
        if (tot >= 32) {
            const u32 r = tot - 32;  // <= 31
            // 32-bit arithmetic on purpose.  The obvious (u32)(t >> r) on a 64-bit t compiled to v_lshrrev_b64 with
            // a just-computed VGPR shift amount and, at full occupancy, stored a wrong word about once in 10^8
            // (tools/stress_fast_kernels.py: ~3 words per 1 GiB encode held the bits accumulated BEFORE this call);
            // this form has been stress-tested clean.  w - r = 32 - nacc is in [1, 32); r == 0 means w == 32 - nacc.
            const u32 word = (r == 0) ? ((hi << (w & 31)) | v) : ((hi << (w - r)) | (v >> r));
            *reinterpret_cast<u32 *>(lds + ra) = __builtin_bswap32(word);
            ra = (ra + THREADS * 4) & (RING_BYTES - 1);
            ++pend;
            hi = v & ((1u << r) - 1u);
            nacc = r;
        } else {
            hi = (hi << w) | v;  // tot < 32, so w < 32
            nacc = tot;
        }
    }
    // put for the per-symbol field of the arithmetic encoders (w <= 32; w = 0 allowed with v = 0): pending bits and field side
    // by side in 64 bits, only the completed word is conditional -- no case split on which half the word comes from and no
    // values merged after a branch.  (The round-1 fault blamed on this 64-bit form was a register spill --
    // profiles/r03_fwd_writer_fault_analysis.txt; no kernel spills now, tests/test_no_scratch.py.)
    __device__ __forceinline__ void put_field(char *lds, u32 v, u32 w) {
        const u32 tot = nacc + w;              // <= 63
        const u64 wide = ((u64)hi << w) | v;   // tot valid bits, right-aligned
        nacc = tot & 31u;
        if (tot >= 32) {
            *reinterpret_cast<u32 *>(lds + ra) = __builtin_bswap32((u32)(wide >> nacc));  // tot - 32 = tot & 31 here
            ra = (ra + THREADS * 4) & (RING_BYTES - 1);
            ++pend;
        }
        hi = (u32)wide & ((1u << nacc) - 1u);
    }
    // 16 pending words leave the ring at a time; call after at most 16 new words
    __device__ __forceinline__ void maybe_flush(char *lds) {
        if (pend >= 16) {
            const char *r = lds + fa;
            u32 w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = *reinterpret_cast<const u32 *>(r + j * THREADS * 4);
            const uint4 q0 = make_uint4(w[0], w[1], w[2], w[3]), q1 = make_uint4(w[4], w[5], w[6], w[7]);
            const uint4 q2 = make_uint4(w[8], w[9], w[10], w[11]), q3 = make_uint4(w[12], w[13], w[14], w[15]);
            if (have_held) {  // second half of the line whose first half is held
                uint4 *p = reinterpret_cast<uint4 *>(slot + 4 * (u64)(nfl - 16));
                p[0] = held[0];
                p[1] = held[1];
                p[2] = held[2];
                p[3] = held[3];
                p[4] = q0;
                p[5] = q1;
                p[6] = q2;
                p[7] = q3;
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
            fa ^= 16 * THREADS * 4;
        }
    }
    __device__ __forceinline__ void put_run(char *lds, u32 bit, u32 count) {  // `count` copies of `bit`
        while (count >= 32) {
            put(lds, bit ? 0xFFFFFFFFu : 0u, 32);
            maybe_flush(lds);
            count -= 32;
        }
        if (count) put(lds, bit ? ((1u << count) - 1u) : 0u, count);
    }
    __device__ __forceinline__ u64 finish(char *lds) {  // returns the stream length in bits
        maybe_flush(lds);
        if (have_held) {
            uint4 *p = reinterpret_cast<uint4 *>(slot + 4 * (u64)(nfl - 16));
            p[0] = held[0];
            p[1] = held[1];
            p[2] = held[2];
            p[3] = held[3];
        }
        u32 *dst = reinterpret_cast<u32 *>(slot) + nfl;
        u32 a = fa;
        for (u32 j = 0; j < pend; ++j) {
            dst[j] = *reinterpret_cast<const u32 *>(lds + a);
            a = (a + THREADS * 4) & (RING_BYTES - 1);
        }
        if (nacc) dst[pend] = __builtin_bswap32(hi << (32 - nacc));  // zero bits behind the stream
        return (u64)(nfl + pend) * 32 + nacc;
    }
};
