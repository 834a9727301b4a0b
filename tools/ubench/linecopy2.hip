// linecopy2.hip -- variations of the lane-per-chunk line pattern (see linecopy.hip): what limits it?
//   A: baseline read-only / write-only (per-lane 8 x 16 B per 128-byte line)
//   B: chunk stride padded by 128 B (is it the power-of-two stride?)
//   C: wave-cooperative access to the SAME lines (8 lanes per line: one instruction = 8 whole lines)
//   D: two lines (256 B) per lane per access
//   E: non-temporal loads / stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ uint4 ld(const uint4 *p) {
    if (NT) { u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
    return *p;
}
template <int NT> __device__ __forceinline__ void st(uint4 *p, uint4 v) {
    if (NT) { u32x4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p)); } else *p = v;
}

// per-lane: LPA lines per access
template <int WRITE, int LPA, int NT>
__global__ void __launch_bounds__(256, 4) per_lane(uint4 *__restrict__ buf, u64 n_chunks, u64 stride16, u32 n_lines, u32 *sink) {
    const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    uint4 *p = buf + c * stride16;
    uint4 acc = make_uint4(c, 2, 3, 4);
    for (u32 j = 0; j < n_lines; j += LPA) {
        if (WRITE) {
            for (int i = 0; i < 8 * LPA; ++i) st<NT>(p + j * 8 + i, make_uint4(acc.x + i, j, c, i));
        } else {
            uint4 v[8 * LPA];
            for (int i = 0; i < 8 * LPA; ++i) v[i] = ld<NT>(p + j * 8 + i);
            for (int i = 0; i < 8 * LPA; ++i) { acc.x ^= v[i].x; acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w; }
        }
    }
    if (!WRITE && acc.x == 0x12345678 && acc.y == 0x9abcdef0) sink[0] = acc.z + acc.w;
}

// cooperative: the wave's 64 chunks, line j: instruction i (0..7) serves chunks 8i..8i+7 of the wave, lane 8q+p = piece p of chunk 8i+q
template <int WRITE, int NT>
__global__ void __launch_bounds__(256, 4) coop(uint4 *__restrict__ buf, u64 n_chunks, u64 stride16, u32 n_lines, u32 *sink) {
    const u64 c0 = (u64)blockIdx.x * 256 + (threadIdx.x & ~63u);
    if (c0 >= n_chunks) return;
    const u32 lane = threadIdx.x & 63u, q = lane >> 3, pc = lane & 7u;
    uint4 acc = make_uint4(lane, 2, 3, 4);
    for (u32 j = 0; j < n_lines; ++j) {
        if (WRITE) {
            for (int i = 0; i < 8; ++i) st<NT>(buf + (c0 + 8 * i + q) * stride16 + j * 8 + pc, make_uint4(acc.x + i, j, q, i));
        } else {
            uint4 v[8];
            for (int i = 0; i < 8; ++i) v[i] = ld<NT>(buf + (c0 + 8 * i + q) * stride16 + j * 8 + pc);
            for (int i = 0; i < 8; ++i) { acc.x ^= v[i].x; acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w; }
        }
    }
    if (!WRITE && acc.x == 0x12345678 && acc.y == 0x9abcdef0) sink[0] = acc.z + acc.w;
}

// cooperative with the lane grouping of scl_transpose8: lane (l0 = lane & 7, k = lane >> 3) = piece k of chunk l0 + 8 i
template <int WRITE>
__global__ void __launch_bounds__(256, 4) coop_s8(uint4 *__restrict__ buf, u64 n_chunks, u64 stride16, u32 n_lines, u32 *sink) {
    const u64 c0 = (u64)blockIdx.x * 256 + (threadIdx.x & ~63u);
    if (c0 >= n_chunks) return;
    const u32 lane = threadIdx.x & 63u, l0 = lane & 7u, k = lane >> 3;
    uint4 acc = make_uint4(lane, 2, 3, 4);
    for (u32 j = 0; j < n_lines; ++j) {
        if (WRITE) {
            for (int i = 0; i < 8; ++i) st<0>(buf + (c0 + 8 * i + l0) * stride16 + j * 8 + k, make_uint4(acc.x + i, j, k, i));
        } else {
            uint4 v[8];
            for (int i = 0; i < 8; ++i) v[i] = ld<0>(buf + (c0 + 8 * i + l0) * stride16 + j * 8 + k);
            for (int i = 0; i < 8; ++i) { acc.x ^= v[i].x; acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w; }
        }
    }
    if (!WRITE && acc.x == 0x12345678 && acc.y == 0x9abcdef0) sink[0] = acc.z + acc.w;
}

// quad-cooperative 64-byte half lines: 4 adjacent lanes write 64 contiguous bytes of ONE chunk; 4 rounds per half line
// step (round r serves chunk 4Q + r of quad Q); ACTIVE_PCT of the (quad, round) pairs really store
template <int ACTIVE_PCT>
__global__ void __launch_bounds__(256, 4) quad_halflines(uint4 *__restrict__ buf, u64 n_chunks, u64 stride16, u32 n_lines, u32 *sink) {
    const u64 c0 = (u64)blockIdx.x * 256 + (threadIdx.x & ~3u);
    if (c0 >= n_chunks) return;
    const u32 j4 = threadIdx.x & 3u;
    u32 rng = (u32)(c0 * 2654435761u) | 1u;
    // every chunk gets 2 * n_lines half lines in total; each step a (quad, round) pair stores with probability
    // ACTIVE_PCT, so the number of steps is scaled to write the same bytes
    u32 done[4] = {0, 0, 0, 0};
    const u32 target = 2 * n_lines;
    for (u32 step = 0; step < target * 100 / ACTIVE_PCT + 64; ++step) {
        for (int r = 0; r < 4; ++r) {
            rng = rng * 1664525u + 1013904223u;
            const bool act = ((rng >> 8) % 100u) < (u32)ACTIVE_PCT && done[r] < target;
            if (act) {
                st<0>(buf + (c0 + r) * stride16 + done[r] * 4 + j4, make_uint4(step, r, j4, rng));
                ++done[r];
            }
        }
    }
}

template <typename F> static float time_ms(F f, int reps = 10) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const u64 n_chunks = 262144;
    uint4 *buf; u32 *sink;
    (void)hipMalloc(&buf, n_chunks * 6400 + 65536); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, n_chunks * 6400);
    const u32 blocks = n_chunks / 256;
#define T(NAME, KERNEL, STRIDE, LINES)                                                                     \
    {                                                                                                      \
        float ms = time_ms([&] { hipLaunchKernelGGL((KERNEL), dim3(blocks), dim3(256), 0, 0, buf, n_chunks, (u64)(STRIDE) / 16, (u32)(LINES), sink); }); \
        printf("%-58s stride %5d lines %2d : %.3f ms  %.2f TB/s\n", NAME, (int)(STRIDE), (int)(LINES), ms, (double)n_chunks * (LINES) * 128 / ms / 1e9); \
    }
    T("read  per-lane", (per_lane<0, 1, 0>), 4096, 32)
    T("read  per-lane, padded stride", (per_lane<0, 1, 0>), 4224, 32)
    T("read  per-lane, stride 6272", (per_lane<0, 1, 0>), 6272, 32)
    T("read  per-lane, non-temporal", (per_lane<0, 1, 1>), 4096, 32)
    T("read  per-lane, 2 lines per access", (per_lane<0, 2, 0>), 4096, 32)
    T("read  per-lane, 4 lines per access", (per_lane<0, 4, 0>), 4096, 32)
    T("read  cooperative (8 lanes per line)", (coop<0, 0>), 4096, 32)
    T("read  cooperative, non-temporal", (coop<0, 1>), 4096, 32)
    T("write per-lane", (per_lane<1, 1, 0>), 6272, 29)
    T("write per-lane, stride 4096", (per_lane<1, 1, 0>), 4096, 29)
    T("write per-lane, non-temporal", (per_lane<1, 1, 1>), 6272, 29)
    T("write per-lane, 2 lines per access", (per_lane<1, 2, 0>), 6272, 28)
    T("write cooperative (8 lanes per line)", (coop<1, 0>), 6272, 29)
    T("write cooperative, non-temporal", (coop<1, 1>), 6272, 29)
    T("read  cooperative, lanes l0 + 8k (transpose8 grouping)", (coop_s8<0>), 4096, 32)
    T("write cooperative, lanes l0 + 8k (transpose8 grouping)", (coop_s8<1>), 6272, 29)
    T("write quad half-lines (4 lanes x 16 B), all active", (quad_halflines<100>), 6272, 29)
    T("write quad half-lines, 46 % of (quad, round) active", (quad_halflines<46>), 6272, 29)
    return 0;
}
