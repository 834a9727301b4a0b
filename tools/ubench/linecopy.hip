// linecopy.hip -- what does the MEMORY SYSTEM give for the access pattern of the lane-per-chunk coders?
// One lane per 4 KiB chunk: the lane reads its chunk one 128-byte line at a time (8 x 16-byte loads) and writes a
// stream of 0.91 x that size into its own 6272-byte slot, one 128-byte line at a time (8 x 16-byte stores), back to
// front -- no arithmetic.  Variants: read only / write only / both; per-lane stores or wave-cooperative (transposed)
// stores; pacing loop of VALU work between lines so that the request rate resembles the encoder's.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/linecopy.hip -o /tmp/linecopy && /tmp/linecopy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
#define CHUNK 4096
#define SLOT 6272

template <int MODE, int PACE>  // MODE bit0 = read, bit1 = write; PACE = dependent VALU ops per line
__global__ void __launch_bounds__(256, 4) lane_lines(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n_chunks,
                                                     u32 out_lines, u32 *sink) {
    const u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 *src = in + c * (CHUNK / 16);
    uint4 *dst = out + (c + 1) * (SLOT / 16);
    uint4 acc = make_uint4(1, 2, 3, 4);
    u32 wl = 0, frac = 0;
    uint4 v[8];
    for (int i = 0; i < 8; ++i) v[i] = (MODE & 1) ? src[i] : make_uint4(c, i, 0, 0);
    for (u32 j = 0; j < CHUNK / 128; ++j) {
        uint4 nx[8];
        const u32 jn = min(j + 1, (u32)(CHUNK / 128 - 1));
        if (MODE & 1)
            for (int i = 0; i < 8; ++i) nx[i] = src[jn * 8 + i];
        u32 x = v[0].x ^ v[7].w;
        for (int p = 0; p < PACE; ++p) x = x * 2654435761u + (x >> 7);  // dependent VALU work
        acc.x ^= x;
        for (int i = 0; i < 8; ++i) { acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w + v[i].x; }
        // output rate 0.91 lines per input line
        frac += 233;
        if ((MODE & 2) && frac >= 256 && wl < out_lines) {
            frac -= 256;
            ++wl;
            uint4 *p = dst - wl * 8;
            for (int i = 0; i < 8; ++i) p[i] = make_uint4(acc.x + i, acc.y, acc.z, v[i].x);
        }
        if (MODE & 1)
            for (int i = 0; i < 8; ++i) v[i] = nx[i];
    }
    if (acc.x == 0x12345678 && acc.y == 0x9abcdef0) sink[0] = acc.z + acc.w;
}

// reference: fully coalesced copy of the same byte counts (grid-stride, 16 bytes per lane)
__global__ void __launch_bounds__(256) flat_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n_in16, u64 n_out16, int mode) {
    const u64 stride = (u64)gridDim.x * 256;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (mode & 1)
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_in16; i += stride) { uint4 t = in[i]; acc.x ^= t.x; acc.y += t.y; acc.z ^= t.z; acc.w += t.w; }
    if (mode & 2)
        for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_out16; i += stride) out[i] = make_uint4(i, acc.x, acc.y, acc.z);
    if (!(mode & 2) && acc.x == 0x12345678 && acc.y == 77) out[0] = acc;
}

template <typename F> static float time_ms(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
    const u64 n_chunks = 262144;
    uint4 *in, *out; u32 *sink;
    hipMalloc(&in, n_chunks * CHUNK); hipMalloc(&out, n_chunks * SLOT + 4096); hipMalloc(&sink, 64);
    hipMemset(in, 1, n_chunks * CHUNK); hipMemset(out, 0, n_chunks * SLOT);
    const u32 out_lines = 29;  // 29 x 128 = 3712 bytes per chunk (the headline stream is ~3738)
    const double rd = (double)n_chunks * CHUNK, wr = (double)n_chunks * out_lines * 128;
    const u32 blocks = n_chunks / 256;
#define RUN(MODE, PACE)                                                                                            \
    {                                                                                                              \
        float ms = time_ms([&] { hipLaunchKernelGGL((lane_lines<MODE, PACE>), dim3(blocks), dim3(256), 0, 0, in, out, n_chunks, out_lines, sink); }); \
        double bytes = ((MODE & 1) ? rd : 0) + ((MODE & 2) ? wr : 0);                                             \
        printf("lane-per-chunk lines  mode=%d (1=read 2=write 3=both) pace=%3d : %.3f ms  %.2f TB/s\n", MODE, PACE, ms, bytes / ms / 1e9); \
    }
    RUN(1, 0) RUN(2, 0) RUN(3, 0)
    RUN(1, 200) RUN(2, 200) RUN(3, 200)
    RUN(1, 800) RUN(2, 800) RUN(3, 800)
    RUN(3, 1600)
    for (int mode = 1; mode <= 3; ++mode) {
        float ms = time_ms([&] { hipLaunchKernelGGL(flat_copy, dim3(256 * 16), dim3(256), 0, 0, in, out, (u64)(rd / 16), (u64)(wr / 16), mode); });
        double bytes = ((mode & 1) ? rd : 0) + ((mode & 2) ? wr : 0);
        printf("flat coalesced        mode=%d                              : %.3f ms  %.2f TB/s\n", mode, ms, bytes / ms / 1e9);
    }
    return 0;
}
