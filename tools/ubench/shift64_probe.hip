// probe for the rare wrong-word fault seen in the first AnsFwdWriter::put (profiles/r01_fwd_writer_fault_note.txt):
// (u32)(t >> r) on a 64-bit t with a just-computed VGPR shift amount, against the 32-bit formulation, at full
// occupancy, with the surrounding pattern of the original (LDS ring write in a divergent branch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef uint64_t u64;
__global__ void __launch_bounds__(256, 4) probe(u32 *bad, u32 *sink, u32 iters, u32 seed) {
    __shared__ u32 ring[32 * 256];
    u32 s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
    u32 hi = 0, nacc = 0, ra = threadIdx.x, acc = 0, nbad = 0;
    for (u32 i = 0; i < iters; ++i) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const u32 w = 1 + (s & 15);               // 1..16 new bits
        const u32 v = (s >> 8) & ((1u << w) - 1u);
        const u64 t = ((u64)hi << w) | v;
        const u32 tot = nacc + w;
        if (tot >= 32) {
            const u32 r = tot - 32;
            const u32 w64 = (u32)(t >> r);                                   // the suspicious form
            const u32 w32 = (r == 0) ? ((hi << (w & 31)) | v) : ((hi << (w - r)) | (v >> r));
            ring[ra] = __builtin_bswap32(w64);
            ra = (ra + 256) & (32 * 256 - 1);
            nbad += (w64 != w32);
            acc ^= w64;
            hi = (u32)t & ((1u << r) - 1u);
            nacc = r;
        } else {
            hi = (u32)t;
            nacc = tot;
        }
        if ((i & 63) == 63) acc ^= ring[(ra + 256 * 7) & (32 * 256 - 1)];
    }
    if (nbad) atomicAdd(bad, nbad);
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    u32 *bad, *sink; (void)hipMalloc(&bad, 4); (void)hipMalloc(&sink, 4096 * 256 * 4); (void)hipMemset(bad, 0, 4);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, bad, sink, 20000u, 77u + rep);
    (void)hipDeviceSynchronize();
    u32 h = 0; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("shift64 probe: %u mismatches in %.2e word formations\n", h, 20.0 * 4096 * 256 * 20000 * 0.27);
    return 0;
}
