// Microbenchmark: NREC records of 16 bytes scattered to NS slot regions, position taken with a
// returning global atomicAdd on the slot's cursor (what a one-level super-k-mer scatter would do).
// Variants: cursor stride (4 B packed / 64 B padded), atomics only, stores only.
// hipcc --offload-arch=gfx950 -O3 -o global_scatter global_scatter.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;

__device__ __forceinline__ u32 mixu(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// MODE 0: atomic + store, 1: atomic only, 2: store only (position from the record number)
template <int MODE>
__global__ __launch_bounds__(256) void kern(u32* cur, u32 cstride, uint4* recs, u32 ns, u32 cap, u32 per_thread, u32* sink) {
    const u32 gid = blockIdx.x * 256 + threadIdx.x;
    u32 acc = 0;
    for (u32 i = 0; i < per_thread; ++i) {
        const u32 id = gid * per_thread + i;
        const u32 slot = (u32)(((u64)mixu(id * 2654435761u + 12345u) * ns) >> 32);
        u32 pos;
        if (MODE == 2) pos = mixu(id) % cap;
        else pos = atomicAdd(&cur[(u64)slot * cstride], 1u);
        if (MODE != 1) { if (pos < cap) recs[(u64)slot * cap + pos] = make_uint4(id, slot, pos, i); }
        else acc += pos;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
int main() {
    const u32 nrec = 23u << 20, per_thread = 24;
    const u32 blocks = nrec / (256 * per_thread);
    for (u32 ns : {73000u, 146000u}) {
        const u32 cap = 2 * (nrec / ns) + 64;
        u32 *cur, *sink; uint4* recs;
        hipMalloc(&cur, (size_t)ns * 64); hipMalloc(&sink, 64); hipMalloc(&recs, (size_t)ns * cap * 16);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (u32 stride : {1u, 16u}) {
            for (int m = 0; m < 3; ++m) {
                float best = 1e9f;
                for (int it = 0; it < 4; ++it) {
                    hipMemset(cur, 0, (size_t)ns * 64);
                    hipDeviceSynchronize();
                    hipEventRecord(a, 0);
                    if (m == 0) hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(256), 0, 0, cur, stride, recs, ns, cap, per_thread, sink);
                    if (m == 1) hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(256), 0, 0, cur, stride, recs, ns, cap, per_thread, sink);
                    if (m == 2) hipLaunchKernelGGL(kern<2>, dim3(blocks), dim3(256), 0, 0, cur, stride, recs, ns, cap, per_thread, sink);
                    hipEventRecord(b, 0);
                    hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (it && ms < best) best = ms;
                }
                printf("ns=%u cursor stride %2u B  %-12s %.3f ms for %.1f M records\n", ns, stride * 4,
                       m == 0 ? "atomic+store" : m == 1 ? "atomic only" : "store only", best, nrec / 1e6);
            }
        }
        hipFree(cur); hipFree(sink); hipFree(recs);
    }
    return 0;
}
