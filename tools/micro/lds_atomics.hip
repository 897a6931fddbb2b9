// Microbenchmark: cycles per workgroup for 4096 LDS atomics (512 threads x 8, random addresses),
// by instruction kind.  hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;
template <int MODE>
__global__ __launch_bounds__(512, 4) void kern(const u32* idx, u64* out, u64* cyc, int reps) {
    __shared__ u64 tbl[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) tbl[i] = ~0ull;
    u32 a[8];
    for (int e = 0; e < 8; ++e) a[e] = idx[(blockIdx.x * 8 + e) * 512 + threadIdx.x] & 8191;
    __syncthreads();
    u64 acc = 0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        u64 o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const u32 s = (a[e] + r * 977) & 8191;
            u32* t32 = reinterpret_cast<u32*>(tbl);
            if (MODE == 0) o[e] = atomicAdd(&t32[s], 1u);                               // ds_add_rtn_u32
            else if (MODE == 1) o[e] = atomicCAS(&t32[s], 0xffffffffu, a[e] + r);       // ds_cmpst_rtn_b32
            else if (MODE == 2) o[e] = atomicCAS(&tbl[s & 8191], ~0ull, (u64)a[e] + r); // ds_cmpst_rtn_b64
            else if (MODE == 3) { atomicOr(&tbl[s], 1ull << (e + r)); o[e] = 0; }        // ds_or_b64 (no return)
            else if (MODE == 4) { atomicOr(&t32[s], 1u << (e + r)); o[e] = 0; }          // ds_or_b32 (no return)
            else if (MODE == 5) o[e] = atomicMin(&tbl[s], (u64)a[e] + r);                // ds_min_rtn_u64
            else if (MODE == 6) o[e] = atomicAdd(&tbl[s], 1ull);                         // ds_add_rtn_u64
            else if (MODE == 7) o[e] = tbl[s];                                            // ds_read_b64
            else if (MODE == 8) o[e] = atomicMin(&t32[s], a[e] + r);                      // ds_min_rtn_u32
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += o[e];
    }
    __syncthreads();
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    const int blocks = 512, reps = 16;
    std::vector<u32> h(blocks * 8 * 512);
    u32 x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    u32* d; u64 *o, *c;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 8); hipMalloc(&c, blocks * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[] = {"add_rtn_u32", "cmpst_rtn_b32", "cmpst_rtn_b64", "or_b64", "or_b32", "min_rtn_u64", "add_rtn_u64", "read_b64", "min_rtn_u32"};
    for (int m = 0; m < 9; ++m) {
        for (int it = 0; it < 2; ++it) {
            switch (m) {
#define L(M) case M: hipLaunchKernelGGL(kern<M>, dim3(blocks), dim3(512), 0, 0, d, o, c, reps); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8)
            }
            hipDeviceSynchronize();
        }
        std::vector<u64> hc(blocks);
        hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : hc) s += v;
        printf("%-14s %8.0f cycles per workgroup per 4096 ops (avg over %d WGs, 2 per CU)\n", names[m], s / blocks / reps, blocks);
    }
    return 0;
}
