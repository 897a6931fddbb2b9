// Microbenchmark: what one SIMD of gfx950 issues per cycle for the integer vector instructions the k-mer
// kernels are made of, at 1 / 2 / 4 / 8 waves per SIMD on every CU.  The ruler for `issue_roofline` in bench.py.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue > profiles/r03_valu_issue.txt
//
// Every wave runs ITERS x 128 instructions of one kind (16 independent accumulators, inline asm so that nothing is
// folded); a workgroup is 4 x wps waves (one per SIMD and wave slot) and LDS is requested so that exactly one
// (wps <= 4) or two (wps = 8: 2 x 1024 threads) workgroups fit a CU.  Reported per kind and occupancy:
//   cycles per wave-instruction per SIMD = (shader cycles of the loop) / (ITERS * 128 * wps)
// (shader cycles from s_memtime; the clock from s_memtime / s_memrealtime, 100 MHz), and the chip-wide rate
// 256 CUs x 4 SIMDs x clock / that.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

enum Kind { ADD32, XOR32, LSHL_OR, AND_OR, BFE, ALIGNBIT, BFREV, CNDMASK, MUL_LO, LSHL64, LSHR64, CMP64_CND, ADD64, MAD64, POPC, FFBL, NKINDS };
static const char* kind_name[NKINDS] = {"v_add_u32", "v_xor_b32", "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_alignbit_b32",
                                        "v_bfrev_b32", "v_cndmask_b32", "v_mul_lo_u32", "v_lshlrev_b64", "v_lshrrev_b64",
                                        "v_cmp_lt_u64+cndmask", "v_add_co+addc (64-bit add)", "v_mad_u64_u32", "v_bcnt_u32_b32", "v_ffbl_b32"};
// how many wave instructions one "op" below is
static const int kind_insts[NKINDS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1};

template <int KIND>
__device__ __forceinline__ void op(u32& a, u32& b, u32 c, u32 d) {
    if (KIND == ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == XOR32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a) : "v"(c));
    else if (KIND == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d));
    else if (KIND == BFE) asm volatile("v_bfe_u32 %0, %0, 2, 7" : "+v"(a));
    else if (KIND == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(a) : "v"(c));
    else if (KIND == BFREV) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a));
    else if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c));
    else if (KIND == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c));
    else if (KIND == LSHL64) { u64 x = ((u64)b << 32) | a; asm volatile("v_lshlrev_b64 %0, 2, %0" : "+v"(x)); a = (u32)x; b = (u32)(x >> 32); }
    else if (KIND == LSHR64) { u64 x = ((u64)b << 32) | a; asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(x)); a = (u32)x; b = (u32)(x >> 32); }
    else if (KIND == CMP64_CND) {
        u64 x = ((u64)b << 32) | a, y = ((u64)d << 32) | c;
        asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : : "v"(x), "v"(y), "v"(a), "v"(c) : "vcc");
    } else if (KIND == ADD64) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "vcc");
    else if (KIND == MAD64) { u64 x = ((u64)b << 32) | a; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(c), "v"(d) : "vcc"); a = (u32)x; b = (u32)(x >> 32); }
    else if (KIND == POPC) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(c));
    else if (KIND == FFBL) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a));
}

template <int KIND>
__global__ __launch_bounds__(1024) void kern(u32* out, u64* cyc, u64* real, u64* t_begin, u64* t_end, u32* hwid, int iters) {
    extern __shared__ u32 lds[];
    u32 a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = a[i] ^ 0x5bd1e995u; }
    const u32 c = threadIdx.x | 1u, d = blockIdx.x + 3u;
    __syncthreads();
    const u64 r0 = __builtin_amdgcn_s_memrealtime();
    const u64 t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {   // 128 independent-in-groups-of-16 instructions per trip: the branch back is amortised
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) op<KIND>(a[i], b[i], c, d);
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    const u64 r1 = __builtin_amdgcn_s_memrealtime();
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= a[i] ^ b[i];
    if (acc == 0x12345u) out[0] = acc + lds[threadIdx.x & 7];
    if ((threadIdx.x & 63) == 0) {
        const u32 w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        cyc[w] = t1 - t0;
        real[w] = r1 - r0;
        t_begin[w] = r0;
        t_end[w] = r1;
        hwid[w] = (__builtin_amdgcn_s_getreg(63492) & 0xffffu) | (__builtin_amdgcn_s_getreg(63508) << 16);   // HW_ID | XCC_ID
    }
}

template <int KIND>
static void run(int wps, int iters, u32* out, u64* cyc, u64* real, u64* tb, u64* te, u32* hw) {
    int dev = 0, cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int per_cu = wps == 8 ? 2 : 1;
    const int threads = 64 * 4 * (wps == 8 ? 4 : wps);
    const size_t lds = wps == 8 ? 70 * 1024 : 100 * 1024;   // one / two workgroups per CU
    const int blocks = cus * per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int waves = blocks * threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(threads), lds, 0, out, cyc, real, tb, te, hw, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<u64> hc(waves), hr(waves);
    hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), real, waves * 8, hipMemcpyDeviceToHost);
    std::sort(hc.begin(), hc.end());
    std::sort(hr.begin(), hr.end());
    const double cycles = (double)hc[waves / 2], realt = (double)hr[waves / 2];   // medians over waves
    const double clock_ghz = cycles / realt * 0.1;
    const double insts = (double)iters * 128 * kind_insts[KIND];
    const double cpi_simd = cycles / (insts * wps);                                 // cycles one SIMD spends per wave instruction (wps waves share it)
    std::vector<u64> hb(waves), he(waves);
    std::vector<u32> hh(waves);
    hipMemcpy(hb.data(), tb, waves * 8, hipMemcpyDeviceToHost);
    hipMemcpy(he.data(), te, waves * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hh.data(), hw, waves * 4, hipMemcpyDeviceToHost);
    const u64 span = *std::max_element(he.begin(), he.end()) - *std::min_element(hb.begin(), hb.end());
    std::vector<u32> simds;   // distinct (xcc, se, sh, cu, simd)
    for (u32 v : hh) simds.push_back(((v >> 16) << 16) | (v & 0xff30u));
    std::sort(simds.begin(), simds.end());
    const size_t nsimd = std::unique(simds.begin(), simds.end()) - simds.begin();
    const double chip = (double)cus * 4 * clock_ghz / cpi_simd;                     // G wave-instructions / s
    const double chip_event = (double)waves * insts / (ms * 1e-3) / 1e9;            // the same from the HIP events (incl. launch ramp)
    printf("%-28s wps %d  %6.3f cycles/wave-instr/SIMD  clock %.2f GHz  chip %7.1f G wave-instr/s (HIP events: %7.1f)  "
           "waves %d on %zu SIMDs, wave time %.0f us of %.0f us span, kernel %.0f us\n",
           kind_name[KIND], wps, cpi_simd, clock_ghz, chip, chip_event, waves, nsimd, realt * 0.01, (double)span * 0.01, ms * 1e3);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

static u64 *g_tb, *g_te;
static u32* g_hw;
template <int KIND> static void sweep(u32* out, u64* cyc, u64* real) {
    for (int wps : {1, 2, 4, 8}) run<KIND>(wps, 4000, out, cyc, real, g_tb, g_te, g_hw);
}

int main() {
    u32* out;
    u64 *cyc, *real;
    hipMalloc(&out, 64);
    hipMalloc(&cyc, 8 * 65536);
    hipMalloc(&real, 8 * 65536);
    hipMalloc(&g_tb, 8 * 65536);
    hipMalloc(&g_te, 8 * 65536);
    hipMalloc(&g_hw, 4 * 65536);
    printf("# gfx950 vector-instruction issue: cycles per wave64 instruction per SIMD, by waves per SIMD (every CU busy)\n");
    sweep<ADD32>(out, cyc, real);
    sweep<XOR32>(out, cyc, real);
    sweep<LSHL_OR>(out, cyc, real);
    sweep<AND_OR>(out, cyc, real);
    sweep<BFE>(out, cyc, real);
    sweep<ALIGNBIT>(out, cyc, real);
    sweep<BFREV>(out, cyc, real);
    sweep<CNDMASK>(out, cyc, real);
    sweep<MUL_LO>(out, cyc, real);
    sweep<LSHL64>(out, cyc, real);
    sweep<LSHR64>(out, cyc, real);
    sweep<CMP64_CND>(out, cyc, real);
    sweep<ADD64>(out, cyc, real);
    sweep<MAD64>(out, cyc, real);
    sweep<POPC>(out, cyc, real);
    sweep<FFBL>(out, cyc, real);
    return 0;
}
