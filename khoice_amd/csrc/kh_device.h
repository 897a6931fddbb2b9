// khoice_amd — device helpers shared by the gfx950 kernel files (kh_kernels.hip, kh_skm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "kh_common.h"

#define KH_WAVE 64

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & (KH_WAVE - 1); }

// Wave-wide inclusive scans on the DPP data path (row shifts inside the 16-lane rows, then the
// two row broadcasts of gfx9): six dependent VALU steps of a few cycles each, where the generic
// __shfl_up goes through ds_bpermute, i.e. pays an LDS round trip per step.  Lanes without a
// source read 0 (bound_ctrl) or keep the 0 passed as `old` (rows masked off).
#define KH_DPP_STEP(v, OP, ctrl, rmask, bctl) \
    v = OP(v, (u32)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xf, bctl))
__device__ __forceinline__ u32 kh_addu(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 kh_maxu(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 wave_scan_add(u32 v) {
    KH_DPP_STEP(v, kh_addu, 0x111, 0xf, true);    // row_shr:1
    KH_DPP_STEP(v, kh_addu, 0x112, 0xf, true);    // row_shr:2
    KH_DPP_STEP(v, kh_addu, 0x114, 0xf, true);    // row_shr:4
    KH_DPP_STEP(v, kh_addu, 0x118, 0xf, true);    // row_shr:8
    KH_DPP_STEP(v, kh_addu, 0x142, 0xa, false);   // row_bcast:15 into rows 1 and 3
    KH_DPP_STEP(v, kh_addu, 0x143, 0xc, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// maximum of the wave, valid in lane 63 (callers read it from there)
__device__ __forceinline__ u32 wave_scan_max(u32 v) {
    KH_DPP_STEP(v, kh_maxu, 0x111, 0xf, true);
    KH_DPP_STEP(v, kh_maxu, 0x112, 0xf, true);
    KH_DPP_STEP(v, kh_maxu, 0x114, 0xf, true);
    KH_DPP_STEP(v, kh_maxu, 0x118, 0xf, true);
    KH_DPP_STEP(v, kh_maxu, 0x142, 0xa, false);
    KH_DPP_STEP(v, kh_maxu, 0x143, 0xc, false);
    return v;
}

constexpr int KH_HALO = 96;                                   // k-1 <= 63 bases + two words of slack for the funnel shifts

// 16 ASCII bases -> 16 two-bit codes + 16 "not ACGTacgt" flags.
__device__ __forceinline__ void decode16(const uint4 v, u32& codes, u32& bad) {
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    codes = 0;
    bad = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const u32 c = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
        const u32 up = c & 0xdfu;                              // fold case
        const u32 code = ((c >> 1) ^ (c >> 2)) & 3u;            // A0 C1 G2 T3
        const bool ok = (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
        codes |= code << (2 * i);
        bad |= (ok ? 0u : 1u) << i;
    }
}

// Bases [p0, p0 + 16 * nwords) of a sequence -> LDS: code[w] = 16 two-bit codes (base j of the
// word at bits 2j), bad16[w] = 16 "breaks a k-mer" flags (not ACGTacgt, or past the end).
template <u32 NT>
__device__ __forceinline__ void load_codes(const u8* __restrict__ sbase, const u64 len, const u64 p0,
                                           u32* code, u16* bad16, const u32 nwords) {
    for (u32 w = threadIdx.x; w < nwords; w += NT) {
        const u64 b0 = p0 + 16ull * w;
        u32 codes = 0, bad = 0xffffu;
        if (b0 < len) {
            const u64 left = len - b0;
            uint4 v;
            if (left >= 16) {
                v = *reinterpret_cast<const uint4*>(sbase + b0);
            } else {   // last, partial word of the sequence: never touch bytes past its end
                u32 w4[4] = {0, 0, 0, 0};
                for (u32 i = 0; i < (u32)left; ++i) w4[i >> 2] |= (u32)sbase[b0 + i] << (8 * (i & 3));
                v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            decode16(v, codes, bad);
            if (left < 16) bad |= (0xffffu << (u32)left) & 0xffffu;
        }
        code[w] = codes;
        bad16[w] = (u16)bad;
    }
}

// reverse the order of the 32 two-bit groups of x
__device__ __forceinline__ u64 kh_revpairs64(u64 x) {
    x = ((u64)__builtin_bitreverse32((u32)x) << 32) | (u64)__builtin_bitreverse32((u32)(x >> 32));
    return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}
