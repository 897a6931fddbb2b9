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

// 16 ASCII bases -> 16 two-bit codes + 16 "not ACGTacgt" flags, four bytes at a time (SWAR):
//   code = ((c >> 1) ^ (c >> 2)) & 3 maps A C G T (either case) to 0 1 2 3; the letter that code stands for is
//   0x41 + {0, 2, 6, 0x13}[code], and a byte is a base iff its case-folded value equals that letter.
// The two multiplies gather the four 2-bit codes / the four flags of a word into its top byte / nibble
// (the partial products land on distinct bits: no carries).
__device__ __forceinline__ void decode4(const u32 x, u32& codes8, u32& bad4) {
    const u32 t = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    codes8 = (t * 0x01041040u) >> 24;
    const u32 b0 = t & 0x01010101u, b1 = (t >> 1) & 0x01010101u, both = b0 & b1;
    const u32 letter = 0x41414141u + ((b0 | b1) << 1) + ((b1 & ~b0) << 2) + both + (both << 4);
    const u32 z = (x & 0xDFDFDFDFu) ^ letter;
    const u32 nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;   // bit 7 of every byte that differs
    bad4 = ((nz >> 7) * 0x10204080u) >> 28;
}
__device__ __forceinline__ void decode16(const uint4 v, u32& codes, u32& bad) {
    u32 c0, c1, c2, c3, f0, f1, f2, f3;
    decode4(v.x, c0, f0);
    decode4(v.y, c1, f1);
    decode4(v.z, c2, f2);
    decode4(v.w, c3, f3);
    codes = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    bad = f0 | (f1 << 4) | (f2 << 8) | (f3 << 12);
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
