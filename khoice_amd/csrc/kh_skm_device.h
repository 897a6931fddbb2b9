// khoice_amd — device helpers shared by the two super-k-mer kernel files (kh_skm.hip: one-word keys,
// kh_skm2.hip: two-word keys): minimizer hashing, sliding minimum, the LDS counting-sort flush, code-word prefetch.
#pragma once
#include <hip/hip_runtime.h>

#include "kh_device.h"
#include "kh_launch.h"

namespace {

constexpr u32 SKM_NT = 256;                    // threads of a scatter workgroup
constexpr u32 SKM_PPT = 32;                    // k-mer start positions per thread and sub-tile
constexpr u32 SKM_SUB = SKM_NT * SKM_PPT;      // 8192 positions per sub-tile
constexpr u32 SKM_CW = (SKM_SUB + KH_HALO) / 16;
constexpr u32 SKM_CAP = KH_SKM_STAGE;          // records staged in LDS per flush of the scatter
constexpr int SKM_WMIN = 5, SKM_WMAX = 18;     // m-mers per k-mer the scatter is instantiated for
constexpr u32 SKM_RG_NT = 1024;                // regroup: one workgroup per coarse bucket
constexpr u32 SKM_RG_CAP = 8192;               // records per round of the regroup

__device__ __forceinline__ u32 revpairs32(u32 x) {
    x = __builtin_bitreverse32(x);
    return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}
__device__ __forceinline__ u32 mmer_hash(u32 canon) {   // order of the m-mers: a bijection on 32 bits
    u32 h = canon * 0x9E3779B1u;   // (one multiply: quarter rate, and this runs once per base)
    h ^= h >> 15;
    return h;
}
__device__ __forceinline__ u32 slot_of(u32 minv, u32 nslots) {   // slot of a minimizer: independent of its rank
    u32 x = minv * 0xC2B2AE35u;
    x ^= x >> 16;
    x *= 0x27D4EB2Fu;
    x ^= x >> 15;
    return (u32)(((u64)x * (u64)nslots) >> 32);
}
// the value of the next lane (lane 63 keeps `old`): DPP wave_shl:1, no LDS round trip
__device__ __forceinline__ u32 next_lane(u32 v, u32 old) {
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x130, 0xf, 0xf, false);
}
// v[s] for a run-time s < 32.  Per-lane register indexing does not exist; written as a tree of bit selects
// (v_bfi) so that the compiler does not turn it into a scratch array.
__device__ __forceinline__ u32 bsel(u32 m, u32 a, u32 b) { return (a & m) | (b & ~m); }   // m ? a : b, bitwise
__device__ __forceinline__ u32 pick32(const u32 (&v)[SKM_PPT], u32 s) {
    const u32 m0 = 0u - (s & 1u), m1 = 0u - ((s >> 1) & 1u), m2 = 0u - ((s >> 2) & 1u), m3 = 0u - ((s >> 3) & 1u),
              m4 = 0u - ((s >> 4) & 1u);
    u32 a[16], b[8], c[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = bsel(m0, v[2 * i + 1], v[2 * i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = bsel(m1, a[2 * i + 1], a[2 * i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = bsel(m2, b[2 * i + 1], b[2 * i]);
    return bsel(m4, bsel(m3, c[3], c[2]), bsel(m3, c[1], c[0]));
}

// Sliding minimum over windows of WW positions, in registers: in: cur[0 .. PPT + WW - 2], out: cur[j] =
// min(cur[j .. j + WW - 1]) for j < PPT.  Doubling up to the largest power of two L <= WW, then two
// overlapping windows of L.
template <int WW>
__device__ __forceinline__ void window_min(u32 (&cur)[SKM_PPT + WW - 1]) {
    constexpr int L = WW >= 32 ? 32 : (WW >= 16 ? 16 : (WW >= 8 ? 8 : (WW >= 4 ? 4 : (WW >= 2 ? 2 : 1))));
    constexpr int X = (int)SKM_PPT + WW - 1;   // extent
    // after the level of stride s, cur[i] = min over 2s positions for i < X - (2s - 1)
#pragma unroll
    for (int s = 1; s < L; s <<= 1) {
#pragma unroll
        for (int i = 0; i < X - (2 * s - 1); ++i) cur[i] = cur[i] < cur[i + s] ? cur[i] : cur[i + s];
    }
    constexpr int D = WW - L;
    if (D) {
#pragma unroll
        for (int j = 0; j < (int)SKM_PPT; ++j) cur[j] = cur[j] < cur[j + D] ? cur[j] : cur[j + D];
    }
}

struct FlushLds {
    uint4* stage;   // [CAP * RW / 4]: records of RW 32-bit words
    u16* sid;       // [CAP] bucket of every staged record
    u32* bcnt;      // [nbk] records per bucket (zero on entry to a round)
    u32* bstart;    // [nbk + 1]
    u32* gpos;      // [nbk]
    u32* wsum;      // [16]
};
template <u32 CAP, u32 RW = 4> __device__ __forceinline__ FlushLds flush_lds(u8* base, u32 nbk_alloc) {
    FlushLds L;
    L.stage = reinterpret_cast<uint4*>(base);
    L.sid = reinterpret_cast<u16*>(base + (size_t)CAP * 4 * RW);
    L.bcnt = reinterpret_cast<u32*>(base + (size_t)CAP * (4 * RW + 2));
    L.bstart = L.bcnt + nbk_alloc;
    L.gpos = L.bstart + nbk_alloc + 4;
    L.wsum = L.gpos + nbk_alloc;
    return L;
}
template <u32 CAP, u32 RW = 4> constexpr size_t flush_lds_bytes(u32 nbk_alloc) {
    return (size_t)CAP * (4 * RW + 2) + (size_t)(3 * nbk_alloc + 4 + 16) * 4;
}

// Counting sort of the n staged records by bucket inside LDS, then every bucket's run goes to its region,
// consecutive lanes storing consecutive records.  The run's position comes from ONE returning global
// atomic (LOCAL == false: the scatter, whose buckets are shared by all workgroups) or from a cursor in LDS
// (LOCAL: the regroup, where a workgroup owns its buckets).  nbk <= 2 * NT.
// Entry: a barrier has made stage / sid / bcnt visible.  Exit: bcnt zeroed, a barrier passed.
// Records of a slot whose region is full (regroup): they go to a global side list with the number of
// their slot; the slots they belong to are taken by a kernel of their own afterwards (k_skm_big).
struct SkmSpill {
    uint4* rec = nullptr;     // [cap] records (of one or two uint4)
    u32* slot = nullptr;      // [cap]
    u32* n = nullptr;         // records spilled so far (may run past cap: the host then falls back)
    u32 cap = 0;
    u32 first_slot = 0;       // global number of the workgroup's bucket 0
};
template <u32 NT, u32 CAP, bool LOCAL, u32 RW = 4>
__device__ __forceinline__ void skm_flush(const FlushLds& L, const u32 n, const u32 nbk, u32* cursors,
                                          uint4* __restrict__ region, const u32 region_cap, u32* __restrict__ ctl,
                                          const SkmSpill sp = SkmSpill()) {
    constexpr u32 CS = LOCAL ? 1u : KH_SKM_CUR1_STRIDE;   // words between two cursors
    constexpr int RPT = (int)(CAP / NT);
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    // ---- exclusive scan of the bucket counts (two buckets per thread), run reservation
    u32 c0 = 0, c1 = 0;
    const u32 b0 = 2 * tid, b1 = 2 * tid + 1;
    if (b0 < nbk) c0 = L.bcnt[b0];
    if (b1 < nbk) c1 = L.bcnt[b1];
    const u32 incl = wave_scan_add(c0 + c1);
    if (lane == KH_WAVE - 1) L.wsum[wid] = incl;
    if (c0) {
        u32 g;
        if (LOCAL) { g = cursors[b0]; cursors[b0] = g + c0; }
        else g = atomicAdd(&cursors[(size_t)b0 * CS], c0);
        L.gpos[b0] = g;
        if (g + c0 > region_cap && !sp.rec) atomicOr(ctl, KH_ERR_CAPACITY);
    }
    if (c1) {
        u32 g;
        if (LOCAL) { g = cursors[b1]; cursors[b1] = g + c1; }
        else g = atomicAdd(&cursors[(size_t)b1 * CS], c1);
        L.gpos[b1] = g;
        if (g + c1 > region_cap && !sp.rec) atomicOr(ctl, KH_ERR_CAPACITY);
    }
    // the staged records of this thread, into registers (they are placed in place)
    constexpr int Q = (int)(RW / 4);   // uint4 per record
    u32 rx[RPT * Q], ry[RPT * Q], rz[RPT * Q], rw[RPT * Q], bk[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const u32 i = tid + (u32)r * NT;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint4 v = L.stage[(i < n ? i : 0) * Q + q];
            rx[r * Q + q] = v.x; ry[r * Q + q] = v.y; rz[r * Q + q] = v.z; rw[r * Q + q] = v.w;
        }
        bk[r] = L.sid[i < n ? i : 0];
    }
    __syncthreads();
    if (!LOCAL) SKM_STAMP(12);
    u32 run = incl - (c0 + c1);
    for (u32 w = 0; w < wid; ++w) run += L.wsum[w];
    if (b0 < nbk) { L.bstart[b0] = run; L.bcnt[b0] = 0; }
    if (b1 < nbk) { L.bstart[b1] = run + c0; L.bcnt[b1] = 0; }
    __syncthreads();
    // ---- placement
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const u32 i = tid + (u32)r * NT;
        if (i < n) {
            const u32 at = L.bstart[bk[r]] + atomicAdd(&L.bcnt[bk[r]], 1u);
#pragma unroll
            for (int q = 0; q < Q; ++q) L.stage[at * Q + q] = make_uint4(rx[r * Q + q], ry[r * Q + q], rz[r * Q + q], rw[r * Q + q]);
            L.sid[at] = (u16)bk[r];
        }
    }
    __syncthreads();
    if (!LOCAL) SKM_STAMP(13);
    // ---- write-out
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const u32 i = tid + (u32)r * NT;
        if (i < n) {
            const u32 b = L.sid[i];
            const u32 dest = L.gpos[b] + (i - L.bstart[b]);
            if (dest < region_cap) {
#pragma unroll
                for (int q = 0; q < Q; ++q) region[((u64)b * region_cap + dest) * Q + q] = L.stage[i * Q + q];
            } else if (sp.rec) {   // the slot's region is full
                const u32 at = atomicAdd(sp.n, 1u);
                if (at < sp.cap) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) sp.rec[(u64)at * Q + q] = L.stage[i * Q + q];
                    sp.slot[at] = sp.first_slot + b;
                }
            }
        }
    }
    __syncthreads();
    if (b0 < nbk) L.bcnt[b0] = 0;
    if (b1 < nbk) L.bcnt[b1] = 0;
    __syncthreads();
}

// the code words of one sub-tile, in flight while the previous one is processed
struct SkmFetch { uint4 v[3]; u32 left[3]; };
__device__ __forceinline__ void skm_fetch(const u8* __restrict__ sbase, const u64 len, const u64 p0, SkmFetch& f) {
#pragma unroll
    for (u32 r = 0; r < 3; ++r) {
        const u32 w = threadIdx.x + r * SKM_NT;
        f.left[r] = 0;
        f.v[r] = make_uint4(0, 0, 0, 0);
        if (w < SKM_CW) {
            const u64 b0 = p0 + 16ull * w;
            if (b0 < len) {
                const u64 left = len - b0;
                f.left[r] = left >= 16 ? 16u : (u32)left;
                if (left >= 16) {
                    f.v[r] = *reinterpret_cast<const uint4*>(sbase + b0);
                } else {   // last, partial word of the sequence: never touch bytes past its end
                    u32 w4[4] = {0, 0, 0, 0};
                    for (u32 i = 0; i < (u32)left; ++i) w4[i >> 2] |= (u32)sbase[b0 + i] << (8 * (i & 3));
                    f.v[r] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
            }
        }
    }
}
__device__ __forceinline__ void skm_store(const SkmFetch& f, u32* code, u16* bad16) {
#pragma unroll
    for (u32 r = 0; r < 3; ++r) {
        const u32 w = threadIdx.x + r * SKM_NT;
        if (w < SKM_CW) {
            u32 codes = 0, bad = 0xffffu;
            if (f.left[r]) {
                decode16(f.v[r], codes, bad);
                if (f.left[r] < 16) bad |= (0xffffu << f.left[r]) & 0xffffu;
            }
            code[w] = codes;
            bad16[w] = (u16)bad;
        }
    }
}

}   // namespace
