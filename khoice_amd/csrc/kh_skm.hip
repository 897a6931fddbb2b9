// khoice_amd — the super-k-mer form of the fused experiment-type-1 path (gfx950, wave64).
//
// What it replaces: steps 1-8 of workflow/rules/exp_type_1.smk:156-259 when only the histograms and the
// per-genome distinct counts are wanted (kh_exp1_run without group_sets / across_set).  The key-array
// form (kh_kernels.hip: passes A-C + k_union_hash) moves every k-mer through HBM as an 8-byte key three
// times.  KMC itself does not do that: it bins *super-k-mers* by a minimizer signature.  Same idea here,
// laid out for this machine:
//
//   k_skm_scatter   bases -> hash of every canonical m-mer -> sliding minimum over the w = k-m+1 m-mers of
//                   a k-mer (the minimizer) -> slot = f(minimizer).  A k-mer and its reverse complement
//                   have the same canonical m-mers, hence the same slot; consecutive k-mers mostly share
//                   their minimizer, so a run of n of them travels as ONE 16-byte record (n + k - 1 bases
//                   at two bits, genome tag, n): ~2 bytes per k-mer instead of 8.  Records are counting-
//                   sorted by coarse bucket in LDS and flushed as runs (one global atomic per run).
//   k_skm_regroup   one coarse bucket at a time, 2048 records per workgroup: the same LDS counting sort
//                   by fine slot -> every slot of the key space is one contiguous record range.
//   k_skm_union     one workgroup per slot: records -> k-mers (balanced: a thread takes 8 consecutive
//                   k-mer indices of the slot, whatever records they fall in) -> canonical key -> LDS hash
//                   set {key, genome mask} -> popcount per group / number of groups -> histogram bins.
//                   A repeated (key, genome) pair is seen when its mask bit is already set: distinct
//                   k-mers of a genome = its valid k-mer instances - those repeats.
//
// Nothing here is ordered and nothing needs to be: equal k-mers only have to meet, and they do because the
// slot is a function of the k-mer as a set of m-mers.  All integer work, no MFMA.
#include <hip/hip_runtime.h>

#include "kh_device.h"
#include "kh_launch.h"

namespace {

constexpr u32 SKM_NT = 256;                    // threads of a scatter / regroup workgroup
constexpr u32 SKM_PPT = 32;                    // k-mer start positions per thread and sub-tile
constexpr u32 SKM_SUB = SKM_NT * SKM_PPT;      // 8192 positions per sub-tile
constexpr u32 SKM_CW = (SKM_SUB + KH_HALO) / 16;
constexpr u32 SKM_CAP = KH_SKM_STAGE;          // records staged in LDS per flush
constexpr u32 SKM_ROW = SKM_NT + 1;            // row stride of the transposed per-position arrays
constexpr u32 SKM_RPT = SKM_CAP / SKM_NT;      // staged records per thread in a flush
constexpr int SKM_MAXL = 16;                   // largest power-of-two window of m-mers (k <= 32, m >= 15: w <= 18)

__device__ __forceinline__ u32 revpairs32(u32 x) {
    x = __builtin_bitreverse32(x);
    return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}
__device__ __forceinline__ u32 mmer_hash(u32 canon) {   // order of the m-mers: a bijection on 32 bits
    u32 h = canon * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}
__device__ __forceinline__ u32 slot_of(u32 minv, u32 nslots) {   // slot of a minimizer: independent of its rank
    u32 x = minv * 0xC2B2AE35u;
    x ^= x >> 16;
    x *= 0x27D4EB2Fu;
    x ^= x >> 15;
    return (u32)(((u64)x * (u64)nslots) >> 32);
}

// Sliding minimum over windows of L (a power of two) positions by doubling, in registers:
// in: cur[0 .. PPT + L - 2], out: cur[j] = min(cur[j .. j + L - 1]) for j < PPT.
template <int L>
__device__ __forceinline__ void window_min(u32 (&cur)[SKM_PPT + SKM_MAXL - 1]) {
    // after the level of stride s, cur[i] = min over 2s positions for i < PPT + L - 2s
#pragma unroll
    for (int s = 1; s < L; s <<= 1) {
#pragma unroll
        for (int i = 0; i < (int)SKM_PPT + L - 2 * s; ++i) cur[i] = cur[i] < cur[i + s] ? cur[i] : cur[i + s];
    }
}

struct FlushLds {
    uint4* stage;   // [SKM_CAP]
    u16* sid;       // [SKM_CAP] bucket of every staged record
    u32* bcnt;      // [nbk] records per bucket (zero on entry to a round)
    u32* bstart;    // [nbk + 1]
    u32* gpos;      // [nbk]
    u32* wsum;      // [8]
};

// Counting sort of the n staged records by bucket inside LDS, then every bucket's run goes to its region
// (position from ONE returning global atomic per run), consecutive lanes storing consecutive records.
// Entry: a barrier has made stage / sid / bcnt visible.  Exit: bcnt zeroed, a barrier passed.
__device__ __forceinline__ void skm_flush(const FlushLds& L, const u32 n, const u32 nbk, u32* __restrict__ cursors,
                          uint4* __restrict__ region, const u32 region_cap, u32* __restrict__ ctl) {
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    // ---- exclusive scan of the bucket counts (two buckets per thread), run reservation
    u32 c0 = 0, c1 = 0;
    const u32 b0 = 2 * tid, b1 = 2 * tid + 1;
    if (b0 < nbk) c0 = L.bcnt[b0];
    if (b1 < nbk) c1 = L.bcnt[b1];
    const u32 incl = wave_scan_add(c0 + c1);
    if (lane == KH_WAVE - 1) L.wsum[wid] = incl;
    if (c0) {
        const u32 g = atomicAdd(&cursors[b0], c0);
        L.gpos[b0] = g;
        if (g + c0 > region_cap) atomicOr(ctl, KH_ERR_CAPACITY);
    }
    if (c1) {
        const u32 g = atomicAdd(&cursors[b1], c1);
        L.gpos[b1] = g;
        if (g + c1 > region_cap) atomicOr(ctl, KH_ERR_CAPACITY);
    }
    // the staged records of this thread, into registers (they are placed in place)
    u32 rx[SKM_RPT], ry[SKM_RPT], rz[SKM_RPT], rw[SKM_RPT], bk[SKM_RPT];
#pragma unroll
    for (int r = 0; r < (int)SKM_RPT; ++r) {
        const u32 i = tid + (u32)r * SKM_NT;
        const uint4 v = L.stage[i < n ? i : 0];
        rx[r] = v.x; ry[r] = v.y; rz[r] = v.z; rw[r] = v.w;
        bk[r] = L.sid[i < n ? i : 0];
    }
    __syncthreads();
    u32 run = incl - (c0 + c1);
    for (u32 w = 0; w < wid; ++w) run += L.wsum[w];
    if (b0 < nbk) { L.bstart[b0] = run; L.bcnt[b0] = 0; }
    if (b1 < nbk) { L.bstart[b1] = run + c0; L.bcnt[b1] = 0; }
    __syncthreads();
    // ---- placement
#pragma unroll
    for (int r = 0; r < (int)SKM_RPT; ++r) {
        const u32 i = tid + (u32)r * SKM_NT;
        if (i < n) {
            const u32 at = L.bstart[bk[r]] + atomicAdd(&L.bcnt[bk[r]], 1u);
            L.stage[at] = make_uint4(rx[r], ry[r], rz[r], rw[r]);
            L.sid[at] = (u16)bk[r];
        }
    }
    __syncthreads();
    // ---- write-out
#pragma unroll
    for (int r = 0; r < (int)SKM_RPT; ++r) {
        const u32 i = tid + (u32)r * SKM_NT;
        if (i < n) {
            const u32 b = L.sid[i];
            const u32 dest = L.gpos[b] + (i - L.bstart[b]);
            if (dest < region_cap) region[(u64)b * region_cap + dest] = L.stage[i];
        }
    }
    __syncthreads();
    if (b0 < nbk) L.bcnt[b0] = 0;
    if (b1 < nbk) L.bcnt[b1] = 0;
    __syncthreads();
}

__device__ __forceinline__ FlushLds flush_lds(u8* base, u32 nbk_alloc) {
    FlushLds L;
    L.stage = reinterpret_cast<uint4*>(base);
    L.sid = reinterpret_cast<u16*>(base + (size_t)SKM_CAP * 16);
    L.bcnt = reinterpret_cast<u32*>(base + (size_t)SKM_CAP * 18);
    L.bstart = L.bcnt + nbk_alloc;
    L.gpos = L.bstart + nbk_alloc + 4;
    L.wsum = L.gpos + nbk_alloc;
    return L;
}
constexpr size_t flush_lds_bytes(u32 nbk_alloc) { return (size_t)SKM_CAP * 18 + (size_t)(3 * nbk_alloc + 4 + 8) * 4; }

}   // namespace

size_t kh_skm_scatter_lds_bytes(u32 nb1) {
    const u32 nbk = (nb1 + 3) & ~3u;
    return flush_lds_bytes(nbk) + (size_t)SKM_CW * 4 + (((size_t)SKM_CW * 2 + 15) & ~(size_t)15) +
           (size_t)SKM_PPT * SKM_ROW * 4 + 96 * 4 + 64;
}
size_t kh_skm_regroup_lds_bytes(u32 S) { return flush_lds_bytes((S + 3) & ~3u) + 64; }

// ------------------------------------------------------------------------------------------
// S1: bases -> records, partitioned by coarse bucket
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SKM_NT, 2) void k_skm_scatter(const KhSkmJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const u32 nbk = (jb.nb1 + 3) & ~3u;
    const FlushLds L = flush_lds(lds_raw, nbk);
    u8* p = lds_raw + flush_lds_bytes(nbk);
    u32* code = reinterpret_cast<u32*>(p);                    p += (size_t)SKM_CW * 4;
    u16* bad16 = reinterpret_cast<u16*>(p);                   p += ((size_t)SKM_CW * 2 + 15) & ~(size_t)15;
    u32* hT = reinterpret_cast<u32*>(p);                      p += (size_t)SKM_PPT * SKM_ROW * 4;   // [PPT][NT + 1], transposed
    u32* tailh = reinterpret_cast<u32*>(p);                   p += 64 * 4;   // hashes of positions SUB .. SUB + 63
    u32* taila = reinterpret_cast<u32*>(p);                   p += 32 * 4;   // window minima of positions SUB .. SUB + 31
    u32* misc = reinterpret_cast<u32*>(p);                    // [0] staged records, [1] valid k-mers of the tile, [2..] scan scratch

    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const KhTile t = jb.tiles[blockIdx.x];
    const KhSeg sg = jb.segs[t.seg];
    const int k = jb.k, m = jb.m;
    const u32 w = jb.w, nslots = jb.nslots, S = jb.S, nmax = jb.nmax;
    u32 Lw = 1;
    while (2 * Lw <= w) Lw <<= 1;              // largest power of two <= w
    const u32 d = w - Lw;
    const u32 mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    const u64 tile_pos0 = (u64)t.tile_in_seg * jb.tile_pos;
    const int subtiles = (int)(jb.tile_pos / SKM_SUB);

    for (u32 i = tid; i < nbk; i += SKM_NT) L.bcnt[i] = 0;
    if (tid < 2) misc[tid] = 0;
    u32 staged = 0;   // uniform copy of misc[0]

    for (int sub = 0; sub < subtiles; ++sub) {
        const u64 p0 = tile_pos0 + (u64)sub * SKM_SUB;
        if (p0 >= sg.npos) break;   // uniform
        __syncthreads();
        load_codes<SKM_NT>(sg.seq, sg.len, p0, code, bad16, SKM_CW);
        __syncthreads();
        // ---- hashes of the m-mers starting at the thread's 32 positions (32-bit rolling words)
        u32 cw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cw[i] = code[2 * tid + i];
        {
            const u32 pm = (u32)(m - 1);
            const u32 pre = cw[0] & ((1u << (2 * pm)) - 1u);
            u32 f = revpairs32(pre) >> (32 - 2 * pm);
            u32 r = ((~pre) & ((1u << (2 * pm)) - 1u)) << 2;
            u32 nw[2];
            nw[0] = __builtin_amdgcn_alignbit(cw[1], cw[0], 2 * pm);
            nw[1] = __builtin_amdgcn_alignbit(cw[2], cw[1], 2 * pm);
#pragma unroll
            for (int j = 0; j < (int)SKM_PPT; ++j) {
                const u32 c = (nw[j >> 4] >> (2 * (j & 15))) & 3u;
                f = ((f << 2) | c) & mmask;
                r = (r >> 2) | ((3u - c) << (2 * pm));
                hT[(u32)j * SKM_ROW + tid] = mmer_hash(f < r ? f : r);
            }
        }
        if (tid < 64) {   // positions SUB .. SUB + 63, straight from the packed window
            const u32 q = SKM_SUB + tid, wq = q >> 4, oq = q & 15u;
            const u32 x0 = __builtin_amdgcn_alignbit(code[wq + 1], code[wq], 2 * oq);
            const u32 x = x0 & mmask;
            const u32 f = revpairs32(x) >> (32 - 2 * m);
            const u32 r = (~x) & mmask;
            tailh[tid] = mmer_hash(f < r ? f : r);
        }
        __syncthreads();
        // ---- minimizer of every k-mer: minimum over its w m-mers
        u32 cur[SKM_PPT + SKM_MAXL - 1];
#pragma unroll
        for (int j = 0; j < (int)SKM_PPT; ++j) cur[j] = hT[(u32)j * SKM_ROW + tid];
        {
            const bool last = tid == SKM_NT - 1;
            const u32* hal = last ? tailh : (hT + tid + 1);
            const u32 hs = last ? 1u : SKM_ROW;
#pragma unroll
            for (int j = 0; j < SKM_MAXL - 1; ++j) cur[SKM_PPT + j] = (u32)j < Lw - 1 ? hal[(u32)j * hs] : 0xffffffffu;
        }
        switch (Lw) {
            case 1: break;
            case 2: window_min<2>(cur); break;
            case 4: window_min<4>(cur); break;
            case 8: window_min<8>(cur); break;
            default: window_min<16>(cur); break;
        }
        if (d) {   // w is not a power of two: two overlapping windows of Lw
            if (tid < 32) {
                u32 v = 0xffffffffu;
                for (u32 i = 0; i < Lw; ++i) { const u32 x = tailh[tid + i]; v = x < v ? x : v; }
                taila[tid] = v;
            }
            __syncthreads();   // every thread has read its hashes
#pragma unroll
            for (int j = 0; j < (int)SKM_PPT; ++j) hT[(u32)j * SKM_ROW + tid] = cur[j];
            __syncthreads();
            const bool last = tid == SKM_NT - 1;
#pragma unroll
            for (int j = 0; j < (int)SKM_PPT; ++j) {
                const u32 jd = (u32)j + d;
                u32 o;
                if (jd < SKM_PPT) o = hT[jd * SKM_ROW + tid];
                else o = last ? taila[jd - SKM_PPT] : hT[(jd - SKM_PPT) * SKM_ROW + tid + 1];
                cur[j] = cur[j] < o ? cur[j] : o;
            }
        }
        // ---- which of the 32 start positions have k valid bases
        u32 vm;
        {
            u64 bm = (u64)bad16[2 * tid] | ((u64)bad16[2 * tid + 1] << 16) | ((u64)bad16[2 * tid + 2] << 32) |
                     ((u64)bad16[2 * tid + 3] << 48);
            u32 cover = 1;
            while (2 * cover <= (u32)k) { bm |= bm >> cover; cover <<= 1; }
            bm |= bm >> ((u32)k - cover);
            vm = ~(u32)bm;
        }
        // ---- slots; runs of valid positions with one slot become records
        __syncthreads();   // hT is re-used for the slots of the positions
        u32 cont = 0, prev = 0;
#pragma unroll
        for (int j = 0; j < (int)SKM_PPT; ++j) {
            const u32 s = slot_of(cur[j], nslots);
            hT[(u32)j * SKM_ROW + tid] = s;
            if (j && s == prev) cont |= 1u << j;
            prev = s;
        }
        cont &= vm & (vm << 1);
        u32 starts = vm & ~cont;
        auto run_len = [&](u32 s) -> u32 { return 1u + (u32)__builtin_ctzll(~((u64)cont >> (s + 1))); };
        u32 nrec = 0;
        {
            u32 st = starts;
            while (st) {
                const u32 s = (u32)__builtin_ctz(st);
                st &= st - 1;
                const u32 len = run_len(s);
                nrec += (len + nmax - 1) / nmax;
            }
        }
        {   // valid k-mer instances of the tile
            const u32 tot = wave_scan_add((u32)__popc(vm));
            if (lane == KH_WAVE - 1 && tot) atomicAdd(&misc[1], tot);
        }
        // ---- append to the staging array; a full array is flushed (a prefix of the threads fits)
        bool done = false;
        while (true) {
            const u32 mine = done ? 0u : nrec;
            const u32 incl = wave_scan_add(mine);
            if (lane == KH_WAVE - 1) misc[4 + wid] = incl;
            __syncthreads();
            u32 excl = incl - mine, total = 0;
            for (u32 q = 0; q < SKM_NT / 64; ++q) {
                const u32 v = misc[4 + q];
                excl += q < wid ? v : 0u;
                total += v;
            }
            const bool fits = staged + excl + mine <= SKM_CAP;
            if (!done && fits) {
                u32 at = staged + excl;
                u32 st = starts;
                while (st) {
                    const u32 s = (u32)__builtin_ctz(st);
                    st &= st - 1;
                    u32 len = run_len(s);
                    const u32 slot = hT[s * SKM_ROW + tid];
                    const u32 coarse = slot / S, fine = slot - coarse * S;
                    for (u32 s2 = s; len; ) {
                        const u32 n = len < nmax ? len : nmax;
                        const u64 lo = ((u64)cw[1] << 32) | cw[0], hi = ((u64)cw[3] << 32) | cw[2];
                        const u32 sh = 2 * s2;
                        u64 rlo = sh ? (lo >> sh) | ((hi << 1) << (63 - sh)) : lo;
                        u64 rhi = hi >> sh;
                        const u32 bits = 2 * (n + (u32)k - 1);
                        if (bits < 64) { rlo &= (1ull << bits) - 1ull; rhi = 0; }
                        else rhi &= kh_mask((int)bits - 64);
                        rhi |= ((u64)fine << 44) | ((u64)t.seg << 53) | ((u64)n << 59);
                        L.stage[at] = make_uint4((u32)rlo, (u32)(rlo >> 32), (u32)rhi, (u32)(rhi >> 32));
                        L.sid[at] = (u16)coarse;
                        atomicAdd(&L.bcnt[coarse], 1u);
                        ++at;
                        s2 += n;
                        len -= n;
                    }
                }
                done = true;
            }
            // records appended in this round: those of the fitting prefix of threads
            const u32 room = SKM_CAP - staged;
            const u32 emitted = total <= room ? total : 0xffffffffu;   // all fitted: the common case
            if (emitted != 0xffffffffu) {
                staged += total;
                __syncthreads();
                break;
            }
            // some threads did not fit: the prefix that did ends at the largest excl + mine <= room
            if (tid == 0) misc[2] = 0;
            __syncthreads();
            if (fits && mine) atomicMax(&misc[2], excl + mine);
            __syncthreads();
            const u32 part = misc[2];
            skm_flush(L, staged + part, jb.nb1, jb.cur1, jb.reg1, jb.cap1, jb.ctl);
            staged = 0;
        }
    }
    __syncthreads();
    if (staged) skm_flush(L, staged, jb.nb1, jb.cur1, jb.reg1, jb.cap1, jb.ctl);
    if (tid == 0 && misc[1]) atomicAdd(&jb.inst[t.seg], (unsigned long long)misc[1]);
}

// ------------------------------------------------------------------------------------------
// S2a: the records of one coarse bucket, 2048 at a time, regrouped by fine slot
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SKM_NT, 3) void k_skm_regroup(const KhSkmJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const u32 nbk = (jb.S + 3) & ~3u;
    const FlushLds L = flush_lds(lds_raw, nbk);
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    const u32 have = jb.cur1[b];
    const u32 cnt = have < jb.cap1 ? have : jb.cap1;
    const u32 start = blockIdx.y * SKM_CAP;
    if (start >= cnt) return;   // uniform
    const u32 n = cnt - start < SKM_CAP ? cnt - start : SKM_CAP;
    for (u32 i = tid; i < nbk; i += SKM_NT) L.bcnt[i] = 0;
    __syncthreads();
    const uint4* __restrict__ src = jb.reg1 + (u64)b * jb.cap1 + start;
    const u32 first_slot = b * jb.S;
    const u32 nfine = jb.nslots - first_slot < jb.S ? jb.nslots - first_slot : jb.S;
#pragma unroll
    for (u32 r = 0; r < SKM_RPT; ++r) {
        const u32 i = tid + r * SKM_NT;
        if (i < n) {
            const uint4 rec = src[i];
            u32 fine = (rec.w >> 12) & 511u;
            if (fine >= nfine) { fine = 0; atomicOr(jb.ctl, KH_ERR_ORDER); }   // a corrupt record never leaves its bucket
            L.stage[i] = rec;
            L.sid[i] = (u16)fine;
            atomicAdd(&L.bcnt[fine], 1u);
        }
    }
    __syncthreads();
    skm_flush(L, n, nfine, jb.cur2 + first_slot, jb.reg2 + (u64)first_slot * jb.cap2, jb.cap2, jb.ctl);
}

// ------------------------------------------------------------------------------------------
// S2b: one slot per workgroup -> LDS hash set {canonical k-mer, genome mask} -> histogram bins
// ------------------------------------------------------------------------------------------
constexpr u32 SKM_UNT = 512, SKM_UT = 4096, SKM_UE = SKM_UT / SKM_UNT, SKM_UT2 = SKM_UT / 16;
constexpr u32 SKM_URPT = 4;                       // records per thread when the slot is read: cap2 <= 4 * 512
constexpr u32 SKM_OWN = 2048;                     // chunk owners: a slot of up to 8 * 2048 k-mer instances
size_t kh_skm_union_lds_bytes(u32 nbins) {
    return (size_t)SKM_UT * 16 + (size_t)SKM_UT2 * 16 + 256 + 128 + 256 + (((size_t)nbins * 32 + 15) & ~(size_t)15) +
           (size_t)(SKM_URPT * SKM_UNT + 8) * 2 + (size_t)SKM_OWN * 2;
}

__global__ __launch_bounds__(SKM_UNT, 4) void k_skm_union(const KhSkmJob jb, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr u32 NT = SKM_UNT, T = SKM_UT, T2 = SKM_UT2, HBITS = 12;
    constexpr int E = (int)SKM_UE;
    constexpr u64 EMPTY = ~0ull;   // never a canonical key: the reverse complement of the all-T k-mer is 0
    struct alignas(16) Ent { unsigned long long key, mask; };
    u8* p = lds_raw;
    Ent* tbl = reinterpret_cast<Ent*>(p);                     p += (size_t)T * 16;
    Ent* ovf = reinterpret_cast<Ent*>(p);                     p += (size_t)T2 * 16;
    u32* ginfo = reinterpret_cast<u32*>(p);                   p += 256;
    u32* scratch = reinterpret_cast<u32*>(p);                 p += 128;
    u32* dupc = reinterpret_cast<u32*>(p);                    p += 256;
    u32* hstripe = reinterpret_cast<u32*>(p);                 p += ((size_t)jb.nbins * 32 + 15) & ~(size_t)15;
    u16* roff = reinterpret_cast<u16*>(p);                    p += (size_t)(SKM_URPT * NT + 8) * 2;
    u16* owner = reinterpret_cast<u16*>(p);
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const u32 nbins = jb.nbins, cap2 = jb.cap2;
    const int k = jb.k;
    const u32 slot = blockIdx.x;
    const uint4* __restrict__ reg = jb.reg2 + (u64)slot * cap2;
    // ---- the slot's records: four per thread, requested before their number is known
    const u32 have = jb.cur2[slot];
    uint4 rr[SKM_URPT];
#pragma unroll
    for (u32 j = 0; j < SKM_URPT; ++j) {
        const u32 i = SKM_URPT * tid + j;
        rr[j] = i < cap2 ? reg[i] : make_uint4(0, 0, 0, 0);
    }
    auto clear_tables = [&]() {
        uint4* t4 = reinterpret_cast<uint4*>(tbl);
#pragma unroll
        for (int e = 0; e < E; ++e) t4[(u32)e * NT + tid] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
        uint4* o4 = reinterpret_cast<uint4*>(ovf);
        for (u32 i = tid; i < T2; i += NT) o4[i] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
    };
    for (u32 i = tid; i < (u32)KH_TAG_MAX_OPS; i += NT) { ginfo[i] = jb.ginfo[i]; dupc[i] = 0; }
    for (u32 i = tid; i < nbins * 8u; i += NT) hstripe[i] = 0;
    clear_tables();
    const u32 nrec = have < cap2 ? have : cap2;
    u32 nj[SKM_URPT], mine = 0;
#pragma unroll
    for (u32 j = 0; j < SKM_URPT; ++j) {
        nj[j] = SKM_URPT * tid + j < nrec ? rr[j].w >> 27 : 0u;
        mine += nj[j];
    }
    const u32 incl = wave_scan_add(mine);
    if (lane == KH_WAVE - 1) scratch[wid] = incl;
    __syncthreads();
    u32 off = incl - mine, N = 0;
    for (u32 q = 0; q < NT / 64; ++q) {
        const u32 v = scratch[q];
        off += q < wid ? v : 0u;
        N += v;
    }
    if (N > 8u * SKM_OWN) {   // uniform: a slot this full goes back to the host
        if (tid == 0) { atomicOr(jb.ctl, KH_ERR_CAPACITY); atomicMax(jb.ctl + 1, N); }
        N = 0;
    }
    if (N) {
#pragma unroll
        for (u32 j = 0; j < SKM_URPT; ++j) {
            if (nj[j]) {
                roff[SKM_URPT * tid + j] = (u16)off;
                for (u32 c = (off + 7) >> 3; c <= (off + nj[j] - 1) >> 3; ++c) owner[c] = (u16)(SKM_URPT * tid + j);
                off += nj[j];
            }
        }
    }
    __syncthreads();
    const u32 R = (N + T - 1) / T;   // key subsets handled one after the other (1 unless the slot is overfull)
    const u64 kmask = kh_mask(2 * k);
    auto eval_mask = [&](u64 mask, u32 g) -> u32 {
        u32 ng = 0;
        while (true) {
            const u32 g0 = g & 0xffu, gn = (g >> 8) & 0xffu, bin0 = g >> 16;
            const u64 gm = (gn >= 64u ? ~0ull : ((1ull << gn) - 1ull)) << g0;
            u32 c = (u32)__popcll(mask & gm);
            c = c < cs ? c : cs;
            atomicAdd(&hstripe[(bin0 + c) * 8u + (lane & 7u)], 1u);
            mask &= ~gm;
            ++ng;
            if (!mask) break;
            g = ginfo[__ffsll((unsigned long long)mask) - 1];
        }
        return ng < cs ? ng : cs;
    };
    for (u32 q = 0; q < R; ++q) {
        if (q) { clear_tables(); __syncthreads(); }
        for (u32 base = 0; base < N; base += NT * (u32)E) {
            const u32 j0 = base + (u32)E * tid;
            u64 kreg[E];
            u32 tagp[(E + 3) / 4], slot_[E], hh[E], act = 0;
#pragma unroll
            for (int w2 = 0; w2 < (E + 3) / 4; ++w2) tagp[w2] = 0;
            if (j0 < N) {
                u32 ri = owner[j0 >> 3];
                u32 o = j0 - roff[ri];
                const uint4 r0 = reg[ri];
                const uint4 r1 = ri + 1 < nrec ? reg[ri + 1] : make_uint4(0, 0, 0, 0);
                const uint4 r2 = ri + 2 < nrec ? reg[ri + 2] : make_uint4(0, 0, 0, 0);
                u64 clo = ((u64)r0.y << 32) | r0.x, chi = ((u64)r0.w << 32) | r0.z;
                u32 cn = r0.w >> 27, ctag = (r0.w >> 21) & 63u, nxt = 1;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    kreg[e] = EMPTY;
                    hh[e] = 0;
                    if (j0 + (u32)e < N) {
                        if (o == cn) {
                            ++ri;
                            const uint4 r = nxt == 1 ? r1 : (nxt == 2 ? r2 : reg[ri < nrec ? ri : nrec - 1]);
                            ++nxt;
                            clo = ((u64)r.y << 32) | r.x;
                            chi = ((u64)r.w << 32) | r.z;
                            cn = r.w >> 27;
                            ctag = (r.w >> 21) & 63u;
                            o = 0;
                        }
                        const u32 sh = 2 * o;
                        const u64 x = (sh ? (clo >> sh) | ((chi << 1) << (63 - sh)) : clo) & kmask;
                        const u64 f = kh_revpairs64(x) >> (64 - 2 * k);
                        const u64 rc = (~x) & kmask;
                        const u64 can = f < rc ? f : rc;
                        const u32 h = ((u32)can ^ (u32)(can >> 32)) * 0x9E3779B1u;
                        kreg[e] = can;
                        hh[e] = h;
                        tagp[e >> 2] |= ctag << (8 * (e & 3));
                        if (R == 1 || (((h >> 4) & 0xffffu) * R) >> 16 == q) act |= 1u << e;
                        ++o;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) { kreg[e] = EMPTY; hh[e] = 0; }
            }
            auto tag = [&](int e) -> u32 { return (tagp[e >> 2] >> (8 * (e & 3))) & 63u; };
#pragma unroll
            for (int e = 0; e < E; ++e) slot_[e] = hh[e] >> (32 - HBITS);
#define SKM_PROBE_ROUNDS(TBL, TMASK, ROUNDS)                                                                          \
    for (u32 round = 0; round < (ROUNDS) && __builtin_amdgcn_ballot_w64(act != 0); ++round) {                        \
        unsigned long long old[E];                                                                                    \
        _Pragma("unroll") for (int e = 0; e < E; ++e)                                                                 \
            old[e] = (act & (1u << e)) ? atomicCAS(&(TBL)[slot_[e]].key, EMPTY, (unsigned long long)kreg[e]) : 0ull;  \
        _Pragma("unroll") for (int e = 0; e < E; ++e) {                                                               \
            if (act & (1u << e)) {                                                                                    \
                if (old[e] == EMPTY || old[e] == kreg[e]) {                                                           \
                    const unsigned long long bit = 1ull << tag(e);                                                    \
                    const unsigned long long was = atomicOr(&(TBL)[slot_[e]].mask, bit);                              \
                    if (was & bit) atomicAdd(&dupc[tag(e)], 1u);                                                      \
                    act &= ~(1u << e);                                                                                \
                } else {                                                                                              \
                    slot_[e] = (slot_[e] + 1u) & (TMASK);                                                             \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    }
            SKM_PROBE_ROUNDS(tbl, T - 1u, (u32)KH_HASH_ROUNDS)
            if (__builtin_amdgcn_ballot_w64(act != 0)) {
#pragma unroll
                for (int e = 0; e < E; ++e) slot_[e] = ((hh[e] ^ (hh[e] >> 15)) * 0x85EBCA77u) >> 24;   // T2 = 256
                SKM_PROBE_ROUNDS(ovf, T2 - 1u, T2)
                if (__builtin_amdgcn_ballot_w64(act != 0)) {   // second table full of other keys: on in the main table
#pragma unroll
                    for (int e = 0; e < E; ++e) slot_[e] = ((hh[e] >> (32 - HBITS)) + (u32)KH_HASH_ROUNDS) & (T - 1u);
                    SKM_PROBE_ROUNDS(tbl, T - 1u, T)
                    if (__builtin_amdgcn_ballot_w64(act != 0) && lane == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
                }
            }
#undef SKM_PROBE_ROUNDS
        }
        __syncthreads();
        // ---- every occupied entry is one distinct key of the slot: genome mask -> histogram bins
        u64 mk[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const Ent en = tbl[(u32)e * NT + tid];
            mk[e] = en.key != EMPTY ? en.mask : 0ull;
        }
        u32 gi[E];
#pragma unroll
        for (int e = 0; e < E; ++e) gi[e] = ginfo[mk[e] ? __ffsll((unsigned long long)mk[e]) - 1 : 0];
        u32 ones = 0;   // keys that sit in exactly one group
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (!mk[e]) continue;
            const u32 ng = eval_mask(mk[e], gi[e]);
            if (ng == 1u) ++ones;
            else atomicAdd(&hstripe[(jb.abase + ng) * 8u + (lane & 7u)], 1u);
        }
        for (u32 i = tid; i < T2; i += NT) {   // keys that moved to the second table (a few per slot)
            const Ent en = ovf[i];
            if (en.key != EMPTY) {
                const u32 ng = eval_mask(en.mask, ginfo[__ffsll((unsigned long long)en.mask) - 1]);
                if (ng == 1u) ++ones;
                else atomicAdd(&hstripe[(jb.abase + ng) * 8u + (lane & 7u)], 1u);
            }
        }
        ones = wave_scan_add(ones);
        if (lane == KH_WAVE - 1 && ones) atomicAdd(&hstripe[(jb.abase + 1u) * 8u], ones);
        __syncthreads();
    }
    unsigned long long* __restrict__ rep = jb.hist + (u64)(blockIdx.x % jb.reps) * nbins;
    for (u32 i = tid; i < nbins; i += NT) {
        u32 v = 0;
#pragma unroll
        for (u32 j = 0; j < 8; ++j) v += hstripe[i * 8u + j];
        if (v) atomicAdd(&rep[i], (unsigned long long)v);
    }
    if (tid < (u32)KH_TAG_MAX_OPS && dupc[tid]) atomicAdd(&jb.dup[tid], (unsigned long long)dupc[tid]);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <class K> static void skm_allow_lds(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
void kh_launch_skm_scatter(const KhSkmJob& job, u32 ntiles, hipStream_t st) {
    if (!ntiles) return;
    const size_t lds = kh_skm_scatter_lds_bytes(job.nb1);
    skm_allow_lds(k_skm_scatter, lds);
    hipLaunchKernelGGL(k_skm_scatter, dim3(ntiles), dim3(SKM_NT), lds, st, job);
}
void kh_launch_skm_regroup(const KhSkmJob& job, hipStream_t st) {
    const size_t lds = kh_skm_regroup_lds_bytes(job.S);
    skm_allow_lds(k_skm_regroup, lds);
    const u32 chunks = (job.cap1 + SKM_CAP - 1) / SKM_CAP;
    hipLaunchKernelGGL(k_skm_regroup, dim3(job.nb1, chunks), dim3(SKM_NT), lds, st, job);
}
void kh_launch_skm_union(const KhSkmJob& job, u32 cs, hipStream_t st) {
    const size_t lds = kh_skm_union_lds_bytes(job.nbins);
    skm_allow_lds(k_skm_union, lds);
    hipLaunchKernelGGL(k_skm_union, dim3(job.nslots), dim3(SKM_UNT), lds, st, job, cs);
}
