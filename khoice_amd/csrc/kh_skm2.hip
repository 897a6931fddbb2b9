// khoice_amd — the super-k-mer form of the fused experiment-type-1 path for TWO-WORD keys (33 <= k <= 63).
//
// Same three steps as kh_skm.hip (which see: minimizer records, two LDS counting-sort levels, one LDS hash
// set per slot), with what 2k > 64 bits changes:
//   * a record is 32 bytes: bits [0, 2(n+k-1)) the bases, in the last word n (6 bits, 26..31), the genome
//     number (20..25), the fine slot index (10 bits, 10..19); n <= min(63, 118 - k);
//   * a k-mer has up to 49 m-mers: the sliding minimum needs the hashes of the NEXT TWO threads (two DPP
//     wave_shl steps) and 96 validity flags per thread;
//   * the hash set cannot claim a 128-bit key with one compare-and-swap.  An entry is claimed on the LOW word
//     (compare-and-swap against the empty marker); the winner then stores the high word over the "not yet"
//     marker the entry was cleared with (a high word has at most 62 bits for k <= 63; one aligned 8-byte LDS
//     store).  A key that finds its own low word in an entry reads the high word until it is no longer the
//     marker and compares: equal — its entry; different — another key
//     (same low word: as good as never), on to the next entry.  In every round the owners publish before the
//     readers of the same wave look, so a wave never waits for itself; other waves run on without barriers.
//     The all-ones low word cannot be canonical for k <= 63 (a k-mer that ends in 32 T has a reverse
//     complement below 2^62), so it stays the empty marker; k = 64 takes the key-array form.
#include <hip/hip_runtime.h>

#include "kh_device.h"
#include "kh_launch.h"

#define SKM_STAMP(idx) do {} while (0)   // (the phase stamps live in kh_skm.hip's kernels only)
#include "kh_skm_device.h"

namespace {

#ifndef KH_TUNE_SKM2_STAGE
#define KH_TUNE_SKM2_STAGE 1536   // (768 / 1024 / 1536: scatter 0.99 / 0.90 / 0.85 ms at k = 41)
#endif
constexpr u32 SKM2_CAP = KH_TUNE_SKM2_STAGE;    // 32-byte records staged per flush of the scatter
constexpr u32 SKM2_RG_CAP = 4096;               // records per round of the regroup
constexpr u32 SKM2_CWN = 10;                    // code words a thread keeps: bases p .. p + 159

__device__ __forceinline__ u32 rec2_n(u32 w7) { return w7 >> 26; }
__device__ __forceinline__ u32 rec2_tag(u32 w7) { return (w7 >> 20) & 63u; }
__device__ __forceinline__ u32 rec2_fine(u32 w7) { return (w7 >> 10) & 1023u; }

}   // namespace

size_t kh_skm2_scatter_lds_bytes(u32 nb1) {
    const u32 nbk = (nb1 + 3) & ~3u;
    return flush_lds_bytes<SKM2_CAP, 8>(nbk) + (size_t)SKM_CW * 4 + (((size_t)SKM_CW * 2 + 15) & ~(size_t)15) + 64 * 4 + 4 * 64 * 4 + 64;
}
size_t kh_skm2_regroup_lds_bytes(u32 S) { return flush_lds_bytes<SKM2_RG_CAP, 8>((S + 3) & ~3u) + (size_t)((S + 3) & ~3u) * 4 + 64; }

// ------------------------------------------------------------------------------------------
// S1: bases -> 32-byte records, partitioned by coarse bucket.  WW = m-mers per k-mer, compile time.
// ------------------------------------------------------------------------------------------
template <int WW>
__global__ __launch_bounds__(SKM_NT, 2) void k_skm2_scatter(const KhSkmJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const u32 nbk = (jb.nb1 + 3) & ~3u;
    const FlushLds L = flush_lds<SKM2_CAP, 8>(lds_raw, nbk);
    u8* p = lds_raw + flush_lds_bytes<SKM2_CAP, 8>(nbk);
    u32* code = reinterpret_cast<u32*>(p);                    p += (size_t)SKM_CW * 4;
    u16* bad16 = reinterpret_cast<u16*>(p);                   p += ((size_t)SKM_CW * 2 + 15) & ~(size_t)15;
    u32* tailh = reinterpret_cast<u32*>(p);                   p += 64 * 4;       // hashes of positions SUB .. SUB + 63
    u32* xch = reinterpret_cast<u32*>(p);                     p += 4 * 64 * 4;   // [wave][j]: hashes of the wave's first 64 positions
    u32* misc = reinterpret_cast<u32*>(p);                    // [1] valid k-mers of the tile, [2] scratch, [4..] scan scratch

    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const KhTile t = jb.tiles[blockIdx.x];
    const KhSeg sg = jb.segs[t.seg];
    const u32 rtag = jb.seg_tag ? jb.seg_tag[t.seg] : t.seg;   // the tag of the tile's records: the genome, or its group
    const int k = jb.k, m = jb.m;
    const u32 nslots = jb.nslots, S = jb.S, nmax = jb.nmax;
    const u64 smagic = ((1ull << 40) + S - 1) / S;   // slot / S == (slot * smagic) >> 40 for slot < 2^20
    const u32 mmask = m >= 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    const u64 tile_pos0 = (u64)t.tile_in_seg * jb.tile_pos;
    const int subtiles = (int)(jb.tile_pos / SKM_SUB);

    for (u32 i = tid; i < nbk; i += SKM_NT) L.bcnt[i] = 0;
    if (tid < 4) misc[tid] = 0;
    u32 staged = 0, tile_recs = 0;   // uniform

    SkmFetch pre;
    skm_fetch(sg.seq, sg.len, tile_pos0, pre);
    for (int sub = 0; sub < subtiles; ++sub) {
        const u64 p0 = tile_pos0 + (u64)sub * SKM_SUB;
        if (p0 >= sg.npos) break;   // uniform
        __syncthreads();
        skm_store(pre, code, bad16);
        __syncthreads();
        if (sub + 1 < subtiles && p0 + SKM_SUB < sg.npos) skm_fetch(sg.seq, sg.len, p0 + SKM_SUB, pre);
        // ---- hashes of the m-mers starting at the thread's 32 positions
        u32 cw[SKM2_CWN];   // (the last threads' words past the sub-tile's halo are never part of a record: clamped)
#pragma unroll
        for (int i = 0; i < (int)SKM2_CWN; ++i) cw[i] = code[2 * tid + i < SKM_CW ? 2 * tid + i : SKM_CW - 1];
        u32 cur[SKM_PPT + WW - 1];
        {
            const u32 pm = (u32)(m - 1);
            const u32 pre_w = cw[0] & ((1u << (2 * pm)) - 1u);
            u32 f = revpairs32(pre_w) >> (32 - 2 * pm);
            u32 r = ((~pre_w) & ((1u << (2 * pm)) - 1u)) << 2;
            u32 nw[2];
            nw[0] = __builtin_amdgcn_alignbit(cw[1], cw[0], 2 * pm);
            nw[1] = __builtin_amdgcn_alignbit(cw[2], cw[1], 2 * pm);
#pragma unroll
            for (int j = 0; j < (int)SKM_PPT; ++j) {
                const u32 c = (nw[j >> 4] >> (2 * (j & 15))) & 3u;
                f = ((f << 2) | c) & mmask;
                r = (r >> 2) | ((3u - c) << (2 * pm));
                cur[j] = mmer_hash(f < r ? f : r);
            }
        }
        {
            if (tid < 64) {   // positions SUB .. SUB + 63, straight from the packed window
                const u32 q = SKM_SUB + tid, wq = q >> 4, oq = q & 15u;
                const u32 x = __builtin_amdgcn_alignbit(code[wq + 1], code[wq], 2 * oq) & mmask;
                const u32 f = revpairs32(x) >> (32 - 2 * m);
                const u32 r = (~x) & mmask;
                tailh[tid] = mmer_hash(f < r ? f : r);
            }
            if (lane < 2) {
#pragma unroll
                for (int j = 0; j < (int)SKM_PPT; ++j) xch[wid * 64 + lane * SKM_PPT + (u32)j] = cur[j];
            }
            __syncthreads();
            // the WW - 1 hashes behind the thread's own: the next thread's 32, then the one after's
            const u32* nx = wid + 1 < SKM_NT / 64 ? xch + (wid + 1) * 64 : tailh;
            constexpr int H1 = WW - 1 < (int)SKM_PPT ? WW - 1 : (int)SKM_PPT;
#pragma unroll
            for (int j = 0; j < H1; ++j) cur[SKM_PPT + j] = next_lane(cur[j], lane == KH_WAVE - 1 ? nx[j] : 0u);
#pragma unroll
            for (int j = 0; j < WW - 1 - H1; ++j)
                cur[2 * SKM_PPT + j] = next_lane(cur[SKM_PPT + j], lane == KH_WAVE - 1 ? nx[SKM_PPT + j] : 0u);
            window_min<WW>(cur);
        }
        // ---- which of the 32 start positions have k valid bases (96 flags)
        u32 vm;
        {
            u64 lo = (u64)bad16[2 * tid] | ((u64)bad16[2 * tid + 1] << 16) | ((u64)bad16[2 * tid + 2] << 32) |
                     ((u64)bad16[2 * tid + 3] << 48);
            const u32 i4 = 2 * tid + 4 < SKM_CW ? 2 * tid + 4 : SKM_CW - 1, i5 = 2 * tid + 5 < SKM_CW ? 2 * tid + 5 : SKM_CW - 1;
            u64 hi = (u64)bad16[i4] | ((u64)bad16[i5] << 16);
            auto shr_or = [&](u32 c) {   // (hi:lo) |= (hi:lo) >> c, 0 < c < 64
                const u64 nlo = (lo >> c) | ((hi << 1) << (63 - c)), nhi = hi >> c;
                lo |= nlo;
                hi |= nhi;
            };
            u32 cover = 1;
            while (2 * cover <= (u32)k) { shr_or(cover); cover <<= 1; }
            if ((u32)k > cover) shr_or((u32)k - cover);
            vm = ~(u32)lo;
        }
        // ---- runs of valid positions with one minimizer become records (see kh_skm.hip)
        u32 sl[SKM_PPT];
        u32 cont = 0;
#pragma unroll
        for (int j = 0; j < (int)SKM_PPT; ++j) {
            sl[j] = cur[j];
            if (j && cur[j] == cur[j - 1]) cont |= 1u << j;
        }
        cont &= vm & (vm << 1);
        const u32 tr = (vm >> 31) ? 1u + (u32)__builtin_clz(~cont | 1u) : 0u;
        const u32 lead = 1u + (u32)__builtin_ctz(~(cont >> 1));
        const u32 p_min = (u32)__builtin_amdgcn_update_dpp(0, (int)cur[SKM_PPT - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
        const u32 p_tr = (u32)__builtin_amdgcn_update_dpp(0, (int)tr, 0x138, 0xf, 0xf, false);
        // (a run goes over ONE thread boundary at most: the previous thread's last run must start inside it, and
        // mine must end inside me)
        const bool merge_in = lane != 0 && p_tr != 0 && p_tr < SKM_PPT && lead < SKM_PPT && (vm & 1u) && p_min == cur[0] &&
                              p_tr + lead <= nmax;
        const u32 ext = next_lane(merge_in ? lead : 0u, 0u);
        const u32 starts = (vm & ~cont) & ~(merge_in ? 1u : 0u);
        auto run_len = [&](u32 s) -> u32 {
            const u32 len = 1u + (u32)__builtin_ctzll(~((u64)cont >> (s + 1)));
            return s + len == SKM_PPT ? len + ext : len;
        };
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
        bool done = false, first_round = true;
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
            if (first_round) { tile_recs += total; first_round = false; }
            const bool fits = staged + excl + mine <= SKM2_CAP;
            if (!done && fits) {
                u32 at = staged + excl;
                u32 st = starts;
                while (st) {
                    const u32 s = (u32)__builtin_ctz(st);
                    st &= st - 1;
                    u32 len = run_len(s);
                    const u32 slot = slot_of(pick32(sl, s), nslots);
                    const u32 coarse = (u32)(((u64)slot * smagic) >> 40), fine = slot - coarse * S;
                    for (u32 s2 = s; len; ) {
                        const u32 n = len < nmax ? len : nmax;
                        const u32 sh = 2 * s2, r5 = sh & 31u;
                        const bool up = sh >= 32;
                        u32 src[9], out[8];
#pragma unroll
                        for (int i = 0; i < 9; ++i) src[i] = up ? cw[i + 1] : cw[i];
                        const u32 bits = 2 * (n + (u32)k - 1);   // <= 234: the bases end below the header (bit 10 of the last word)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const u32 wv = __builtin_amdgcn_alignbit(src[i + 1], src[i], r5);
                            out[i] = bits >= 32u * (i + 1) ? wv : (bits > 32u * i ? wv & ((1u << (bits - 32u * i)) - 1u) : 0u);
                        }
                        out[7] |= (fine << 10) | (rtag << 20) | (n << 26);
                        L.stage[2 * at] = make_uint4(out[0], out[1], out[2], out[3]);
                        L.stage[2 * at + 1] = make_uint4(out[4], out[5], out[6], out[7]);
                        L.sid[at] = (u16)coarse;
                        atomicAdd(&L.bcnt[coarse], 1u);
                        ++at;
                        s2 += n;
                        len -= n;
                    }
                }
                done = true;
            }
            const u32 room = SKM2_CAP - staged;
            if (total <= room) {   // all fitted: the common case
                staged += total;
                __syncthreads();
                break;
            }
            if (tid == 0) misc[2] = 0;
            __syncthreads();
            if (fits && mine) atomicMax(&misc[2], excl + mine);
            __syncthreads();
            const u32 part = misc[2];
            skm_flush<SKM_NT, SKM2_CAP, false, 8>(L, staged + part, jb.nb1, jb.cur1, jb.reg1, jb.cap1, jb.ctl);
            staged = 0;
        }
    }
    __syncthreads();
    if (staged) skm_flush<SKM_NT, SKM2_CAP, false, 8>(L, staged, jb.nb1, jb.cur1, jb.reg1, jb.cap1, jb.ctl);
    if (tid == 0 && misc[1]) atomicAdd(&jb.inst[t.seg], (unsigned long long)misc[1]);
    if (tid == 0 && tile_recs) atomicAdd(jb.ctl + 2, tile_recs);
}

// ------------------------------------------------------------------------------------------
// S2a: one workgroup per coarse bucket, 4096 records per round, regrouped by fine slot (cursors in LDS)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SKM_RG_NT, 4) void k_skm2_regroup(const KhSkmJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr int RPT = (int)(SKM2_RG_CAP / SKM_RG_NT);
    const u32 nbk = (jb.S + 3) & ~3u;
    const FlushLds L = flush_lds<SKM2_RG_CAP, 8>(lds_raw, nbk);
    u32* lcur = reinterpret_cast<u32*>(lds_raw + flush_lds_bytes<SKM2_RG_CAP, 8>(nbk));   // [nbk] records written per slot
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    const u32 have = jb.cur1[(size_t)b * KH_SKM_CUR1_STRIDE];
    const u32 cnt = have < jb.cap1 ? have : jb.cap1;
    const u32 first_slot = b * jb.S;
    const u32 nfine = jb.nslots - first_slot < jb.S ? jb.nslots - first_slot : jb.S;
    for (u32 i = tid; i < nbk; i += SKM_RG_NT) { L.bcnt[i] = 0; lcur[i] = 0; }
    const uint4* __restrict__ src = jb.reg1 + (u64)b * jb.cap1 * 2;
    uint4* __restrict__ dst = jb.reg2 + (u64)first_slot * jb.cap2 * 2;
    uint4 nx[RPT][2];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const u32 i = tid + (u32)r * SKM_RG_NT;
        nx[r][0] = i < cnt ? src[2 * i] : make_uint4(0, 0, 0, 0);
        nx[r][1] = i < cnt ? src[2 * i + 1] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    for (u32 start = 0; start < cnt; start += SKM2_RG_CAP) {
        const u32 n = cnt - start < SKM2_RG_CAP ? cnt - start : SKM2_RG_CAP;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const u32 i = tid + (u32)r * SKM_RG_NT;
            if (i < n) {
                u32 fine = rec2_fine(nx[r][1].w);
                if (fine >= nfine) { fine = 0; atomicOr(jb.ctl, KH_ERR_ORDER); }   // a corrupt record never leaves its bucket
                L.stage[2 * i] = nx[r][0];
                L.stage[2 * i + 1] = nx[r][1];
                L.sid[i] = (u16)fine;
                atomicAdd(&L.bcnt[fine], 1u);
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {   // the next round's records: in flight during the flush
            const u32 i = start + SKM2_RG_CAP + tid + (u32)r * SKM_RG_NT;
            nx[r][0] = i < cnt ? src[2 * i] : make_uint4(0, 0, 0, 0);
            nx[r][1] = i < cnt ? src[2 * i + 1] : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        SkmSpill sp;
        sp.rec = jb.spill_rec; sp.slot = jb.spill_slot; sp.n = jb.ctl + 5; sp.cap = jb.spill_cap; sp.first_slot = first_slot;
        skm_flush<SKM_RG_NT, SKM2_RG_CAP, true, 8>(L, n, nfine, lcur, dst, jb.cap2, jb.ctl, sp);
    }
    for (u32 i = tid; i < nfine; i += SKM_RG_NT) jb.cur2[first_slot + i] = lcur[i];
}

// ------------------------------------------------------------------------------------------
// S2b: LDS hash set {128-bit canonical k-mer, genome mask} per slot -> histogram bins.  The same plan as the one-word
// union of kh_skm.hip (which see): PERSISTENT workgroups with the next slot's record on its way while this one is
// worked on; identical records (the same piece of sequence in several genomes of a group) merged before anything is
// expanded; chunks of two consecutive k-mers per thread, the second by rolling both strands; every thread turns the
// entries it created into histogram bins once the masks are final.
// ------------------------------------------------------------------------------------------
// Geometry: 512 threads and a table of 1792 entries (24 bytes each) = 52 KB of LDS: THREE workgroups per CU.  The kernel
// is bound by the chain of LDS round trips per slot, not by what a slot's threads do: 1024 threads x 2048 entries (two
// per CU) and 512 x 2048 (two per CU) both took 3.07-3.11 ms at k = 41 — a slot of ~95 records and ~350 chunks leaves
// most of 1024 threads without work — so what counts is how many slots a CU has in flight.
#ifndef KH_TUNE_SKM2_UNT
#define KH_TUNE_SKM2_UNT 512
#endif
#ifndef KH_TUNE_SKM2_UT
#define KH_TUNE_SKM2_UT 1792
#endif
constexpr u32 SKM2_UNT = KH_TUNE_SKM2_UNT, SKM2_UT = KH_TUNE_SKM2_UT, SKM2_UT2 = 32;   // (the table need not be a power of two)
constexpr u32 SKM2_UE = 2;                         // k-mers per chunk
constexpr u32 SKM2_OB = 5;                         // bits of a chunk's number inside its record (n <= 63)
constexpr u32 SKM2_STAGE = SKM2_UT * 8 / 32;       // 32-byte records the low key plane holds
constexpr u32 SKM2_MAXREC = SKM2_STAGE < SKM2_UNT ? SKM2_STAGE : SKM2_UNT;   // records of a slot: one per thread, staged in the low key plane
constexpr u32 SKM2_DD = SKM2_UT * 2;               // entries of the set of record contents (the high key plane)
constexpr u32 SKM2_PASSES = 4;
constexpr u32 SKM2_MAXCH = SKM2_PASSES * SKM2_UNT; // chunks of a slot after the merge
constexpr u32 SKM2_HSTRIPE_WORDS = 288;
size_t kh_skm2_union_lds_bytes(u32) {
    return (size_t)SKM2_UT * 24 + (size_t)SKM2_UT2 * 24 + 1024 + 128 + 256 + (size_t)SKM2_HSTRIPE_WORDS * 4 + (size_t)SKM2_MAXCH * 2 +
           (size_t)SKM2_MAXREC * 4;
}
u32 kh_skm2_max_cap2() { return SKM2_MAXREC; }
u32 kh_skm2_table() { return SKM2_UT; }
u32 kh_skm2_union_per_cu() { return kh_skm2_union_lds_bytes(0) * 3 <= 160 * 1024 && SKM2_UNT * 3 <= 2048 ? 3u : 2u; }

__device__ __forceinline__ u32 key2_hash(u64 lo, u64 hi) { return ((u32)lo ^ (u32)(lo >> 32) ^ (u32)hi ^ (u32)(hi >> 32)) * 0x9E3779B1u; }

__global__ __launch_bounds__(SKM2_UNT, SKM2_UNT == 1024 ? 8 : 6) void k_skm2_union(const KhSkmJob jb, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr u32 NT = SKM2_UNT, T = SKM2_UT, T2 = SKM2_UT2, NW = NT / 64;
    static_assert((T2 & (T2 - 1)) == 0 && T % 4 == 0, "second table: a power of two");
    constexpr int E = (int)SKM2_UE;
    constexpr u64 EMPTY = ~0ull;   // never the low word of a canonical key for k <= 63 (see the head of the file)
    u8* p = lds_raw;
    unsigned long long* tklo = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T * 8;
    unsigned long long* tkhi = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T * 8;
    u32* tmlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    u32* tmhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    unsigned long long* oklo = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T2 * 8;
    unsigned long long* okhi = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T2 * 8;
    u32* omlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    u32* omhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    uint4* gtab = reinterpret_cast<uint4*>(p);                             p += 1024;
    u32* scratch = reinterpret_cast<u32*>(p);                              p += 128;
    u32* dupc = reinterpret_cast<u32*>(p);                                 p += 256;
    u32* hstripe = reinterpret_cast<u32*>(p);                              p += (size_t)SKM2_HSTRIPE_WORDS * 4;
    u16* owner = reinterpret_cast<u16*>(p);                                p += (size_t)SKM2_MAXCH * 2;
    u32* rmask = reinterpret_cast<u32*>(p);
    // the records are staged in the low key plane (32 bytes each), the set of their contents is the high key plane
    uint4* stage = reinterpret_cast<uint4*>(tklo);
    u32* dd = reinterpret_cast<u32*>(tkhi);   // [T * 2] record number + 1
    constexpr u32 DD = SKM2_DD;
    const u32 tid0 = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    u32 tid = tid0, lane = lane_id();
    const u32 nbins = jb.nbins, cap2 = jb.cap2, nslots = jb.nslots, stride = gridDim.x;
    const int k = jb.k;
    const u32 sshift = nbins <= 72u ? 2u : (nbins <= 144u ? 1u : 0u), smask = (1u << sshift) - 1u;
    // the 2k-bit mask, the shift that right-aligns a reversed 128-bit window, where the last base sits (33 <= k <= 63)
    const u32 kb = 2 * (u32)k;          // 66 .. 126
    u32 km[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) km[i] = kb >= 32u * (i + 1) ? 0xffffffffu : (kb > 32u * i ? (1u << (kb - 32u * i)) - 1u : 0u);
    const u64 kmhi = ((u64)km[3] << 32) | km[2];
    const u32 fs = 128u - kb;           // 2 .. 62
    const u32 tsh = kb - 2u - 64u;      // where the last base sits in the high word: 0 .. 60
    auto clear_keys = [&]() {   // key words all ones (low: empty, high: not yet published): T * 16 bytes
        uint4* k4 = reinterpret_cast<uint4*>(tklo);
        for (u32 i = tid; i < T; i += NT) k4[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    };
    auto clear_masks = [&]() {
        for (u32 i = tid; i < T / 2; i += NT) reinterpret_cast<uint4*>(tmlo)[i] = make_uint4(0u, 0u, 0u, 0u);   // (both mask planes: T * 8 bytes)
        unsigned long long e0 = EMPTY;
        asm volatile("" : "+v"(e0));
        if (tid < T2) { oklo[tid] = e0; okhi[tid] = e0; omlo[tid] = 0u; omhi[tid] = 0u; }
    };
    typedef const u32 __attribute__((address_space(4))) * ConstU32;
    const ConstU32 counts = (ConstU32)(unsigned long long)jb.cur2;
    const u32 fit = cap2 < SKM2_MAXREC ? cap2 : SKM2_MAXREC;
    auto count_of = [&](u32 sl) -> u32 {   // (an overfull slot is k_skm2_big's: empty here)
        const u32 n = sl < nslots ? counts[sl] : 0u;
        return n <= fit ? n : 0u;
    };
    if (tid < (u32)KH_TAG_MAX_OPS) {
        const u32 g = jb.ginfo[tid], g0 = g & 0xffu, gn = (g >> 8) & 0xffu;
        const u64 gm = gn ? (gn >= 64u ? ~0ull : ((1ull << gn) - 1ull)) << g0 : 0ull;
        gtab[tid] = make_uint4((u32)gm, (u32)(gm >> 32), (g >> 16) << sshift, 0u);
        dupc[tid] = 0;
    }
    if (tid < SKM2_HSTRIPE_WORDS) hstripe[tid] = 0;
    if (tid == 0) scratch[0] = 0;
    auto eval_mask = [&](u32 mlo, u32 mhi) -> bool {
        const u32 lsel = lane & smask;
        u32 ng = 0;
        do {
            const u32 first = mlo ? (u32)__builtin_ctz(mlo) : 32u + (u32)__builtin_ctz(mhi);
            const uint4 g = gtab[first];
            u32 c = (u32)__popc(mlo & g.x) + (u32)__popc(mhi & g.y);
            c = c < cs ? c : cs;
            atomicAdd(&hstripe[g.z + (c << sshift) + lsel], 1u);
            const u32 keep_hi = mlo ? ~0u : mhi - 1u;
            mlo &= ~g.x & (mlo - 1u);
            mhi &= ~g.y & keep_hi;
            ++ng;
        } while (mlo | mhi);
        if (ng == 1u) return true;
        atomicAdd(&hstripe[((jb.abase + (ng < cs ? ng : cs)) << sshift) + lsel], 1u);
        return false;
    };
    u32 st_full = 0, st_exp = 0;
    u32 slot = blockIdx.x;
    u32 nrec = count_of(slot);
    u32 nrec_next = count_of(slot + stride);
    uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
    if (tid < nrec) { const uint4* r = jb.reg2 + ((u64)slot * cap2 + tid) * 2; ra = r[0]; rb = r[1]; }
    for (; slot < nslots; slot += stride) {
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & (KH_WAVE - 1u);
        const uint4* __restrict__ reg = jb.reg2 + (u64)slot * cap2 * 2;
        const u32 nrec_after = count_of(slot + 2u * stride);
        if (counts[slot] > fit && tid0 == 0) {   // an overfull slot: listed for k_skm2_big
            const u32 at = atomicAdd(jb.ctl + 6, 1u);
            if (at < jb.big_cap) jb.big_list[at] = slot;
        }
        // ---- stage this slot's records, one per thread
        u32 nj = 0;
        const u32 tg = rec2_tag(rb.w);
        const u32 a0 = ra.x, a1 = ra.y, a2 = ra.z, a3 = ra.w, b0 = rb.x, b1 = rb.y, b2 = rb.z, b3 = rb.w;
        {
            u32 z = 0;
            asm volatile("" : "+v"(z));
            for (u32 i = tid; i < DD / 4; i += NT) reinterpret_cast<uint4*>(dd)[i] = make_uint4(z, z, z, z);
        }
        if (tid < nrec) {
            nj = rec2_n(b3);
            stage[2 * tid] = ra;
            stage[2 * tid + 1] = rb;
            rmask[tid] = 1u << (tg & 31u);
        }
        __syncthreads();
        // ---- identical records meet
        if (__builtin_amdgcn_ballot_w64(nj != 0)) {
            u32 h = a0 * 0x9E3779B1u ^ a1 * 0x85EBCA77u ^ a2 * 0xC2B2AE3Du ^ a3 * 0x27D4EB2Fu ^ b0 * 0x165667B1u ^ b1 * 0xD3A2646Cu ^
                    b2 * 0xFD7046C5u ^ (b3 & ~(63u << 20)) * 0xB55A4F09u;
            h ^= h >> 15;
            h *= 0x2C1B3C6Du;
            u32 hp = (u32)(((u64)h * DD) >> 32);
            const u32 nch = (nj + (u32)E - 1u) / (u32)E;
            bool pend = nj != 0, won = false;
            while (__builtin_amdgcn_ballot_w64(pend)) {
                if (pend) {
                    const u32 old = atomicCAS(&dd[hp], 0u, tid + 1u);
                    if (old == 0u) { won = true; pend = false; }
                    else {
                        const uint4 oa = stage[2 * (old - 1u)], ob = stage[2 * (old - 1u) + 1];
                        if (oa.x == a0 && oa.y == a1 && oa.z == a2 && oa.w == a3 && ob.x == b0 && ob.y == b1 && ob.z == b2 &&
                            ((ob.w ^ b3) & ~(31u << 20)) == 0u) {   // the same bases and number of k-mers, a genome of the same half
                            const u32 bit = 1u << (tg & 31u);
                            const u32 was = atomicOr(&rmask[old - 1u], bit);
                            if (was & bit) {
                                atomicAdd(&dupc[tg], nj);
                                rmask[tid] = 0u;   // "has counted repeats" (taken back if the slot is handed to k_skm2_big)
                            }
                            pend = false;
                        } else hp = hp + 1u == DD ? 0u : hp + 1u;
                    }
                }
            }
            const u32 mine = won ? (nch | (nj << 16)) : 0u;
            const u32 incl = wave_scan_add(mine);
            u32 wbase = 0;
            if (lane == KH_WAVE - 1 && incl) wbase = atomicAdd(&scratch[0], incl);
            wbase = (u32)__builtin_amdgcn_readlane((int)wbase, KH_WAVE - 1);
            if (won) {
                const u32 cstart = (wbase + incl - mine) & 0xffffu;
                if (cstart + nch <= SKM2_MAXCH) {
                    for (u32 cc = 0; cc < nch; ++cc) owner[cstart + cc] = (u16)((tid << SKM2_OB) | cc);
                }
            }
        }
        __syncthreads();
        const u32 sc0 = scratch[0];
        if ((sc0 & 0xffffu) > SKM2_MAXCH) {   // uniform, rare: the slot goes to k_skm2_big below; what the merge has counted is taken back
            if (tid < nrec && rmask[tid] == 0u) {
                const u32 w = stage[2 * tid + 1].w;
                atomicSub(&dupc[rec2_tag(w)], rec2_n(w));
            }
            __syncthreads();
        }
        // ---- the next slot's record sets out; the table is made
        ra = make_uint4(0, 0, 0, 0); rb = make_uint4(0, 0, 0, 0);
        if (tid < nrec_next) { const uint4* r = jb.reg2 + ((u64)(slot + stride) * cap2 + tid) * 2; ra = r[0]; rb = r[1]; }
        clear_keys();
        clear_masks();
        u32 C = sc0 & 0xffffu, N = sc0 >> 16;
        st_full = N > st_full ? N : st_full;
        st_exp += N;
        if (C > SKM2_MAXCH) {   // uniform: more chunks than are numbered here: k_skm2_big takes the slot
            if (tid == 0) {
                if (jb.big_list) {
                    const u32 at = atomicAdd(jb.ctl + 6, 1u);
                    if (at < jb.big_cap) jb.big_list[at] = slot;
                } else atomicOr(jb.ctl, KH_ERR_CAPACITY);
            }
            C = 0;
            N = 0;
        }
        __syncthreads();
        if (tid == 0) scratch[0] = 0;
        const u32 R = (N + T - 1) / T;
        const u32 ctid = tid;
        for (u32 q = 0; q < R; ++q) {
            if (q) { clear_keys(); clear_masks(); __syncthreads(); }
            u32 made[SKM2_PASSES];
#pragma unroll
            for (u32 ps = 0; ps < SKM2_PASSES; ++ps) made[ps] = 0u;
            u32 ones = 0;
#pragma unroll
            for (u32 pass = 0; pass < SKM2_PASSES; ++pass) {
                if (pass * NT >= C) break;
                const u32 c = pass * NT + ctid;
                u64 klo[E], khi[E];
                u32 slot_[E], act = 0, bits = 0, half = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) { klo[e] = EMPTY; khi[e] = 0; slot_[e] = 0; }
                if (c < C) {
                    const u32 o = owner[c], ri = o >> SKM2_OB, first = (o & ((1u << SKM2_OB) - 1u)) * (u32)E;
                    const uint4 a = reg[2 * ri], b = reg[2 * ri + 1];
                    bits = rmask[ri];
                    half = (b.w >> 25) & 1u;
                    const u32 left = rec2_n(b.w) - first;
                    const u32 cnt = left < (u32)E ? left : (u32)E;
                    // 128 bits of the record from base `first` on: words wq .. wq + 4, funnel-shifted
                    const u32 sh = 2 * first, wq = sh >> 5, r5 = sh & 31u;
                    const u32 R0 = a.x, R1 = a.y, R2 = a.z, R3 = a.w, R4 = b.x, R5 = b.y, R6 = b.z, R7 = b.w & 0x3ffu;
                    const bool q1 = wq & 1u, q2 = wq & 2u;
                    auto sel = [&](u32 v0, u32 v1, u32 v2, u32 v3) -> u32 {
                        const u32 lo2 = q1 ? v1 : v0, hi2 = q1 ? v3 : v2;
                        return q2 ? hi2 : lo2;
                    };
                    const u32 s0 = sel(R0, R1, R2, R3), s1 = sel(R1, R2, R3, R4), s2 = sel(R2, R3, R4, R5),
                              s3 = sel(R3, R4, R5, R6), s4 = sel(R4, R5, R6, R7);
                    u32 xw[4];
                    xw[0] = __builtin_amdgcn_alignbit(s1, s0, r5);
                    xw[1] = __builtin_amdgcn_alignbit(s2, s1, r5);
                    xw[2] = __builtin_amdgcn_alignbit(s3, s2, r5);
                    xw[3] = __builtin_amdgcn_alignbit(s4, s3, r5);
                    // the base behind the first k-mer (bits 2k .. 2k + 1 of the window): the second k-mer's last
                    const u32 nbw = kb >= 96u ? xw[3] : xw[2];
                    const u32 nb = (nbw >> (kb & 31u)) & 3u;
                    const u32 x0 = xw[0] & km[0], x1 = xw[1] & km[1], x2 = xw[2] & km[2], x3 = xw[3] & km[3];
                    // forward key: the window with the order of its bases reversed, right-aligned
                    const u32 y0 = revpairs32(x3), y1 = revpairs32(x2), y2 = revpairs32(x1), y3 = revpairs32(x0);
                    const u64 ylo = ((u64)y1 << 32) | y0, yhi = ((u64)y3 << 32) | y2;
                    u64 flo = (ylo >> fs) | ((yhi << 1) << (63 - fs)), fhi = yhi >> fs;
                    // reverse complement key: the complemented window
                    u64 rlo = ((u64)(x1 ^ km[1]) << 32) | (x0 ^ km[0]), rhi = ((u64)(x3 ^ km[3]) << 32) | (x2 ^ km[2]);
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if (e) {   // roll both strands by one base
                            fhi = ((fhi << 2) | (flo >> 62)) & kmhi;
                            flo = (flo << 2) | nb;
                            rlo = (rlo >> 2) | (rhi << 62);
                            rhi = (rhi >> 2) | ((u64)(3u - nb) << tsh);
                        }
                        const bool fwd = fhi < rhi || (fhi == rhi && flo < rlo);
                        klo[e] = fwd ? flo : rlo;
                        khi[e] = fwd ? fhi : rhi;
                        const u32 h = key2_hash(klo[e], khi[e]);
                        slot_[e] = (u32)(((u64)h * T) >> 32);
                        if ((u32)e < cnt && (R == 1 || (((h >> 4) & 0xffffu) * R) >> 16 == q)) act |= 1u << e;
                    }
                }
                if (!__builtin_amdgcn_ballot_w64(act != 0)) continue;
                u32* const mp = half ? tmhi : tmlo;
                u32 was[E], f16[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { was[e] = 0u; f16[e] = 0u; }
                // One round: every active key tries its current entry.  Owners publish their high word before the
                // readers of the same wave look at entries that hold their low word.
#define SKM2_ROUND(KLO, KHI, MP, TSIZE, SECOND)                                                                        \
    {                                                                                                                  \
        unsigned long long old[E];                                                                                     \
        _Pragma("unroll") for (int e = 0; e < E; ++e)                                                                  \
            old[e] = (act & (1u << e)) ? atomicCAS(&(KLO)[slot_[e]], EMPTY, (unsigned long long)klo[e]) : 0ull;        \
        _Pragma("unroll") for (int e = 0; e < E; ++e)                                                                  \
            if ((act & (1u << e)) && old[e] == EMPTY)                                                                  \
                __hip_atomic_store(&(KHI)[slot_[e]], (unsigned long long)khi[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        _Pragma("unroll") for (int e = 0; e < E; ++e) {                                                                \
            if (act & (1u << e)) {                                                                                     \
                bool hit = old[e] == EMPTY;                                                                            \
                const bool fresh = hit;                                                                                \
                if (!hit && old[e] == klo[e]) {                                                                        \
                    unsigned long long h2;                                                                             \
                    do h2 = __hip_atomic_load(&(KHI)[slot_[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       \
                    while (h2 == EMPTY);                                                                               \
                    hit = h2 == khi[e];                                                                                \
                }                                                                                                      \
                if (hit) {                                                                                             \
                    was[e] = atomicOr((MP) + slot_[e], bits);                                                          \
                    act &= ~(1u << e);                                                                                 \
                    if (fresh) f16[e] = 0x8000u | (SECOND) | slot_[e];                                                 \
                } else {                                                                                               \
                    slot_[e] = slot_[e] + 1u == (TSIZE) ? 0u : slot_[e] + 1u;                                          \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
                for (u32 round = 0; round < 4u && __builtin_amdgcn_ballot_w64(act != 0); ++round) SKM2_ROUND(tklo, tkhi, mp, T, 0u)
                if (__builtin_amdgcn_ballot_w64(act != 0)) {
                    u32* const mp2 = half ? omhi : omlo;
#pragma unroll
                    for (int e = 0; e < E; ++e) { const u32 h = key2_hash(klo[e], khi[e]); slot_[e] = ((h ^ (h >> 15)) * 0x85EBCA77u) >> 27; }   // T2 = 32
                    for (u32 round = 0; round < 8u && __builtin_amdgcn_ballot_w64(act != 0); ++round) SKM2_ROUND(oklo, okhi, mp2, T2, 0x4000u)
                    if (__builtin_amdgcn_ballot_w64(act != 0)) {   // a crowded second table: on in the main one
#pragma unroll
                        for (int e = 0; e < E; ++e) { const u32 x = (u32)(((u64)key2_hash(klo[e], khi[e]) * T) >> 32) + 4u; slot_[e] = x >= T ? x - T : x; }
                        for (u32 round = 0; round < T && __builtin_amdgcn_ballot_w64(act != 0); ++round) SKM2_ROUND(tklo, tkhi, mp, T, 0u)
                        if (__builtin_amdgcn_ballot_w64(act != 0) && lane == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
                    }
                }
#undef SKM2_ROUND
                made[pass] = f16[0] | (f16[1] << 16);
                {
                    const u32 d = (was[0] | was[1]) & bits;
                    if (d) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            u32 de = was[e] & bits;
                            while (de) { atomicAdd(&dupc[half * 32u + (u32)__builtin_ctz(de)], 1u); de &= de - 1u; }
                        }
                    }
                }
            }
            __syncthreads();
            // ---- all masks are final: every thread turns the entries it created into histogram bins
#pragma unroll
            for (u32 pass = 0; pass < SKM2_PASSES; ++pass) {
                if (!__builtin_amdgcn_ballot_w64(made[pass] != 0)) continue;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 f = (made[pass] >> (16 * e)) & 0xffffu;
                    if (f & 0x8000u) {
                        const u32 at = f & 0x3fffu;
                        const bool second = f & 0x4000u;
                        if (eval_mask((second ? omlo : tmlo)[at], (second ? omhi : tmhi)[at])) ++ones;
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(ones != 0)) {
                ones = wave_scan_add(ones);
                if (lane == KH_WAVE - 1) atomicAdd(&hstripe[(jb.abase + 1u) << sshift], ones);
            }
            if (q + 1 < R) __syncthreads();
        }
        nrec = nrec_next;
        nrec_next = nrec_after;
    }
    __syncthreads();
    tid = tid0;
    unsigned long long* __restrict__ rep = jb.hist + (u64)(blockIdx.x % jb.reps) * nbins;
    for (u32 i = tid; i < nbins; i += NT) {
        u32 v = 0;
        for (u32 j = 0; j <= smask; ++j) v += hstripe[(i << sshift) + j];
        if (v) atomicAdd(&rep[i], (unsigned long long)v);
    }
    if (tid < (u32)KH_TAG_MAX_OPS && dupc[tid]) atomicAdd(&jb.dup[tid], (unsigned long long)dupc[tid]);
    if (tid == 0) {
        if (st_full > T) atomicMax(jb.ctl + 1, st_full);
        atomicAdd(jb.ctl + 3, st_exp);
    }
}

// ------------------------------------------------------------------------------------------
// Overfull slots with two-word keys (as k_skm_big in kh_skm.hip): one workgroup per listed slot takes the records in
// its region and its records on the side list, every k-mer into a 2048-entry table in rounds of key subsets, read-out by
// a scan of the table.  Nothing here is tuned: a handful of slots per run.
// ------------------------------------------------------------------------------------------
constexpr u32 SKM2_BIG_NT = 1024, SKM2_BIG_T = 2048, SKM2_BIG_T2 = 64, SKM2_BIG_IDX = 4096;
constexpr u32 SKM2_BIG_BATCH = 128, SKM2_BIG_MAXCH = SKM2_BIG_BATCH << SKM2_OB;
size_t kh_skm2_big_lds_bytes() {
    return (size_t)SKM2_BIG_T * 24 + (size_t)SKM2_BIG_T2 * 24 + 1024 + 128 + 256 + (size_t)SKM2_HSTRIPE_WORDS * 4 +
           (size_t)SKM2_BIG_MAXCH * 2 + (size_t)SKM2_BIG_IDX * 4;
}
__global__ __launch_bounds__(SKM2_BIG_NT) void k_skm2_big(const KhSkmJob jb, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr u32 NT = SKM2_BIG_NT, T = SKM2_BIG_T, T2 = SKM2_BIG_T2, HBITS = 11;
    constexpr int E = (int)SKM2_UE;
    constexpr u64 EMPTY = ~0ull;
    u8* p = lds_raw;
    unsigned long long* tklo = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T * 8;
    unsigned long long* tkhi = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T * 8;
    u32* tmlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    u32* tmhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    unsigned long long* oklo = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T2 * 8;
    unsigned long long* okhi = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T2 * 8;
    u32* omlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    u32* omhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    uint4* gtab = reinterpret_cast<uint4*>(p);                             p += 1024;
    u32* scratch = reinterpret_cast<u32*>(p);                              p += 128;
    u32* dupc = reinterpret_cast<u32*>(p);                                 p += 256;
    u32* hstripe = reinterpret_cast<u32*>(p);                              p += (size_t)SKM2_HSTRIPE_WORDS * 4;
    u16* owner = reinterpret_cast<u16*>(p);                                p += (size_t)SKM2_BIG_MAXCH * 2;
    u32* sidx = reinterpret_cast<u32*>(p);
    const u32 tid = threadIdx.x, lane = lane_id();
    const u32 nbins = jb.nbins, cap2 = jb.cap2;
    const int k = jb.k;
    const u32 slot = jb.big_list[blockIdx.x];
    const u32 sshift = nbins <= 72u ? 2u : (nbins <= 144u ? 1u : 0u), smask = (1u << sshift) - 1u;
    const u32 kb = 2 * (u32)k;
    u32 km[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) km[i] = kb >= 32u * (i + 1) ? 0xffffffffu : (kb > 32u * i ? (1u << (kb - 32u * i)) - 1u : 0u);
    const u64 kmhi = ((u64)km[3] << 32) | km[2];
    const u32 fs = 128u - kb, tsh = kb - 2u - 64u;
    if (tid < (u32)KH_TAG_MAX_OPS) {
        const u32 g = jb.ginfo[tid], g0 = g & 0xffu, gn = (g >> 8) & 0xffu;
        const u64 gm = gn ? (gn >= 64u ? ~0ull : ((1ull << gn) - 1ull)) << g0 : 0ull;
        gtab[tid] = make_uint4((u32)gm, (u32)(gm >> 32), (g >> 16) << sshift, 0u);
        dupc[tid] = 0;
    }
    if (tid < SKM2_HSTRIPE_WORDS) hstripe[tid] = 0;
    if (tid < 8) scratch[tid] = 0;
    __syncthreads();
    u32 nspill_all = jb.ctl[5];
    nspill_all = nspill_all < jb.spill_cap ? nspill_all : jb.spill_cap;
    for (u32 i = tid; i < nspill_all; i += NT) {
        if (jb.spill_slot[i] == slot) {
            const u32 at = atomicAdd(&scratch[3], 1u);
            if (at < SKM2_BIG_IDX) sidx[at] = i;
        }
    }
    __syncthreads();
    u32 nside = scratch[3];
    if (nside > SKM2_BIG_IDX) {
        if (tid == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
        nside = SKM2_BIG_IDX;
    }
    const u32 nreg = jb.cur2[slot] < cap2 ? jb.cur2[slot] : cap2;
    const uint4* __restrict__ reg = jb.reg2 + (u64)slot * cap2 * 2;
    const u32 nall = nreg + nside;
    auto rec_a = [&](u32 i) -> uint4 { return i < nreg ? reg[2 * i] : jb.spill_rec[2 * (u64)sidx[i - nreg]]; };
    auto rec_b = [&](u32 i) -> uint4 { return i < nreg ? reg[2 * i + 1] : jb.spill_rec[2 * (u64)sidx[i - nreg] + 1]; };
    {
        u32 mine = 0;
        for (u32 i = tid; i < nall; i += NT) mine += rec2_n(rec_b(i).w);
        const u32 tot = wave_scan_add(mine);
        if (lane == KH_WAVE - 1 && tot) atomicAdd(&scratch[2], tot);
    }
    __syncthreads();
    const u32 N = scratch[2];
    const u32 R = (N + 1535u) / 1536u;
    if (tid == 0 && N > T) atomicMax(jb.ctl + 1, N);
    auto eval_mask = [&](u32 mlo, u32 mhi) -> bool {
        const u32 lsel = lane & smask;
        u32 ng = 0;
        do {
            const u32 first = mlo ? (u32)__builtin_ctz(mlo) : 32u + (u32)__builtin_ctz(mhi);
            const uint4 g = gtab[first];
            u32 c = (u32)__popc(mlo & g.x) + (u32)__popc(mhi & g.y);
            c = c < cs ? c : cs;
            atomicAdd(&hstripe[g.z + (c << sshift) + lsel], 1u);
            const u32 keep_hi = mlo ? ~0u : mhi - 1u;
            mlo &= ~g.x & (mlo - 1u);
            mhi &= ~g.y & keep_hi;
            ++ng;
        } while (mlo | mhi);
        if (ng == 1u) return true;
        atomicAdd(&hstripe[((jb.abase + (ng < cs ? ng : cs)) << sshift) + lsel], 1u);
        return false;
    };
    for (u32 q = blockIdx.y; q < R; q += gridDim.y) {   // (the rounds are independent: workgroups (slot, y) share them out)
        for (u32 i = tid; i < T; i += NT) reinterpret_cast<uint4*>(tklo)[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        for (u32 i = tid; i < T / 2; i += NT) reinterpret_cast<uint4*>(tmlo)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < T2) { oklo[tid] = EMPTY; okhi[tid] = EMPTY; omlo[tid] = 0u; omhi[tid] = 0u; }
        __syncthreads();
        for (u32 b0 = 0; b0 < nall; b0 += SKM2_BIG_BATCH) {
            const u32 mine_i = b0 + tid;
            const u32 nj = tid < SKM2_BIG_BATCH && mine_i < nall ? rec2_n(rec_b(mine_i).w) : 0u;
            const u32 nch = (nj + (u32)E - 1u) / (u32)E;
            {
                const u32 incl = wave_scan_add(nch);
                u32 wbase = 0;
                if (lane == KH_WAVE - 1 && incl) wbase = atomicAdd(&scratch[0], incl);
                wbase = (u32)__builtin_amdgcn_readlane((int)wbase, KH_WAVE - 1);
                const u32 cstart = wbase + incl - nch;
                if (cstart + nch <= SKM2_BIG_MAXCH)
                    for (u32 cc = 0; cc < nch; ++cc) owner[cstart + cc] = (u16)((tid << SKM2_OB) | cc);
            }
            __syncthreads();
            u32 C = scratch[0];
            if (C > SKM2_BIG_MAXCH) {
                if (tid == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
                C = 0;
            }
            for (u32 c = tid; c < C; c += NT) {
                const u32 o = owner[c], ri = b0 + (o >> SKM2_OB), first = (o & ((1u << SKM2_OB) - 1u)) * (u32)E;
                const uint4 a = rec_a(ri), b = rec_b(ri);
                const u32 tg = rec2_tag(b.w), bit = 1u << (tg & 31u), half = tg >> 5;
                const u32 left = rec2_n(b.w) - first;
                const u32 cnt = left < (u32)E ? left : (u32)E;
                const u32 sh = 2 * first, wq = sh >> 5, r5 = sh & 31u;
                const u32 R0 = a.x, R1 = a.y, R2 = a.z, R3 = a.w, R4 = b.x, R5 = b.y, R6 = b.z, R7 = b.w & 0x3ffu;
                const bool q1 = wq & 1u, q2 = wq & 2u;
                auto sel = [&](u32 v0, u32 v1, u32 v2, u32 v3) -> u32 {
                    const u32 lo2 = q1 ? v1 : v0, hi2 = q1 ? v3 : v2;
                    return q2 ? hi2 : lo2;
                };
                const u32 s0 = sel(R0, R1, R2, R3), s1 = sel(R1, R2, R3, R4), s2 = sel(R2, R3, R4, R5),
                          s3 = sel(R3, R4, R5, R6), s4 = sel(R4, R5, R6, R7);
                u32 xw[4];
                xw[0] = __builtin_amdgcn_alignbit(s1, s0, r5);
                xw[1] = __builtin_amdgcn_alignbit(s2, s1, r5);
                xw[2] = __builtin_amdgcn_alignbit(s3, s2, r5);
                xw[3] = __builtin_amdgcn_alignbit(s4, s3, r5);
                const u32 nbw = kb >= 96u ? xw[3] : xw[2];
                const u32 nb = (nbw >> (kb & 31u)) & 3u;
                const u32 x0 = xw[0] & km[0], x1 = xw[1] & km[1], x2 = xw[2] & km[2], x3 = xw[3] & km[3];
                const u32 y0 = revpairs32(x3), y1 = revpairs32(x2), y2 = revpairs32(x1), y3 = revpairs32(x0);
                const u64 ylo = ((u64)y1 << 32) | y0, yhi = ((u64)y3 << 32) | y2;
                u64 flo = (ylo >> fs) | ((yhi << 1) << (63 - fs)), fhi = yhi >> fs;
                u64 rlo = ((u64)(x1 ^ km[1]) << 32) | (x0 ^ km[0]), rhi = ((u64)(x3 ^ km[3]) << 32) | (x2 ^ km[2]);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (e) {
                        fhi = ((fhi << 2) | (flo >> 62)) & kmhi;
                        flo = (flo << 2) | nb;
                        rlo = (rlo >> 2) | (rhi << 62);
                        rhi = (rhi >> 2) | ((u64)(3u - nb) << tsh);
                    }
                    if ((u32)e >= cnt) break;
                    const bool fwd = fhi < rhi || (fhi == rhi && flo < rlo);
                    const unsigned long long KL = fwd ? flo : rlo, KH = fwd ? fhi : rhi;
                    const u32 H = key2_hash(KL, KH);
                    if (R != 1 && (((H >> 4) & 0xffffu) * R) >> 16 != q) continue;
                    u32 S = H >> (32 - HBITS), probes = 0, level = 0;
                    while (true) {
                        unsigned long long* kl = level == 1 ? oklo : tklo;
                        unsigned long long* kh = level == 1 ? okhi : tkhi;
                        const unsigned long long o2 = atomicCAS(&kl[S], EMPTY, KL);
                        bool hit = o2 == EMPTY;
                        // the owners of this step publish their high words BEFORE any lane of the wave waits for one
                        // (one if / else would let the waiting lanes run first and spin on a masked-off owner for ever)
                        if (hit) __hip_atomic_store(&kh[S], KH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __builtin_amdgcn_wave_barrier();
                        if (!hit && o2 == KL) {
                            unsigned long long h2;
                            do h2 = __hip_atomic_load(&kh[S], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            while (h2 == EMPTY);
                            hit = h2 == KH;
                        }
                        if (hit) {
                            u32* mp = level == 1 ? (half ? omhi : omlo) : (half ? tmhi : tmlo);
                            if (atomicOr(mp + S, bit) & bit) atomicAdd(&dupc[tg], 1u);
                            break;
                        }
                        ++probes;
                        if (level == 0 && probes >= 4u) {
                            level = 1; probes = 0;
                            S = ((H ^ (H >> 15)) * 0x85EBCA77u) >> 26;
                        } else if (level == 1 && probes >= 8u) {
                            level = 2; probes = 0;
                            S = ((H >> (32 - HBITS)) + 4u) & (T - 1u);
                        } else if (level == 2 && probes >= T) {
                            atomicOr(jb.ctl, KH_ERR_CAPACITY);
                            break;
                        } else {
                            S = (S + 1u) & (level == 1 ? T2 - 1u : T - 1u);
                        }
                    }
                }
            }
            __syncthreads();
            if (tid == 0) scratch[0] = 0;
            __syncthreads();
        }
        u32 ones = 0;
        for (u32 i = tid; i < T; i += NT)
            if (tklo[i] != EMPTY && eval_mask(tmlo[i], tmhi[i])) ++ones;
        if (tid < T2 && oklo[tid] != EMPTY && eval_mask(omlo[tid], omhi[tid])) ++ones;
        ones = wave_scan_add(ones);
        if (lane == KH_WAVE - 1 && ones) atomicAdd(&hstripe[(jb.abase + 1u) << sshift], ones);
        __syncthreads();
    }
    unsigned long long* __restrict__ rep = jb.hist + (u64)(blockIdx.x % jb.reps) * nbins;
    for (u32 i = tid; i < nbins; i += NT) {
        u32 v = 0;
        for (u32 j = 0; j <= smask; ++j) v += hstripe[(i << sshift) + j];
        if (v) atomicAdd(&rep[i], (unsigned long long)v);
    }
    if (tid < (u32)KH_TAG_MAX_OPS && dupc[tid]) atomicAdd(&jb.dup[tid], (unsigned long long)dupc[tid]);
    if (tid == 0 && blockIdx.y == 0) atomicAdd(jb.ctl + 3, N);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <class K> static void skm2_allow_lds(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <int WW> static void launch_scatter2_w(const KhSkmJob& job, u32 ntiles, size_t lds, hipStream_t st) {
    skm2_allow_lds(k_skm2_scatter<WW>, lds);
    hipLaunchKernelGGL(k_skm2_scatter<WW>, dim3(ntiles), dim3(SKM_NT), lds, st, job);
}
// m-mers per k-mer the two-word scatter is instantiated for: every third value, so that one of the minimizer
// lengths 14, 15, 16 fits any k
bool kh_skm2_supports_w(u32 w) { return w >= 18 && w <= 51 && w % 3 == 0; }
void kh_launch_skm2_scatter(const KhSkmJob& job, u32 ntiles, hipStream_t st) {
    if (!ntiles) return;
    const size_t lds = kh_skm2_scatter_lds_bytes(job.nb1);
    switch (job.w) {
#define SKM_W(WW) case WW: launch_scatter2_w<WW>(job, ntiles, lds, st); break;
        SKM_W(18) SKM_W(21) SKM_W(24) SKM_W(27) SKM_W(30) SKM_W(33) SKM_W(36) SKM_W(39) SKM_W(42) SKM_W(45) SKM_W(48) SKM_W(51)
#undef SKM_W
        default: break;   // the host asks kh_skm2_supports_w first
    }
}
void kh_launch_skm2_regroup(const KhSkmJob& job, hipStream_t st) {
    const size_t lds = kh_skm2_regroup_lds_bytes(job.S);
    skm2_allow_lds(k_skm2_regroup, lds);
    hipLaunchKernelGGL(k_skm2_regroup, dim3(job.nb1), dim3(SKM_RG_NT), lds, st, job);
}
void kh_launch_skm2_union(const KhSkmJob& job, u32 cs, u32 grid, hipStream_t st) {
    const size_t lds = kh_skm2_union_lds_bytes(job.nbins);
    skm2_allow_lds(k_skm2_union, lds);
    hipLaunchKernelGGL(k_skm2_union, dim3(grid), dim3(SKM2_UNT), lds, st, job, cs);
}
static u32 big_y() { const char* e = getenv("KHOICE_SKM_BIG_Y"); const int v = e ? atoi(e) : 4; return (u32)(v < 1 ? 1 : (v > 16 ? 16 : v)); }
void kh_launch_skm2_big(const KhSkmJob& job, u32 cs, u32 nbig, hipStream_t st) {
    if (!nbig) return;
    const size_t lds = kh_skm2_big_lds_bytes();
    skm2_allow_lds(k_skm2_big, lds);
    hipLaunchKernelGGL(k_skm2_big, dim3(nbig, nbig < 2048u ? big_y() : 1u), dim3(SKM2_BIG_NT), lds, st, job, cs);   // y: the rounds of a slot side by side
}
