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
//                   at two bits, genome tag, n; a run may go on into the next thread's positions): ~2 bytes
//                   per k-mer instead of 8.  2048 records are counting-sorted by coarse bucket in LDS and
//                   flushed as runs (one global atomic per run, cursors on separate memory channels).
//   k_skm_regroup   one workgroup per coarse bucket, 8192 records per round: the same LDS counting sort by
//                   fine slot (cursors in LDS: the workgroup owns its slots) -> every slot of the key space
//                   is one contiguous record range.
//   k_skm_union     one workgroup of 1024 threads per slot: records -> k-mers (balanced: a thread takes 4
//                   consecutive k-mer indices of the slot, whatever records they fall in) -> canonical key -> LDS hash
//                   set {key, genome mask} -> popcount per group / number of groups -> histogram bins.
//                   A repeated (key, genome) pair is seen when its mask bit is already set: distinct
//                   k-mers of a genome = its valid k-mer instances - those repeats.
//
// Nothing here is ordered and nothing needs to be: equal k-mers only have to meet, and they do because the
// slot is a function of the k-mer as a set of m-mers.  All integer work, no MFMA.
#include <hip/hip_runtime.h>

#include "kh_device.h"
#include "kh_launch.h"

// Diagnostic build only (-DKH_STAMPS): thread 0 of a workgroup stores the shader clock at phase boundaries.
#ifdef KH_STAMPS
__device__ u64* g_skm_stamps = nullptr;
void kh_debug_set_stamps_skm(u64* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_skm_stamps), &p, sizeof p); }
#define SKM_STAMP(idx)                                                                     \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (threadIdx.x == 0 && g_skm_stamps)                                              \
            g_skm_stamps[((u64)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (idx)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#else
#define SKM_STAMP(idx) do {} while (0)
#endif

#ifndef KH_TUNE_SKM_UNT
#define KH_TUNE_SKM_UNT 1024    // threads of a union workgroup: two of them per CU = 8 waves per SIMD (512: 2.27 ms against 1.89)
#endif
#ifndef KH_TUNE_SKM_FULL_ROUNDS
#define KH_TUNE_SKM_FULL_ROUNDS KH_TUNE_HASH_ROUNDS   // probe rounds made by all keys of a thread together; the rest one key per lane
#endif
#ifndef KH_TUNE_SKM_PREFETCH
#define KH_TUNE_SKM_PREFETCH 1   // union: the record behind the current one waits in registers (1.741 -> 1.725 ms; fits the 64 VGPRs since the addresses of the read-out are formed late)
#endif
#ifndef KH_TUNE_SKM_OVF_SERIAL
#define KH_TUNE_SKM_OVF_SERIAL 1
#endif
#ifndef KH_TUNE_SKM_SCATTER_PREFETCH
#define KH_TUNE_SKM_SCATTER_PREFETCH 0   // 1: the next sub-tile's bases wait in registers while this one is processed (0.656 ms against 0.624: three workgroups per CU hide the load as well, with fewer registers)
#endif
#ifndef KH_TUNE_SKM_SCATTER_WAVES
#define KH_TUNE_SKM_SCATTER_WAVES 3   // waves per SIMD the scatter is compiled for (workgroups of 4 waves per CU)
#endif

#include "kh_skm_device.h"   // (after SKM_STAMP: the shared flush carries stamps)

size_t kh_skm_scatter_lds_bytes(u32 nb1) {
    const u32 nbk = (nb1 + 3) & ~3u;
    return flush_lds_bytes<SKM_CAP>(nbk) + (size_t)SKM_CW * 4 + (((size_t)SKM_CW * 2 + 15) & ~(size_t)15) + 64 * 4 + 4 * 32 * 4 + 64;
}
size_t kh_skm_regroup_lds_bytes(u32 S) { return flush_lds_bytes<SKM_RG_CAP>((S + 3) & ~3u) + (size_t)((S + 3) & ~3u) * 4 + 64; }

// ------------------------------------------------------------------------------------------
// S1: bases -> records, partitioned by coarse bucket.  WW = m-mers per k-mer (k - m + 1), compile time:
// the sliding minimum is register arithmetic with constant indices.
// ------------------------------------------------------------------------------------------
template <int WW>
__global__ __launch_bounds__(SKM_NT, KH_TUNE_SKM_SCATTER_WAVES) void k_skm_scatter(const KhSkmJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const u32 nbk = (jb.nb1 + 3) & ~3u;
    const FlushLds L = flush_lds<SKM_CAP>(lds_raw, nbk);
    u8* p = lds_raw + flush_lds_bytes<SKM_CAP>(nbk);
    u32* code = reinterpret_cast<u32*>(p);                    p += (size_t)SKM_CW * 4;
    u16* bad16 = reinterpret_cast<u16*>(p);                   p += ((size_t)SKM_CW * 2 + 15) & ~(size_t)15;
    u32* tailh = reinterpret_cast<u32*>(p);                   p += 64 * 4;       // hashes of positions SUB .. SUB + 63
    u32* xch = reinterpret_cast<u32*>(p);                     p += 4 * 32 * 4;   // [wave][j]: hashes of the wave's first thread
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
#if KH_TUNE_SKM_SCATTER_PREFETCH
    skm_fetch(sg.seq, sg.len, tile_pos0, pre);
#endif
    for (int sub = 0; sub < subtiles; ++sub) {
        const u64 p0 = tile_pos0 + (u64)sub * SKM_SUB;
        if (p0 >= sg.npos) break;   // uniform
        __syncthreads();
        const bool stamp = sub == 1;
        if (stamp) SKM_STAMP(0);
#if !KH_TUNE_SKM_SCATTER_PREFETCH
        skm_fetch(sg.seq, sg.len, p0, pre);
#endif
        skm_store(pre, code, bad16);
        __syncthreads();
        if (stamp) SKM_STAMP(1);
#if KH_TUNE_SKM_SCATTER_PREFETCH
        if (sub + 1 < subtiles && p0 + SKM_SUB < sg.npos) skm_fetch(sg.seq, sg.len, p0 + SKM_SUB, pre);
#endif
        // ---- hashes of the m-mers starting at the thread's 32 positions (32-bit rolling words)
        u32 cw[6];   // bases p .. p + 95: a record starts in the thread's 32 positions and may run on into the next thread's
#pragma unroll
        for (int i = 0; i < 6; ++i) cw[i] = code[2 * tid + i];
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
        if (WW > 1) {
            if (tid < 64) {   // positions SUB .. SUB + 63, straight from the packed window
                const u32 q = SKM_SUB + tid, wq = q >> 4, oq = q & 15u;
                const u32 x = __builtin_amdgcn_alignbit(code[wq + 1], code[wq], 2 * oq) & mmask;
                const u32 f = revpairs32(x) >> (32 - 2 * m);
                const u32 r = (~x) & mmask;
                tailh[tid] = mmer_hash(f < r ? f : r);
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < WW - 1; ++j) xch[wid * 32 + j] = cur[j];
            }
            __syncthreads();
            // the first WW - 1 hashes of the next thread: next lane, or the next wave's first thread
            const u32* nx = wid + 1 < SKM_NT / 64 ? xch + (wid + 1) * 32 : tailh;
#pragma unroll
            for (int j = 0; j < WW - 1; ++j) {
                const u32 edge = lane == KH_WAVE - 1 ? nx[j] : 0u;
                cur[SKM_PPT + j] = next_lane(cur[j], edge);
            }
            window_min<WW>(cur);
        }
        if (stamp) SKM_STAMP(2);
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
        // (the hash is a bijection: equal minimizer hashes = one minimizer = one slot; the slot itself is
        // worked out once per record)
        u32 sl[SKM_PPT];
        u32 cont = 0;
#pragma unroll
        for (int j = 0; j < (int)SKM_PPT; ++j) {
            sl[j] = cur[j];
            if (j && cur[j] == cur[j - 1]) cont |= 1u << j;
        }
        cont &= vm & (vm << 1);
        // A run may go on into the NEXT thread of the wave (not past the wave's 2048 positions, not over more than
        // one boundary, not beyond nmax k-mers): then its record belongs to the thread it starts in, and the next
        // thread's leading positions are not starts.  Down the wave: minimizer and length of my last run; back up:
        // how many of the next thread's positions join it.  (Records cut at every 32 positions held 6.8 k-mers
        // where the minimizer runs average 8.5.)
        const u32 tr = (vm >> 31) ? 1u + (u32)__builtin_clz(~cont | 1u) : 0u;            // my last run's k-mers (32: all of mine)
        const u32 lead = 1u + (u32)__builtin_ctz(~(cont >> 1));                            // positions 0 .. lead-1 continue position 0's run
        const u32 p_min = (u32)__builtin_amdgcn_update_dpp(0, (int)cur[SKM_PPT - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
        const u32 p_tr = (u32)__builtin_amdgcn_update_dpp(0, (int)tr, 0x138, 0xf, 0xf, false);
        const bool merge_in = lane != 0 && p_tr != 0 && (vm & 1u) && p_min == cur[0] && p_tr + lead <= nmax;
        const u32 ext = next_lane(merge_in ? lead : 0u, 0u);   // k-mers of the next thread that join my last run
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
        if (stamp) SKM_STAMP(3);
        // ---- append to the staging array; a full array is flushed (a prefix of the threads fits).
        // (Measured and not kept: descriptors first, then records built by all threads evenly — the same
        // 0.77 ms, more registers.)
        if (stamp) SKM_STAMP(4);
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
            const bool fits = staged + excl + mine <= SKM_CAP;
            if (stamp) SKM_STAMP(9);
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
                        const u64 w0 = ((u64)cw[1] << 32) | cw[0], w1 = ((u64)cw[3] << 32) | cw[2], w2 = ((u64)cw[5] << 32) | cw[4];
                        const u32 sh = 2 * s2;
                        u64 rlo = sh ? (w0 >> sh) | ((w1 << 1) << (63 - sh)) : w0;
                        u64 rhi = sh ? (w1 >> sh) | ((w2 << 1) << (63 - sh)) : w1;
                        const u32 bits = 2 * (n + (u32)k - 1);
                        if (bits < 64) { rlo &= (1ull << bits) - 1ull; rhi = 0; }
                        else rhi &= kh_mask((int)bits - 64);
                        rhi |= ((u64)fine << 44) | ((u64)rtag << 53) | ((u64)n << 59);
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
            if (stamp) SKM_STAMP(5);
            const u32 room = SKM_CAP - staged;
            if (total <= room) {   // all fitted: the common case
                staged += total;
                if (stamp) SKM_STAMP(6);
                __syncthreads();
                if (stamp) { SKM_STAMP(7); SKM_STAMP(8); }
                break;
            }
            // some threads did not fit: the prefix that did ends at the largest excl + mine <= room
            if (tid == 0) misc[2] = 0;
            __syncthreads();
            if (fits && mine) atomicMax(&misc[2], excl + mine);
            __syncthreads();
            const u32 part = misc[2];
            if (stamp) SKM_STAMP(10);
            skm_flush<SKM_NT, SKM_CAP, false>(L, staged + part, jb.nb1, jb.cur1, jb.reg1, jb.cap1, jb.ctl);
            if (stamp) SKM_STAMP(11);
            staged = 0;
        }
    }
    __syncthreads();
    if (staged) skm_flush<SKM_NT, SKM_CAP, false>(L, staged, jb.nb1, jb.cur1, jb.reg1, jb.cap1, jb.ctl);
    if (tid == 0 && misc[1]) atomicAdd(&jb.inst[t.seg], (unsigned long long)misc[1]);
    if (tid == 0 && tile_recs) atomicAdd(jb.ctl + 2, tile_recs);
}

// ------------------------------------------------------------------------------------------
// S2a: one workgroup per coarse bucket walks its records 8192 at a time and regroups them by fine slot.
// The workgroup owns every slot of its bucket: the slot cursors live in LDS, no global atomic is needed.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SKM_RG_NT, 4) void k_skm_regroup(const KhSkmJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr int RPT = (int)(SKM_RG_CAP / SKM_RG_NT);
    const u32 nbk = (jb.S + 3) & ~3u;
    const FlushLds L = flush_lds<SKM_RG_CAP>(lds_raw, nbk);
    u32* lcur = reinterpret_cast<u32*>(lds_raw + flush_lds_bytes<SKM_RG_CAP>(nbk));   // [nbk] records written per slot
    const u32 tid = threadIdx.x;
    const u32 b = blockIdx.x;
    const u32 have = jb.cur1[(size_t)b * KH_SKM_CUR1_STRIDE];
    const u32 cnt = have < jb.cap1 ? have : jb.cap1;
    const u32 first_slot = b * jb.S;
    const u32 nfine = jb.nslots - first_slot < jb.S ? jb.nslots - first_slot : jb.S;
    for (u32 i = tid; i < nbk; i += SKM_RG_NT) { L.bcnt[i] = 0; lcur[i] = 0; }
    const uint4* __restrict__ src = jb.reg1 + (u64)b * jb.cap1;
    uint4* __restrict__ dst = jb.reg2 + (u64)first_slot * jb.cap2;
    uint4 nx[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const u32 i = tid + (u32)r * SKM_RG_NT;
        nx[r] = i < cnt ? src[i] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    for (u32 start = 0; start < cnt; start += SKM_RG_CAP) {
        const u32 n = cnt - start < SKM_RG_CAP ? cnt - start : SKM_RG_CAP;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const u32 i = tid + (u32)r * SKM_RG_NT;
            if (i < n) {
                u32 fine = (nx[r].w >> 12) & 511u;
                if (fine >= nfine) { fine = 0; atomicOr(jb.ctl, KH_ERR_ORDER); }   // a corrupt record never leaves its bucket
                L.stage[i] = nx[r];
                L.sid[i] = (u16)fine;
                atomicAdd(&L.bcnt[fine], 1u);
            }
        }
        // the next round's records: in flight during the flush
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const u32 i = start + SKM_RG_CAP + tid + (u32)r * SKM_RG_NT;
            nx[r] = i < cnt ? src[i] : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        skm_flush<SKM_RG_NT, SKM_RG_CAP, true>(L, n, nfine, lcur, dst, jb.cap2, jb.ctl);
    }
    for (u32 i = tid; i < nfine; i += SKM_RG_NT) jb.cur2[first_slot + i] = lcur[i];
}

// ------------------------------------------------------------------------------------------
// S2b: one slot per workgroup -> LDS hash set {canonical k-mer, genome mask} -> histogram bins.
//
// Records -> k-mers: a block scan of the records' k-mer counts numbers the slot's k-mers, and every thread
// expands SKM_UE (4) CONSECUTIVE k-mer indices whatever records they fall in (chunk -> record table in LDS, the
// records themselves re-read from L2): balanced lanes.
// Insertion: rounds of SKM_UE compare-and-swaps per thread.  A key gets KH_HASH_ROUNDS probes in the main table,
// moves to a small second table with an independent hash, and from a full second table back to unbounded
// probing of the main one; occupied entries stay occupied, so every copy of a key takes the same decisions as
// the first.  (Measured and rejected: finishing the keys that lost round 1 one per lane in a loop — fewer
// instructions, but a chain of ~20 dependent LDS round trips per wave: 2.63 ms against 2.30.)
// ------------------------------------------------------------------------------------------
constexpr u32 SKM_UNT = KH_TUNE_SKM_UNT, SKM_UT = 4096, SKM_UE = SKM_UT / SKM_UNT, SKM_UT2 = 256;
constexpr u32 SKM_UNW = SKM_UNT / 64;             // waves
constexpr u32 SKM_URPT = 2048 / SKM_UNT;           // records per thread when the slot is read: cap2 <= 2048
constexpr u32 SKM_SPEC = 576;                     // records of a slot read before their number is known (mean ~420)
constexpr u32 SKM_OWN = 2048;                     // chunk owners: a slot of up to SKM_UE * 2048 k-mer instances
size_t kh_skm_union_lds_bytes(u32 nbins) {
    return (size_t)SKM_UT * 16 + (size_t)SKM_UT2 * 16 + 256 + 128 + 256 + (((size_t)nbins * 32 + 15) & ~(size_t)15) +
           (size_t)(SKM_URPT * SKM_UNT + 8) * 2 + (size_t)SKM_OWN * 2;
}

__device__ __forceinline__ u32 key_hash(u64 can) { return ((u32)can ^ (u32)(can >> 32)) * 0x9E3779B1u; }

__global__ __launch_bounds__(SKM_UNT, 2 * (SKM_UNT / 64) / 4) void k_skm_union(const KhSkmJob jb, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr u32 NT = SKM_UNT, T = SKM_UT, T2 = SKM_UT2, HBITS = 12;
    constexpr int E = (int)SKM_UE;
    constexpr u64 EMPTY = ~0ull;   // never a canonical key: the reverse complement of the all-T k-mer is 0
    // Table planes: keys (8-byte stride), low and high halves of the genome masks (4-byte stride).
    struct Tbl { unsigned long long* key; u32* mlo; u32* mhi; };
    u8* p = lds_raw;
    Tbl tbl, ovf;
    tbl.key = reinterpret_cast<unsigned long long*>(p);       p += (size_t)T * 8;
    tbl.mlo = reinterpret_cast<u32*>(p);                      p += (size_t)T * 4;
    tbl.mhi = reinterpret_cast<u32*>(p);                      p += (size_t)T * 4;
    ovf.key = reinterpret_cast<unsigned long long*>(p);       p += (size_t)T2 * 8;
    ovf.mlo = reinterpret_cast<u32*>(p);                      p += (size_t)T2 * 4;
    ovf.mhi = reinterpret_cast<u32*>(p);                      p += (size_t)T2 * 4;
    u32* ginfo = reinterpret_cast<u32*>(p);                   p += 256;
    u32* scratch = reinterpret_cast<u32*>(p);                 p += 128;
    u32* dupc = reinterpret_cast<u32*>(p);                    p += 256;
    u32* hstripe = reinterpret_cast<u32*>(p);                 p += ((size_t)jb.nbins * 32 + 15) & ~(size_t)15;
    u16* roff = reinterpret_cast<u16*>(p);                    p += (size_t)(SKM_URPT * NT + 8) * 2;   // first k-mer index of every record
    u16* owner = reinterpret_cast<u16*>(p);                   // [SKM_OWN] record that holds k-mer index 8c
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const u32 nbins = jb.nbins, cap2 = jb.cap2;
    const int k = jb.k;
    const u32 slot = blockIdx.x;
    const uint4* __restrict__ reg = jb.reg2 + (u64)slot * cap2;
    // ---- the slot's records, SKM_URPT consecutive ones per thread.  Their number is not known yet: the first
    // SKM_SPEC (more than nearly every slot holds) are requested together with it, the others once it is
    // (reading the whole region up to its capacity moved 2.5x the records' bytes).
    const u32 have = jb.cur2[slot];
    uint4 rr[SKM_URPT];
#pragma unroll
    for (u32 j = 0; j < SKM_URPT; ++j) {
        const u32 i = SKM_URPT * tid + j;
        rr[j] = i < (cap2 < SKM_SPEC ? cap2 : SKM_SPEC) ? reg[i] : make_uint4(0, 0, 0, 0);
    }
    auto clear_tables = [&]() {
        uint4* k4 = reinterpret_cast<uint4*>(tbl.key);    // T * 8 bytes of ones, then T * 8 bytes of zeros (both mask planes)
        uint4* m4 = reinterpret_cast<uint4*>(tbl.mlo);
#pragma unroll
        for (int e = 0; e < E / 2; ++e) {
            k4[(u32)e * NT + tid] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            m4[(u32)e * NT + tid] = make_uint4(0u, 0u, 0u, 0u);
        }
        unsigned long long e0 = EMPTY;   // (opaque, as `emptyv` below)
        asm volatile("" : "+v"(e0));
        for (u32 i = tid; i < T2; i += NT) { ovf.key[i] = e0; ovf.mlo[i] = 0u; ovf.mhi[i] = 0u; }
    };
    for (u32 i = tid; i < (u32)KH_TAG_MAX_OPS; i += NT) { ginfo[i] = jb.ginfo[i]; dupc[i] = 0; }
    for (u32 i = tid; i < nbins * 8u; i += NT) hstripe[i] = 0;
    clear_tables();
    const u32 nrec = have < cap2 ? have : cap2;
    if (nrec > SKM_SPEC) {   // uniform, rare
#pragma unroll
        for (u32 j = 0; j < SKM_URPT; ++j) {
            const u32 i = SKM_URPT * tid + j;
            if (i >= SKM_SPEC && i < nrec) rr[j] = reg[i];
        }
    }
    // ---- number the slot's k-mers: record i holds indices [roff[i], roff[i] + n_i)
    u32 nj[SKM_URPT], mine = 0;
#pragma unroll
    for (u32 j = 0; j < SKM_URPT; ++j) {
        nj[j] = SKM_URPT * tid + j < nrec ? rr[j].w >> 27 : 0u;
        mine += nj[j];
    }
    const u32 incl = wave_scan_add(mine);
    if (lane == KH_WAVE - 1) scratch[wid] = incl;
    SKM_STAMP(0);
    __syncthreads();
    SKM_STAMP(1);
    u32 off = incl - mine, N = 0;
    for (u32 q = 0; q < SKM_UNW; ++q) {
        const u32 v = scratch[q];
        off += q < wid ? v : 0u;
        N += v;
    }
    if (tid == 0 && N > T) atomicMax(jb.ctl + 1, N);
    if (N > (u32)SKM_UE * SKM_OWN) {   // uniform: a slot this full goes back to the host
        if (tid == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
        N = 0;
    }
    if (N) {
#pragma unroll
        for (u32 j = 0; j < SKM_URPT; ++j) {
            if (nj[j]) {
                roff[SKM_URPT * tid + j] = (u16)off;
                for (u32 c = (off + (u32)E - 1) / (u32)E; c <= (off + nj[j] - 1) / (u32)E; ++c) owner[c] = (u16)(SKM_URPT * tid + j);
                off += nj[j];
            }
        }
    }
    __syncthreads();
    SKM_STAMP(2);
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
        unsigned long long emptyv = EMPTY;   // (opaque: made here, or the compiler keeps the constant in two VGPRs across the loop and spills it)
        asm volatile("" : "+v"(emptyv));
        for (u32 base = 0; base < N; base += NT * (u32)E) {
            {
                const u32 j0 = base + (u32)E * tid;
                u64 kreg[E];
                u32 tagp[(E + 3) / 4], slot_[E], act = 0;
#pragma unroll
                for (int w2 = 0; w2 < (E + 3) / 4; ++w2) tagp[w2] = 0;
                if (j0 < N) {
                    u32 ri = owner[j0 / (u32)E];
                    u32 o = j0 - roff[ri];
                    const uint4 r0 = reg[ri];
#if KH_TUNE_SKM_PREFETCH
                    uint4 q1 = ri + 1 < nrec ? reg[ri + 1] : make_uint4(0, 0, 0, 0);   // the record behind it, in flight
#endif
                    u64 clo = ((u64)r0.y << 32) | r0.x, chi = ((u64)r0.w << 32) | r0.z;
                    u32 cn = r0.w >> 27, ctag = (r0.w >> 21) & 63u;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        kreg[e] = EMPTY;
                        slot_[e] = 0;
                        if (j0 + (u32)e < N) {
                            if (o == cn) {
                                ++ri;
#if KH_TUNE_SKM_PREFETCH
                                uint4 r = q1;
                                q1 = make_uint4(0, 0, 0, 0);
                                if (!(r.w >> 27)) r = reg[ri < nrec ? ri : nrec - 1];   // past the preloaded one (a record holds n >= 1)
#else
                                const uint4 r = reg[ri < nrec ? ri : nrec - 1];   // an L2 hit: the records were read a moment ago
#endif
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
                            const u32 h = key_hash(can);
                            kreg[e] = can;
                            slot_[e] = h >> (32 - HBITS);
                            tagp[e >> 2] |= ctag << (8 * (e & 3));
                            if (R == 1 || (((h >> 4) & 0xffffu) * R) >> 16 == q) act |= 1u << e;
                            ++o;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) { kreg[e] = EMPTY; slot_[e] = 0; }
                }
                auto tag = [&](int e) -> u32 { return (tagp[e >> 2] >> (8 * (e & 3))) & 63u; };
                SKM_STAMP(3);
                // ---- probe rounds, all of a thread's keys per round (one dependent LDS round trip per round)
                u32 was[E];   // the half of the genome mask that holds this key's bit, as it was: bit set already = a repeat inside the genome
#pragma unroll
                for (int e = 0; e < E; ++e) was[e] = 0u;
#define SKM_PROBE_ROUNDS(TBL, TMASK, ROUNDS)                                                                          \
    for (u32 round = 0; round < (ROUNDS) && __builtin_amdgcn_ballot_w64(act != 0); ++round) {                        \
        unsigned long long old[E];                                                                                    \
        _Pragma("unroll") for (int e = 0; e < E; ++e)                                                                 \
            old[e] = (act & (1u << e)) ? atomicCAS(&(TBL).key[slot_[e]], emptyv, (unsigned long long)kreg[e]) : 0ull;  \
        _Pragma("unroll") for (int e = 0; e < E; ++e) {                                                               \
            if (act & (1u << e)) {                                                                                    \
                if (old[e] == emptyv || old[e] == kreg[e]) {                                                           \
                    was[e] = atomicOr(((tag(e) & 32u) ? (TBL).mhi : (TBL).mlo) + slot_[e],                            \
                                      1u << (tag(e) & 31u));   /* looked at after the last round */                   \
                    act &= ~(1u << e);                                                                                \
                } else {                                                                                              \
                    slot_[e] = (slot_[e] + 1u) & (TMASK);                                                             \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    }
                SKM_PROBE_ROUNDS(tbl, T - 1u, (u32)KH_TUNE_SKM_FULL_ROUNDS)
                SKM_STAMP(9);
#if KH_TUNE_SKM_OVF_SERIAL
                // the few keys still homeless (~2 %): one per lane at a time, second table, then the main one again
                while (__builtin_amdgcn_ballot_w64(act != 0)) {
                    const bool have_one = act != 0;
                    const u32 es = have_one ? (u32)__builtin_ctz(act) : 0u;
                    u64 K = 0;
                    u32 tg = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (es == (u32)e) { K = kreg[e]; tg = tag(e); }
                    const u32 H = key_hash(K);
                    // level 0: the rest of the key's KH_HASH_ROUNDS probes in the main table, 1: second table, 2: main table, unbounded
                    u32 S = ((H ^ (H >> 15)) * 0x85EBCA77u) >> 24, probes = 0, level = 1, tmask = T2 - 1u;
                    if ((u32)KH_TUNE_SKM_FULL_ROUNDS < (u32)KH_HASH_ROUNDS) {
                        S = ((H >> (32 - HBITS)) + (u32)KH_TUNE_SKM_FULL_ROUNDS) & (T - 1u);
                        probes = (u32)KH_TUNE_SKM_FULL_ROUNDS; level = 0; tmask = T - 1u;
                    }
                    bool mine = have_one;
                    while (__builtin_amdgcn_ballot_w64(mine)) {
                        if (mine) {
                            unsigned long long* kp = level == 1 ? ovf.key : tbl.key;
                            const unsigned long long o2 = atomicCAS(&kp[S], emptyv, (unsigned long long)K);
                            if (o2 == emptyv || o2 == K) {
                                u32* mp = level == 1 ? ((tg & 32u) ? ovf.mhi : ovf.mlo) : ((tg & 32u) ? tbl.mhi : tbl.mlo);
                                const u32 w = atomicOr(mp + S, 1u << (tg & 31u));
                                if ((w >> (tg & 31u)) & 1u) atomicAdd(&dupc[tg], 1u);
                                mine = false;
                            } else {
                                ++probes;
                                if (level == 0 && probes >= (u32)KH_HASH_ROUNDS) {
                                    level = 1; probes = 0; tmask = T2 - 1u;
                                    S = ((H ^ (H >> 15)) * 0x85EBCA77u) >> 24;
                                } else if (level == 1 && probes >= T2) {
                                    level = 2; probes = 0; tmask = T - 1u;
                                    S = ((H >> (32 - HBITS)) + (u32)KH_HASH_ROUNDS) & (T - 1u);
                                } else if (level == 2 && probes >= T) {
                                    atomicOr(jb.ctl, KH_ERR_CAPACITY);   // cannot happen: a round holds at most T keys
                                    mine = false;
                                } else {
                                    S = (S + 1u) & tmask;
                                }
                            }
                        }
                    }
                    if (have_one) act &= ~(1u << es);
                }
#endif
                if (__builtin_amdgcn_ballot_w64(act != 0)) {
#pragma unroll
                    for (int e = 0; e < E; ++e) { const u32 h = key_hash(kreg[e]); slot_[e] = ((h ^ (h >> 15)) * 0x85EBCA77u) >> 24; }   // T2 = 256
                    SKM_PROBE_ROUNDS(ovf, T2 - 1u, T2)
                    if (__builtin_amdgcn_ballot_w64(act != 0)) {   // second table full of other keys: on in the main table
#pragma unroll
                        for (int e = 0; e < E; ++e) slot_[e] = ((key_hash(kreg[e]) >> (32 - HBITS)) + (u32)KH_HASH_ROUNDS) & (T - 1u);
                        SKM_PROBE_ROUNDS(tbl, T - 1u, T)
                        if (__builtin_amdgcn_ballot_w64(act != 0) && lane == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
                    }
                }
#undef SKM_PROBE_ROUNDS
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((was[e] >> (tag(e) & 31u)) & 1u) atomicAdd(&dupc[tag(e)], 1u);
                SKM_STAMP(4);
            }
        }
        __syncthreads();
        SKM_STAMP(5);
        // ---- every occupied entry is one distinct key of the slot: genome mask -> histogram bins
        // (the thread number through an opaque copy: addresses formed from it are worked out here, not kept in
        // registers across the insertion loop — where the compiler spilled them, 28 bytes of scratch per
        // thread = 1.5 GB of HBM writes per step)
        u32 tid_ro = tid;
        asm volatile("" : "+v"(tid_ro));
        u64 mk[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u32 i = (u32)e * NT + tid_ro;
            mk[e] = tbl.key[i] != EMPTY ? (((u64)tbl.mhi[i] << 32) | tbl.mlo[i]) : 0ull;
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
        for (u32 i = tid_ro; i < T2; i += NT) {   // keys that moved to the second table (a few per slot)
            if (ovf.key[i] != EMPTY) {
                const u64 emask = ((u64)ovf.mhi[i] << 32) | ovf.mlo[i];
                const u32 ng = eval_mask(emask, ginfo[__ffsll((unsigned long long)emask) - 1]);
                if (ng == 1u) ++ones;
                else atomicAdd(&hstripe[(jb.abase + ng) * 8u + (lane & 7u)], 1u);
            }
        }
        ones = wave_scan_add(ones);
        if (lane == KH_WAVE - 1 && ones) atomicAdd(&hstripe[(jb.abase + 1u) * 8u], ones);
        SKM_STAMP(6);
        __syncthreads();
        SKM_STAMP(7);
    }
    unsigned long long* __restrict__ rep = jb.hist + (u64)(blockIdx.x % jb.reps) * nbins;
    for (u32 i = tid; i < nbins; i += NT) {
        u32 v = 0;
#pragma unroll
        for (u32 j = 0; j < 8; ++j) v += hstripe[i * 8u + j];
        if (v) atomicAdd(&rep[i], (unsigned long long)v);
    }
    if (tid < (u32)KH_TAG_MAX_OPS && dupc[tid]) atomicAdd(&jb.dup[tid], (unsigned long long)dupc[tid]);
    SKM_STAMP(8);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <class K> static void skm_allow_lds(K kern, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <int WW> static void launch_scatter_w(const KhSkmJob& job, u32 ntiles, size_t lds, hipStream_t st) {
    skm_allow_lds(k_skm_scatter<WW>, lds);
    hipLaunchKernelGGL(k_skm_scatter<WW>, dim3(ntiles), dim3(SKM_NT), lds, st, job);
}
bool kh_skm_supports_w(u32 w) { return w >= (u32)SKM_WMIN && w <= (u32)SKM_WMAX; }
void kh_launch_skm_scatter(const KhSkmJob& job, u32 ntiles, hipStream_t st) {
    if (!ntiles) return;
    const size_t lds = kh_skm_scatter_lds_bytes(job.nb1);
    switch (job.w) {
#define SKM_W(WW) case WW: launch_scatter_w<WW>(job, ntiles, lds, st); break;
        SKM_W(5) SKM_W(6) SKM_W(7) SKM_W(8) SKM_W(9) SKM_W(10) SKM_W(11) SKM_W(12) SKM_W(13) SKM_W(14) SKM_W(15) SKM_W(16)
        SKM_W(17) SKM_W(18)
#undef SKM_W
        default: break;   // the host asks kh_skm_supports_w first
    }
}
void kh_launch_skm_regroup(const KhSkmJob& job, hipStream_t st) {
    const size_t lds = kh_skm_regroup_lds_bytes(job.S);
    skm_allow_lds(k_skm_regroup, lds);
    hipLaunchKernelGGL(k_skm_regroup, dim3(job.nb1), dim3(SKM_RG_NT), lds, st, job);
}
void kh_launch_skm_union(const KhSkmJob& job, u32 cs, hipStream_t st) {
    const size_t lds = kh_skm_union_lds_bytes(job.nbins);
    skm_allow_lds(k_skm_union, lds);
    hipLaunchKernelGGL(k_skm_union, dim3(job.nslots), dim3(SKM_UNT), lds, st, job, cs);
}
