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

// ISA study only (-DKH_MARKS with --cuda-device-only -S): comments that delimit the sections of the union in the assembly
#ifdef KH_MARKS
#define SKM_MARK(name) asm volatile("; ===== MARK " name ::: "memory")
#else
#define SKM_MARK(name) do {} while (0)
#endif
#ifndef KH_TUNE_SKM_FULL_ROUNDS
#define KH_TUNE_SKM_FULL_ROUNDS 4   // probe rounds made by all keys of a thread together; the rest one key per lane (1 / 2 / 3 / 4 / 5 / 8 rounds: union 16.3 / 3.33 / 1.59 / 1.51 / 1.52 / 1.55 ms)
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
        SkmSpill sp;
        sp.rec = jb.spill_rec; sp.slot = jb.spill_slot; sp.n = jb.ctl + 5; sp.cap = jb.spill_cap; sp.first_slot = first_slot;
        skm_flush<SKM_RG_NT, SKM_RG_CAP, true>(L, n, nfine, lcur, dst, jb.cap2, jb.ctl, sp);
    }
    for (u32 i = tid; i < nfine; i += SKM_RG_NT) jb.cur2[first_slot + i] = lcur[i];
}

// ------------------------------------------------------------------------------------------
// S2b: one slot per workgroup -> LDS hash set {canonical k-mer, genome mask} -> histogram bins.
//
// 1. Identical records are expanded ONCE.  A record is a content-defined piece of sequence (a run of k-mers with
//    one minimizer), so the genomes of a species hold the same records wherever they agree: the slot's records
//    (one per thread) meet in a small LDS hash set keyed by their bases; a record that finds its own content
//    already there ORs its genome bit into the first one's mask and retires (a copy inside the SAME genome: all
//    its k-mers are repeats).  Related genomes (the workload the reference is written for: groups of one species)
//    halve the k-mers that reach the big table; unrelated ones lose one LDS round trip.
// 2. Records -> k-mers: every surviving record is cut into chunks of 4 consecutive k-mers, a scan numbers the
//    chunks, a thread takes one chunk: first k-mer by a funnel shift + reversal of the 2-bit groups, the other
//    three by ROLLING both strands (32-bit halves: a funnel shift and a shift-or each); the keys of a thread share
//    one genome mask.
// 3. Insertion: rounds of 4 compare-and-swaps per thread.  A key gets KH_HASH_ROUNDS probes in the main table,
//    moves to a small second table with an independent hash, and from a full second table back to unbounded
//    probing of the main one; occupied entries stay occupied, so every copy of a key takes the same decisions as
//    the first.  (Measured and rejected: finishing the keys that lost round 1 one per lane in a loop — fewer
//    instructions, but a chain of ~20 dependent LDS round trips per wave: 2.63 ms against 2.30.)
// 4. Read-out: the thread whose compare-and-swap CREATED an entry remembers where; once all masks are final it
//    turns its own entries into histogram bins (popcount per group / number of groups).  Nobody scans the table,
//    no key is read again, waves without chunks have nothing to do.
//
// Geometry <NT, T>: threads and table entries of a workgroup.  <1024, 4096>: 79 KB of LDS, two workgroups per CU;
// <512, 2048>: 40 KB, four per CU (half-size slots: twice as many independent barrier chains per CU).
// ------------------------------------------------------------------------------------------
#ifndef KH_TUNE_SKM_UE
#define KH_TUNE_SKM_UE 2
#endif
constexpr u32 SKM_UE = KH_TUNE_SKM_UE;                // k-mers per chunk (2 or 4)
constexpr u32 SKM_OB = SKM_UE == 2 ? 4 : 3;           // bits of a chunk's number inside its record (n <= 31)
constexpr u32 SKM_PASSES = SKM_UE == 2 ? 3 : 2;       // chunks of a slot: at most SKM_PASSES per thread
constexpr u32 SKM_HSTRIPE_WORDS = 288;                // histogram copies in LDS: 4 / 2 / 1 per bin for <= 72 / 144 / 255 bins
template <u32 NT, u32 T> struct SkmUnionGeo {
    static constexpr u32 T2 = T >= 4096 ? 128 : 64;   // second table (a power of two; the main one need not be)
    static constexpr u32 MAXREC = NT;                 // records of a slot (cap2 <= this): one per thread
    static constexpr u32 MAXCH = SKM_PASSES * NT;     // chunks of a slot (after the merge of identical records)
    static constexpr size_t LDS = (size_t)T * 16 + (size_t)T2 * 16 + 1024 + 128 + 256 + (size_t)SKM_HSTRIPE_WORDS * 4 +
                                  (size_t)MAXCH * 2 + (size_t)MAXREC * 4;
    static_assert(MAXREC * 16 <= T * 4, "records are staged in the first half of the key plane");
    static_assert(T2 <= NT && T % 4 == 0, "table planes are cleared 16 bytes at a time");
};
u32 kh_skm_union_threads(u32 table) { return table == 2048 ? 512u : (table == 2560 ? 640u : 1024u); }
u32 kh_skm_union_max_cap2(u32 table) { return kh_skm_union_threads(table); }
u32 kh_skm_union_per_cu(u32 table) { return table == 4096 ? 2u : 3u; }
size_t kh_skm_union_lds_bytes(u32 table) {
    return table == 2048 ? SkmUnionGeo<512, 2048>::LDS : (table == 2560 ? SkmUnionGeo<640, 2560>::LDS : SkmUnionGeo<1024, 4096>::LDS);
}

__device__ __forceinline__ u32 key_hash2(u32 lo, u32 hi) { return (lo ^ hi) * 0x9E3779B1u; }

#ifdef KH_STAMPS
#define SKM_USTAMP(idx)                                                                    \
    do {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (threadIdx.x == 0 && g_skm_stamps) g_skm_stamps[(u64)slot * 16 + (idx)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    } while (0)
#else
#define SKM_USTAMP(idx) do {} while (0)
#endif

// PERSISTENT: the grid is two workgroups per CU (what the LDS allows); a workgroup walks slots blockIdx.x,
// blockIdx.x + gridDim.x, ...  While it works on one slot the record of the next one is already on its way to the
// thread's registers and the count of the one after to a scalar register: a workgroup that started with "load the
// count, then load the records" spent 2.5 of its 6 microseconds waiting for memory with nothing else to do.  The
// histogram bins and the repeat counters stay in LDS over all slots of the workgroup and are written once.
template <u32 NT, u32 T>
__global__ __launch_bounds__(NT, 8) void k_skm_union(const KhSkmJob jb, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    using G = SkmUnionGeo<NT, T>;
    constexpr u32 T2 = G::T2, NW = NT / 64;
    constexpr u32 T2SH = T2 == 128 ? 25 : 26;   // 32 - log2(T2)
    constexpr bool TPOW2 = (T & (T - 1u)) == 0u;   // (a power of two wraps by a mask; 2560 by compare)
    constexpr int E = (int)SKM_UE;
    constexpr u64 EMPTY = ~0ull;   // never a canonical key: the reverse complement of the all-T k-mer is 0
    // Table planes: keys (8-byte stride), low and high halves of the genome masks (4-byte stride).
    u8* p = lds_raw;
    unsigned long long* tkey = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T * 8;
    u32* tmlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    u32* tmhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    unsigned long long* okey = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T2 * 8;
    u32* omlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    u32* omhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    uint4* gtab = reinterpret_cast<uint4*>(p);                             p += 1024;   // per operand: its group's mask (two halves), first bin << sshift
    u32* scratch = reinterpret_cast<u32*>(p);                              p += 128;
    u32* dupc = reinterpret_cast<u32*>(p);                                 p += 256;
    u32* hstripe = reinterpret_cast<u32*>(p);                              p += (size_t)SKM_HSTRIPE_WORDS * 4;
    u16* owner = reinterpret_cast<u16*>(p);                                p += (size_t)G::MAXCH * 2;   // chunk -> record << SKM_OB | chunk of the record
    u32* rmask = reinterpret_cast<u32*>(p);                                // [MAXREC] genomes (of one half of the mask) that hold the record
    // While identical records are merged the table is not in use yet (and its KEY plane is not read any more once a
    // slot's insertions are over): the records are staged in the first half of the key plane, the set of their
    // contents (record number + 1, 0: empty) is the second half.
    uint4* stage = reinterpret_cast<uint4*>(tkey);
    u32* dd = reinterpret_cast<u32*>(tkey) + T;
    const u32 tid0 = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    u32 tid = tid0, lane = lane_id();
    const u32 nbins = jb.nbins, cap2 = jb.cap2, nslots = jb.nslots, stride = gridDim.x;
    const int k = jb.k;
    const u32 sshift = nbins <= 72u ? 2u : (nbins <= 144u ? 1u : 0u), smask = (1u << sshift) - 1u;
    auto clear_keys = [&]() {
        uint4* k4 = reinterpret_cast<uint4*>(tkey);
        for (u32 i = tid; i < T / 2; i += NT) k4[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    };
    auto clear_masks = [&]() {
        for (u32 i = tid; i < T / 2; i += NT) reinterpret_cast<uint4*>(tmlo)[i] = make_uint4(0u, 0u, 0u, 0u);   // (both mask planes)
        unsigned long long e0 = EMPTY;   // (opaque, as `emptyv` below)
        asm volatile("" : "+v"(e0));
        if (tid < T2) { okey[tid] = e0; omlo[tid] = 0u; omhi[tid] = 0u; }
    };
    // the record counts were written by the regroup kernel and do not change here: read through the constant address
    // space, i.e. by SCALAR loads (inside the slot loop the compiler cannot prove that for a plain global pointer)
    typedef const u32 __attribute__((address_space(4))) * ConstU32;
    const ConstU32 counts = (ConstU32)(unsigned long long)jb.cur2;
    // (a slot with more records than its region holds — the rest were spilled by the regroup — is not this kernel's:
    // it is listed for k_skm_big and counts as empty here)
    const u32 fit = cap2 < G::MAXREC ? cap2 : G::MAXREC;
    auto count_of = [&](u32 sl) -> u32 {
        const u32 n = sl < nslots ? counts[sl] : 0u;
        return n <= fit ? n : 0u;
    };
    SKM_MARK("init");
    if (tid < (u32)KH_TAG_MAX_OPS) {
        const u32 g = jb.ginfo[tid], g0 = g & 0xffu, gn = (g >> 8) & 0xffu;
        const u64 gm = gn ? (gn >= 64u ? ~0ull : ((1ull << gn) - 1ull)) << g0 : 0ull;
        gtab[tid] = make_uint4((u32)gm, (u32)(gm >> 32), (g >> 16) << sshift, 0u);
        dupc[tid] = 0;
    }
    if (tid < SKM_HSTRIPE_WORDS) hstripe[tid] = 0;
    if (tid == 0) scratch[0] = 0;
    // the 2k-bit mask and where the last base of a k-mer sits, as 32-bit halves (15 <= k <= 32)
    const u32 kml = (u32)kh_mask(2 * k), kmh = (u32)(kh_mask(2 * k) >> 32);
    const u32 fsh = 64u - 2u * (u32)k;          // right-aligns a reversed window
    const u32 tsh = 2u * (u32)k - 2u;           // 28 .. 62
    const bool tsh_high = tsh >= 32u;           // (uniform) the last base of a k-mer sits in the high word
    const u32 tsh_sub = tsh_high ? tsh - 32u : tsh;
    // one distinct k-mer: popcount of its mask per group -> the group's bin; number of groups -> across bin
    // (returns true when the k-mer sits in exactly one group — nearly all do: those are counted per wave)
    auto eval_mask = [&](u32 mlo, u32 mhi) -> bool {
        const u32 lsel = lane & smask;
        u32 ng = 0;
        do {
            const u32 first = mlo ? (u32)__builtin_ctz(mlo) : 32u + (u32)__builtin_ctz(mhi);
            const uint4 g = gtab[first];
            u32 c = (u32)__popc(mlo & g.x) + (u32)__popc(mhi & g.y);
            c = c < cs ? c : cs;
            atomicAdd(&hstripe[g.z + (c << sshift) + lsel], 1u);
            const u32 keep_hi = mlo ? ~0u : mhi - 1u;   // (the lowest bit goes in any case: a tag outside every group cannot hang the loop)
            mlo &= ~g.x & (mlo - 1u);
            mhi &= ~g.y & keep_hi;
            ++ng;
        } while (mlo | mhi);
        if (ng == 1u) return true;
        atomicAdd(&hstripe[((jb.abase + (ng < cs ? ng : cs)) << sshift) + lsel], 1u);
        return false;
    };
    u32 st_full = 0, st_exp = 0;   // fullest slot seen, k-mer instances expanded (uniform)
    // ---- pipeline prologue: this slot's record, the next slot's count
    u32 slot = blockIdx.x;
    u32 nrec = count_of(slot);
    u32 nrec_next = count_of(slot + stride);
    uint4 rr = tid < nrec ? (jb.reg2 + (u64)slot * cap2)[tid] : make_uint4(0, 0, 0, 0);
    for (; slot < nslots; slot += stride) {
        // (the thread number through an opaque copy, per slot: the many LDS addresses formed from it are worked out
        // where they are used instead of being kept in registers — and spilled — across the whole loop)
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & (KH_WAVE - 1u);
        const uint4* __restrict__ reg = jb.reg2 + (u64)slot * cap2;
        const u32 nrec_after = count_of(slot + 2u * stride);   // (a scalar load: two slots ahead)
        if (!nrec && counts[slot] > fit) {   // uniform: an overfull slot
            if (tid0 == 0) {
                const u32 at = atomicAdd(jb.ctl + 6, 1u);
                if (at < jb.big_cap) jb.big_list[at] = slot;
            }
        }
        // ---- stage this slot's records, one per thread
        u32 nj = 0;   // k-mers of this thread's record while it is alive
        const u32 tg = (rr.w >> 21) & 63u;
        const u32 rx = rr.x, ry = rr.y, rz = rr.z, rw = rr.w;   // (this slot's record; `rr` is loaded again below)
        {
            u32 z = 0;   // (opaque: a zero kept in four registers across the loop was spilled)
            asm volatile("" : "+v"(z));
            for (u32 i = tid; i < T / 4; i += NT) reinterpret_cast<uint4*>(dd)[i] = make_uint4(z, z, z, z);
        }
        if (tid < nrec) {
            nj = rr.w >> 27;
            stage[tid] = rr;
            rmask[tid] = 1u << (tg & 31u);
        }
        SKM_USTAMP(0);
        __syncthreads();
        SKM_USTAMP(1);
        SKM_MARK("dedup");
        // ---- identical records meet: the first keeps its place, the others add their genome bit to its mask.  A
        // record that stays takes its chunks from a counter (chunks in the low half, k-mers in the high half) and
        // enters them into the chunk table at once: no block scan, no phase of its own.
        if (__builtin_amdgcn_ballot_w64(nj != 0)) {   // (waves without records go straight to the barrier)
            u32 h = rx * 0x9E3779B1u ^ ry * 0x85EBCA77u ^ rz * 0xC2B2AE3Du ^ (rw & ~(63u << 21)) * 0x27D4EB2Fu;
            h ^= h >> 15;
            h *= 0x2C1B3C6Du;
            u32 hp = (u32)(((u64)h * T) >> 32);
            const u32 nch = (nj + (u32)E - 1u) / (u32)E;
            bool pend = nj != 0, won = false;
            while (__builtin_amdgcn_ballot_w64(pend)) {
                if (pend) {
                    const u32 old = atomicCAS(&dd[hp], 0u, tid + 1u);
                    if (old == 0u) {   // the first record with this content
                        won = true;
                        pend = false;
                    } else {
                        const uint4 o = stage[old - 1u];
                        // the same bases, the same number of k-mers, a genome of the same half of the mask
                        if (o.x == rx && o.y == ry && o.z == rz && ((o.w ^ rw) & ~(31u << 21)) == 0u) {
                            const u32 bit = 1u << (tg & 31u);
                            const u32 was = atomicOr(&rmask[old - 1u], bit);
                            if (was & bit) {   // a second copy inside one genome: nj repeats
                                atomicAdd(&dupc[tg], nj);
                                rmask[tid] = 0u;   // (this record's own mask is unused: zero = "has counted repeats", should the slot be handed on below)
                            }
                            pend = false;
                        } else {
                            hp = TPOW2 ? ((hp + 1u) & (T - 1u)) : (hp + 1u == T ? 0u : hp + 1u);
                        }
                    }
                }
            }
            // the wave's records take consecutive chunks in lane order (their loads in the expansion stay
            // coalesced); the wave's share comes from one returning LDS atomic
            const u32 mine = won ? (nch | (nj << 16)) : 0u;
            const u32 incl = wave_scan_add(mine);
            u32 wbase = 0;
            if (lane == KH_WAVE - 1 && incl) wbase = atomicAdd(&scratch[0], incl);
            wbase = (u32)__builtin_amdgcn_readlane((int)wbase, KH_WAVE - 1);
            if (won) {
                const u32 cstart = (wbase + incl - mine) & 0xffffu;
                if (cstart + nch <= G::MAXCH) {
#pragma unroll
                    for (u32 cc = 0; cc < (1u << SKM_OB); ++cc)
                        if (cc < nch) owner[cstart + cc] = (u16)((tid << SKM_OB) | cc);
                }
            }
        }
        SKM_MARK("dedup_end");
        SKM_USTAMP(2);
        __syncthreads();   // the staged records and the set of contents are dead, the previous slot's masks are read: the table is made
        const u32 sc0 = scratch[0];   // chunks | k-mers << 16 of the merged records (reset behind the next barrier)
        if ((sc0 & 0xffffu) > G::MAXCH) {   // uniform, rare: the slot is handed to k_skm_big below; what the merge has counted is taken back
            if (tid < nrec && rmask[tid] == 0u) {
                const u32 w = stage[tid].w;
                atomicSub(&dupc[(w >> 21) & 63u], w >> 27);
            }
            __syncthreads();
        }
        // ---- the next slot's record sets out now, into the registers this slot's record has just left: it has the
        // expansion and the read-out to arrive
        rr = tid < nrec_next ? (jb.reg2 + (u64)(slot + stride) * cap2)[tid] : make_uint4(0, 0, 0, 0);
        clear_keys();
        clear_masks();
        u32 C = sc0 & 0xffffu, N = sc0 >> 16;
        SKM_USTAMP(3);
        SKM_MARK("after_scan");
        st_full = N > st_full ? N : st_full;
        st_exp += N;
        if (C > G::MAXCH) {   // uniform: more chunks than are numbered here (long runs of one minimizer): k_skm_big takes the slot
            if (tid == 0) {
                if (jb.big_list) {
                    const u32 at = atomicAdd(jb.ctl + 6, 1u);
                    if (at < jb.big_cap) jb.big_list[at] = slot;
                } else atomicOr(jb.ctl, KH_ERR_CAPACITY);
            }
            C = 0;
            N = 0;
        }
#if defined(KH_ABLATE) && KH_ABLATE == 1
        C = 0; N = 0;   // timing study: nothing is expanded
#endif
        __syncthreads();
        if (tid == 0) scratch[0] = 0;   // (everybody has read it; it is added to again behind the next slot's first barrier)
        SKM_MARK("owner_done");
        SKM_USTAMP(4);
        const u32 R = (N + T - 1) / T;   // key subsets handled one after the other (1 unless the slot is overfull)
        const u32 ctid = tid;   // (a rotated numbering of the waves — other SIMDs busy from slot to slot — changed nothing)
        for (u32 q = 0; q < R; ++q) {
            if (q) { clear_keys(); clear_masks(); __syncthreads(); }
            unsigned long long emptyv = EMPTY;   // (opaque: made here, or the compiler keeps the constant in two VGPRs across the loop and spills it)
            asm volatile("" : "+v"(emptyv));
            // entries this thread created, per pass: 16-bit fields (bit 15: set = valid, bit 14: second table)
            u32 made[SKM_PASSES][E / 2];
#pragma unroll
            for (u32 ps = 0; ps < SKM_PASSES; ++ps)
#pragma unroll
                for (int e = 0; e < E / 2; ++e) made[ps][e] = 0u;
            u32 ones = 0;   // keys that sit in exactly one group
#pragma unroll
            for (u32 pass = 0; pass < SKM_PASSES; ++pass) {
                if (pass * NT >= C) break;   // uniform
                const u32 c = pass * NT + ctid;
                u64 kreg[E];
                u32 slot_[E], act = 0, bits = 0, half = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) { kreg[e] = EMPTY; slot_[e] = 0; }
                if (c < C) {
                    const u32 o = owner[c], ri = o >> SKM_OB, first = (o & ((1u << SKM_OB) - 1u)) * (u32)E;
                    const uint4 r0 = reg[ri];   // an L2 hit: the records were read a moment ago
                    bits = rmask[ri];
                    half = (r0.w >> 26) & 1u;
                    const u32 left = (r0.w >> 27) - first;
                    const u32 cnt = left < (u32)E ? left : (u32)E;
                    const u64 clo = ((u64)r0.y << 32) | r0.x, chi = ((u64)r0.w << 32) | r0.z;
                    const u32 sh = 2u * first;   // 0, 2E, .. <= 60
                    const u64 lo = sh ? (clo >> sh) | (chi << (64u - sh)) : clo, hi = chi >> sh;
                    const u32 xl = (u32)lo & kml, xh = (u32)(lo >> 32) & kmh;   // the first k-mer, base j at bits 2j
                    const u64 fw = kh_revpairs64(((u64)xh << 32) | xl) >> fsh;
                    u32 fl = (u32)fw, fh = (u32)(fw >> 32), rl = ~xl & kml, rh = ~xh & kmh;
                    const u32 t = (u32)((lo >> tsh) | (hi << (64u - tsh)));   // bits 2e: the last base of the chunk's k-mer e
                    const u32 tc = ~t;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if (e) {   // roll both strands by one base
                            fh = __builtin_amdgcn_alignbit(fh, fl, 30) & kmh;
                            fl = ((fl << 2) | ((t >> (2 * e)) & 3u)) & kml;
                            rl = __builtin_amdgcn_alignbit(rh, rl, 2);
                            rh >>= 2;
                            if (tsh_high) rh |= ((tc >> (2 * e)) & 3u) << tsh_sub; else rl |= ((tc >> (2 * e)) & 3u) << tsh_sub;
                        }
                        const bool fwd = fh < rh || (fh == rh && fl < rl);
                        const u32 cl = fwd ? fl : rl, ch = fwd ? fh : rh;
                        const u32 h = key_hash2(cl, ch);
                        kreg[e] = ((u64)ch << 32) | cl;
                        slot_[e] = (u32)(((u64)h * T) >> 32);
                        if (R != 1 && (u32)e < cnt && (((h >> 4) & 0xffffu) * R) >> 16 == q) act |= 1u << e;
                    }
                    if (R == 1) act = (1u << cnt) - 1u;
                }
#if defined(KH_ABLATE) && KH_ABLATE == 2
                if (kreg[0] != 0x1234567ull) act = 0;   // timing study: expanded, not inserted
#endif
                SKM_MARK("expanded");
                if (!__builtin_amdgcn_ballot_w64(act != 0)) continue;   // a wave without a chunk
                u32* const mp = half ? tmhi : tmlo;
                // ---- probe rounds, all of a thread's keys per round (one dependent LDS round trip per round)
                u32 was[E];   // the mask half as it was: bits of this record's genomes set already = repeats inside those genomes
#pragma unroll
                for (int e = 0; e < E; ++e) was[e] = 0u;
                u32 mk = 0;   // keys whose compare-and-swap created the entry
                for (u32 round = 0; round < (u32)KH_TUNE_SKM_FULL_ROUNDS && __builtin_amdgcn_ballot_w64(act != 0); ++round) {
                    unsigned long long old[E];
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        old[e] = (act & (1u << e)) ? atomicCAS(&tkey[slot_[e]], emptyv, (unsigned long long)kreg[e]) : 0ull;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if (act & (1u << e)) {
                            const bool fresh = old[e] == emptyv;
                            if (fresh || old[e] == kreg[e]) {
                                was[e] = atomicOr(mp + slot_[e], bits);   // looked at after the last round
                                act &= ~(1u << e);
                                if (fresh) mk |= 1u << e;
                            } else {
                                slot_[e] = TPOW2 ? ((slot_[e] + 1u) & (T - 1u)) : (slot_[e] + 1u == T ? 0u : slot_[e] + 1u);
                            }
                        }
                    }
                }
                SKM_USTAMP(9);
                SKM_MARK("rounds_done");
                u32 f16[E];   // where each key's entry is, if this thread created it
#pragma unroll
                for (int e = 0; e < E; ++e) f16[e] = (mk >> e) & 1u ? (0x8000u | slot_[e]) : 0u;
                // the few keys still homeless: one per lane at a time, second table, then the main one again
                while (__builtin_amdgcn_ballot_w64(act != 0)) {
                    const bool have_one = act != 0;
                    const u32 es = have_one ? (u32)__builtin_ctz(act) : 0u;
                    u64 K = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (es == (u32)e) K = kreg[e];
                    const u32 H = key_hash2((u32)K, (u32)(K >> 32));
                    // level 1: second table, 2: main table, unbounded
                    u32 S = ((H ^ (H >> 15)) * 0x85EBCA77u) >> T2SH, probes = 0, level = 1;
                    bool mine = have_one;
                    u32 where = 0;
                    while (__builtin_amdgcn_ballot_w64(mine)) {
                        if (mine) {
                            unsigned long long* kp = level == 1 ? okey : tkey;
                            const unsigned long long o2 = atomicCAS(&kp[S], emptyv, (unsigned long long)K);
                            if (o2 == emptyv || o2 == K) {
                                u32* mp2 = level == 1 ? (half ? omhi : omlo) : mp;
                                u32 d = atomicOr(mp2 + S, bits) & bits;
                                while (d) { atomicAdd(&dupc[half * 32u + (u32)__builtin_ctz(d)], 1u); d &= d - 1u; }
                                if (o2 == emptyv) where = 0x8000u | (level == 1 ? 0x4000u : 0u) | S;
                                mine = false;
                            } else {
                                ++probes;
                                if (level == 1 && probes >= 8u) {   // a crowded second table: on in the main one
                                    level = 2; probes = 0;
                                    S = (u32)(((u64)H * T) >> 32) + (u32)KH_TUNE_SKM_FULL_ROUNDS;
                                    S = TPOW2 ? (S & (T - 1u)) : (S >= T ? S - T : S);
                                } else if (level == 2 && probes >= T) {
                                    atomicOr(jb.ctl, KH_ERR_CAPACITY);   // cannot happen: a round holds at most T keys
                                    mine = false;
                                } else {
                                    S = level == 1 ? ((S + 1u) & (T2 - 1u)) : (TPOW2 ? ((S + 1u) & (T - 1u)) : (S + 1u == T ? 0u : S + 1u));
                                }
                            }
                        }
                    }
                    if (have_one) {
                        act &= ~(1u << es);
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (es == (u32)e) f16[e] = where;
                    }
                }
                SKM_MARK("serial_done");
#pragma unroll
                for (int e = 0; e < E / 2; ++e) made[pass][e] = f16[2 * e] | (f16[2 * e + 1] << 16);
                {   // repeats inside a genome: the bit was set before this instance came
                    u32 d = 0;
#pragma unroll
                    for (int e = 0; e < E; ++e) d |= was[e] & bits;
                    if (d) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            u32 de = was[e] & bits;
                            while (de) { atomicAdd(&dupc[half * 32u + (u32)__builtin_ctz(de)], 1u); de &= de - 1u; }
                        }
                    }
                }
            }
            SKM_MARK("insert_done");
            SKM_USTAMP(5);
            __syncthreads();
            SKM_USTAMP(6);
            // ---- all masks are final: every thread turns the entries it created into histogram bins
#pragma unroll
            for (u32 pass = 0; pass < SKM_PASSES; ++pass) {
                u32 any = 0;
#pragma unroll
                for (int e = 0; e < E / 2; ++e) any |= made[pass][e];
#if defined(KH_ABLATE) && KH_ABLATE == 3
                if (any != 0x7654321u) any = 0;   // timing study: no read-out
#endif
                if (!__builtin_amdgcn_ballot_w64(any != 0)) continue;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 f = (made[pass][e >> 1] >> (16 * (e & 1))) & 0xffffu;
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
            SKM_MARK("readout_done");
            SKM_USTAMP(7);
            if (q + 1 < R) __syncthreads();
        }
        SKM_USTAMP(8);
        // ---- on to the next slot: its record has arrived in the meantime.  No barrier here: what the next slot writes
        // before its first barrier (staged records, the set of contents: the KEY plane; this thread's record mask)
        // is read by nobody in the read-out.
        nrec = nrec_next;
        nrec_next = nrec_after;
    }
    __syncthreads();
    tid = tid0;
    lane = tid & (KH_WAVE - 1u);
    unsigned long long* __restrict__ rep = jb.hist + (u64)(blockIdx.x % jb.reps) * nbins;
    for (u32 i = tid; i < nbins; i += NT) {
        u32 v = 0;
        for (u32 j = 0; j <= smask; ++j) v += hstripe[(i << sshift) + j];
        if (v) atomicAdd(&rep[i], (unsigned long long)v);
    }
    if (tid < (u32)KH_TAG_MAX_OPS && dupc[tid]) atomicAdd(&jb.dup[tid], (unsigned long long)dupc[tid]);
    if (tid == 0) {
        if (st_full > T) atomicMax(jb.ctl + 1, st_full);
        atomicAdd(jb.ctl + 3, st_exp);   // k-mer instances that were expanded
    }
}

// ------------------------------------------------------------------------------------------
// Overfull slots (one-word keys).  A minimizer that far more k-mers share than a hash predicts — poly-A, a tandem repeat's
// unit, an insertion sequence in 50 copies — fills its slot's region; the regroup puts what does not fit on a side
// list and the union leaves such slots alone.  Here one workgroup takes one of them whatever its size: the records
// in the region, then its records on the side list (found by a scan of the list: it is short), every k-mer into the
// table in rounds of key subsets; no merge of identical records, read-out by a scan of the table.  A handful of
// slots per run: nothing here is tuned, it only has to be right and to keep the run in the fast form.
// ------------------------------------------------------------------------------------------
constexpr u32 SKM_BIG_NT = 1024, SKM_BIG_T = 4096, SKM_BIG_T2 = 128, SKM_BIG_IDX = 4096;
constexpr u32 SKM_BIG_BATCH = 256, SKM_BIG_MAXCH = SKM_BIG_BATCH << SKM_OB;   // records numbered at a time; their chunks at most
size_t kh_skm_big_lds_bytes() {
    return (size_t)SKM_BIG_T * 16 + (size_t)SKM_BIG_T2 * 16 + 1024 + 128 + 256 + (size_t)SKM_HSTRIPE_WORDS * 4 +
           (size_t)SKM_BIG_MAXCH * 2 + (size_t)SKM_BIG_IDX * 4;
}
__global__ __launch_bounds__(SKM_BIG_NT) void k_skm_big(const KhSkmJob jb, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr u32 NT = SKM_BIG_NT, T = SKM_BIG_T, T2 = SKM_BIG_T2, HBITS = 12;
    constexpr int E = (int)SKM_UE;
    constexpr u64 EMPTY = ~0ull;
    u8* p = lds_raw;
    unsigned long long* tkey = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T * 8;
    u32* tmlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    u32* tmhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;
    unsigned long long* okey = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T2 * 8;
    u32* omlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    u32* omhi = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    uint4* gtab = reinterpret_cast<uint4*>(p);                             p += 1024;
    u32* scratch = reinterpret_cast<u32*>(p);                              p += 128;   // [0] chunks, [2] k-mers, [3] side-list records
    u32* dupc = reinterpret_cast<u32*>(p);                                 p += 256;
    u32* hstripe = reinterpret_cast<u32*>(p);                              p += (size_t)SKM_HSTRIPE_WORDS * 4;
    u16* owner = reinterpret_cast<u16*>(p);                                p += (size_t)SKM_BIG_MAXCH * 2;
    u32* sidx = reinterpret_cast<u32*>(p);                                 // [SKM_BIG_IDX] this slot's records on the side list
    const u32 tid = threadIdx.x, lane = lane_id();
    const u32 nbins = jb.nbins, cap2 = jb.cap2;
    const int k = jb.k;
    const u32 slot = jb.big_list[blockIdx.x];
    const u32 sshift = nbins <= 72u ? 2u : (nbins <= 144u ? 1u : 0u), smask = (1u << sshift) - 1u;
    const u32 kml = (u32)kh_mask(2 * k), kmh = (u32)(kh_mask(2 * k) >> 32), fsh = 64u - 2u * (u32)k, tsh = 2u * (u32)k - 2u;
    const bool tsh_high = tsh >= 32u;
    const u32 tsh_sub = tsh_high ? tsh - 32u : tsh;
    if (tid < (u32)KH_TAG_MAX_OPS) {
        const u32 g = jb.ginfo[tid], g0 = g & 0xffu, gn = (g >> 8) & 0xffu;
        const u64 gm = gn ? (gn >= 64u ? ~0ull : ((1ull << gn) - 1ull)) << g0 : 0ull;
        gtab[tid] = make_uint4((u32)gm, (u32)(gm >> 32), (g >> 16) << sshift, 0u);
        dupc[tid] = 0;
    }
    if (tid < SKM_HSTRIPE_WORDS) hstripe[tid] = 0;
    if (tid < 8) scratch[tid] = 0;
    __syncthreads();
    // ---- this slot's records on the side list
    u32 nspill_all = jb.ctl[5];
    nspill_all = nspill_all < jb.spill_cap ? nspill_all : jb.spill_cap;
    for (u32 i = tid; i < nspill_all; i += NT) {
        if (jb.spill_slot[i] == slot) {
            const u32 at = atomicAdd(&scratch[3], 1u);
            if (at < SKM_BIG_IDX) sidx[at] = i;
        }
    }
    __syncthreads();
    u32 nside = scratch[3];
    if (nside > SKM_BIG_IDX) {   // (more than this kernel indexes: the host falls back)
        if (tid == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
        nside = SKM_BIG_IDX;
    }
    const u32 nreg = jb.cur2[slot] < cap2 ? jb.cur2[slot] : cap2;   // (full for an overfull slot; a slot listed for its chunks may hold fewer)
    const uint4* __restrict__ reg = jb.reg2 + (u64)slot * cap2;
    const u32 nall = nreg + nside;
    auto record = [&](u32 i) -> uint4 { return i < nreg ? reg[i] : jb.spill_rec[sidx[i - nreg]]; };
    // ---- k-mer instances of the slot -> rounds
    {
        u32 mine = 0;
        for (u32 i = tid; i < nall; i += NT) mine += record(i).w >> 27;
        const u32 tot = wave_scan_add(mine);
        if (lane == KH_WAVE - 1 && tot) atomicAdd(&scratch[2], tot);
    }
    __syncthreads();
    const u32 N = scratch[2];
    const u32 R = (N + 3071u) / 3072u;
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
        {
            uint4* k4 = reinterpret_cast<uint4*>(tkey);
#pragma unroll
            for (u32 e = 0; e < T / 2 / NT; ++e) k4[e * NT + tid] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            reinterpret_cast<uint4*>(tmlo)[tid] = make_uint4(0u, 0u, 0u, 0u);
            reinterpret_cast<uint4*>(tmhi)[tid] = make_uint4(0u, 0u, 0u, 0u);
            if (tid < T2) { okey[tid] = EMPTY; omlo[tid] = 0u; omhi[tid] = 0u; }
        }
        __syncthreads();
        for (u32 b0 = 0; b0 < nall; b0 += SKM_BIG_BATCH) {   // batches of records, one per thread of the first waves
            const u32 mine_i = b0 + tid;
            const u32 nj = tid < SKM_BIG_BATCH && mine_i < nall ? record(mine_i).w >> 27 : 0u;
            const u32 nch = (nj + (u32)E - 1u) / (u32)E;
            {
                const u32 incl = wave_scan_add(nch);
                u32 wbase = 0;
                if (lane == KH_WAVE - 1 && incl) wbase = atomicAdd(&scratch[0], incl);
                wbase = (u32)__builtin_amdgcn_readlane((int)wbase, KH_WAVE - 1);
                const u32 cstart = wbase + incl - nch;
                if (cstart + nch <= SKM_BIG_MAXCH) {
#pragma unroll
                    for (u32 cc = 0; cc < (1u << SKM_OB); ++cc)
                        if (cc < nch) owner[cstart + cc] = (u16)((tid << SKM_OB) | cc);
                }
            }
            __syncthreads();
            u32 C = scratch[0];
            if (C > SKM_BIG_MAXCH) {   // (cannot happen: a record has at most 1 << SKM_OB chunks)
                if (tid == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
                C = 0;
            }
            for (u32 c = tid; c < C; c += NT) {
                const u32 o = owner[c], ri = b0 + (o >> SKM_OB), first = (o & ((1u << SKM_OB) - 1u)) * (u32)E;
                const uint4 r0 = record(ri);
                const u32 tg = (r0.w >> 21) & 63u, bit = 1u << (tg & 31u), half = tg >> 5;
                const u32 left = (r0.w >> 27) - first;
                const u32 cnt = left < (u32)E ? left : (u32)E;
                const u64 clo = ((u64)r0.y << 32) | r0.x, chi = ((u64)r0.w << 32) | r0.z;
                const u32 sh = 2u * first;
                const u64 lo = sh ? (clo >> sh) | (chi << (64u - sh)) : clo, hi = chi >> sh;
                const u32 xl = (u32)lo & kml, xh = (u32)(lo >> 32) & kmh;
                const u64 fw = kh_revpairs64(((u64)xh << 32) | xl) >> fsh;
                u32 fl = (u32)fw, fh = (u32)(fw >> 32), rl = ~xl & kml, rh = ~xh & kmh;
                const u32 t = (u32)((lo >> tsh) | (hi << (64u - tsh)));
                const u32 tc = ~t;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (e) {
                        fh = __builtin_amdgcn_alignbit(fh, fl, 30) & kmh;
                        fl = ((fl << 2) | ((t >> (2 * e)) & 3u)) & kml;
                        rl = __builtin_amdgcn_alignbit(rh, rl, 2);
                        rh >>= 2;
                        if (tsh_high) rh |= ((tc >> (2 * e)) & 3u) << tsh_sub; else rl |= ((tc >> (2 * e)) & 3u) << tsh_sub;
                    }
                    if ((u32)e >= cnt) break;
                    const bool fwd = fh < rh || (fh == rh && fl < rl);
                    const u32 cl = fwd ? fl : rl, ch = fwd ? fh : rh;
                    const unsigned long long K = ((u64)ch << 32) | cl;
                    const u32 H = key_hash2(cl, ch);
                    if (R != 1 && (((H >> 4) & 0xffffu) * R) >> 16 != q) continue;
                    u32 S = H >> (32 - HBITS), probes = 0, level = 0;
                    while (true) {
                        unsigned long long* kp = level == 1 ? okey : tkey;
                        const unsigned long long o2 = atomicCAS(&kp[S], EMPTY, K);
                        if (o2 == EMPTY || o2 == K) {
                            u32* mp = level == 1 ? (half ? omhi : omlo) : (half ? tmhi : tmlo);
                            if (atomicOr(mp + S, bit) & bit) atomicAdd(&dupc[tg], 1u);   // this genome had the k-mer already
                            break;
                        }
                        ++probes;
                        if (level == 0 && probes >= (u32)KH_TUNE_SKM_FULL_ROUNDS) {
                            level = 1; probes = 0;
                            S = ((H ^ (H >> 15)) * 0x85EBCA77u) >> (32 - HBITS + 5);
                        } else if (level == 1 && probes >= 8u) {
                            level = 2; probes = 0;
                            S = ((H >> (32 - HBITS)) + (u32)KH_TUNE_SKM_FULL_ROUNDS) & (T - 1u);
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
        // ---- read-out: every occupied entry
        u32 ones = 0;
#pragma unroll
        for (u32 e = 0; e < T / NT; ++e) {
            const u32 i = e * NT + tid;
            if (tkey[i] != EMPTY && eval_mask(tmlo[i], tmhi[i])) ++ones;
        }
        if (tid < T2 && okey[tid] != EMPTY && eval_mask(omlo[tid], omhi[tid])) ++ones;
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
// The exchange form (multi-GPU step 7-8): slots are a GLOBAL function of the minimizer, so ranks
// exchange RECORDS by slot range instead of key sets.
//
//   k_skm_pack     one workgroup per slot of this rank's records (tag = local group, < 32): identical records of a
//                  group are the same piece of sequence in several of its genomes — only one travels; identical
//                  records of different groups merge into one with a mask of groups.  The survivors go, with their
//                  32-bit masks, to the record array of the rank that owns the slot (one returning global atomic
//                  per slot); per slot: how many, and where.
//   k_skm_phased   persistent, one slot at a time: the pieces (one per source rank) are PHASES.  A phase expands
//                  its records and ORs their masks into the table's low mask plane; between phases every entry
//                  folds popcount(mask) into its counter (the high plane) and clears the mask: tags of different
//                  phases are different groups, so the counter ends as "in how many groups of all ranks".
// ------------------------------------------------------------------------------------------
// (NT threads = the most records a slot may hold: the host picks 256 / 512 / 1024 from the regions' capacity — a
// workgroup per slot costs ~5 us whatever it holds, and four or eight of the small ones fit a CU instead of two)
static size_t skm_pack_lds_bytes(u32 nt) { return (size_t)nt * 16 + (size_t)nt * 4 * 4 + (size_t)nt * 4 + 64; }
size_t kh_skm_pack_lds_bytes() { return skm_pack_lds_bytes(1024); }

template <u32 NT>
__global__ __launch_bounds__(NT) void k_skm_pack(const KhSkmPackJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr u32 T = 4 * NT, TSH = NT == 1024 ? 20u : (NT == 512 ? 21u : 22u);
    static_assert(NT == 1024 || NT == 512 || NT == 256, "table of 4 NT entries, a power of two");
    uint4* stage = reinterpret_cast<uint4*>(lds_raw);
    u32* dd = reinterpret_cast<u32*>(lds_raw + (size_t)NT * 16);
    u32* rmask = dd + T;
    u32* scratch = rmask + NT;
    const u32 tid = threadIdx.x, lane = lane_id();
    const u32 slot = blockIdx.x;
    u32 nrec = jb.cur2[slot];
    nrec = nrec < jb.cap2 ? nrec : jb.cap2;
    if (nrec > NT) {   // (the host sizes the slots so that this does not happen)
        if (tid == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
        nrec = NT;
    }
    const uint4 rr = tid < nrec ? (jb.reg2 + (u64)slot * jb.cap2)[tid] : make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(dd)[tid] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) { scratch[0] = 0; scratch[1] = 0; }
    const u32 tg = (rr.w >> 21) & 63u;
    u32 nj = 0;
    if (tid < nrec) {
        nj = rr.w >> 27;
        stage[tid] = rr;
        rmask[tid] = 1u << (tg & 31u);
        if (tg >= 32u) atomicOr(jb.ctl, KH_ERR_ORDER);   // tags of the exchange form are below 32
    }
    __syncthreads();
    bool won = false;
    if (__builtin_amdgcn_ballot_w64(nj != 0)) {
        u32 h = rr.x * 0x9E3779B1u ^ rr.y * 0x85EBCA77u ^ rr.z * 0xC2B2AE3Du ^ (rr.w & ~(63u << 21)) * 0x27D4EB2Fu;
        h ^= h >> 15;
        h *= 0x2C1B3C6Du;
        u32 hp = h >> TSH;
        bool pend = nj != 0;
        while (__builtin_amdgcn_ballot_w64(pend)) {
            if (pend) {
                const u32 old = atomicCAS(&dd[hp], 0u, tid + 1u);
                if (old == 0u) { won = true; pend = false; }
                else {
                    const uint4 o = stage[old - 1u];
                    if (o.x == rr.x && o.y == rr.y && o.z == rr.z && ((o.w ^ rr.w) & ~(31u << 21)) == 0u) {
                        const u32 bit = 1u << (tg & 31u);
                        const u32 was = atomicOr(&rmask[old - 1u], bit);
                        if ((was & bit) && jb.dup) atomicAdd(&jb.dup[tg & 31u], (unsigned long long)nj);   // a second copy under one tag
                        pend = false;
                    } else hp = (hp + 1u) & (T - 1u);
                }
            }
        }
    }
    // ---- the survivors, in thread order, to the array of the rank that owns the slot
    const u32 incl = wave_scan_add(won ? 1u : 0u);
    u32 wbase = 0;
    if (lane == KH_WAVE - 1 && incl) wbase = atomicAdd(&scratch[0], incl);
    wbase = (u32)__builtin_amdgcn_readlane((int)wbase, KH_WAVE - 1);
    __syncthreads();   // all masks are final, the slot's total is known
    const u32 nsurv = scratch[0];
    const u32 part = slot / jb.spp;
    if (tid == 0) {
        const u32 sub = slot % jb.nsub;
        const u64 subcap = jb.part_cap / jb.nsub;
        u32 base = nsurv ? atomicAdd(&jb.part_cursor[part * jb.nsub + sub], nsurv) : 0u;
        if ((u64)base + nsurv > subcap) { atomicOr(jb.ctl, KH_ERR_CAPACITY); base = 0; scratch[0] = 0; }
        base += (u32)(sub * subcap);
        scratch[1] = base;
        jb.slot_count[slot] = nsurv;
        jb.slot_off[slot] = base;
    }
    __syncthreads();
    if (won && scratch[0]) {
        const u64 at = (u64)scratch[1] + wbase + incl - 1u;
        if (at < jb.part_cap) {
            jb.out_rec[(u64)part * jb.part_cap + at] = rr;
            jb.out_mask[(u64)part * jb.part_cap + at] = rmask[tid];
        }
    }
}

// A part that was packed through 64 cursors (one returning atomic per slot on a single cursor is a queue: 78 K
// workgroups, 0.8 ms) is moved together: workgroup (sub-range, part) copies its records behind those of the sub-ranges
// before it and rewrites the offsets of its slots.
__global__ __launch_bounds__(256) void k_skm_pack_compact(const KhSkmCompactJob jb) {
    const u32 sub = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    u32 before = 0, total = 0;
    for (u32 q = 0; q < jb.nsub; ++q) {   // uniform
        const u32 v = jb.cursors[part * jb.nsub + q];
        before += q < sub ? v : 0u;
        total += v;
    }
    const u32 n = jb.cursors[part * jb.nsub + sub];
    const u64 subcap = jb.part_cap / jb.nsub;
    const u64 src = (u64)part * jb.part_cap + (u64)sub * subcap, dst = (u64)part * jb.part_cap + before;
    for (u32 i = tid + 256u * blockIdx.z; i < n; i += 256u * gridDim.z) {   // (z: slices of the copy)
        jb.out_rec[dst + i] = jb.tmp_rec[src + i];
        jb.out_mask[dst + i] = jb.tmp_mask[src + i];
    }
    if (blockIdx.z) return;
    const u32 s0 = part * jb.spp, s1 = s0 + jb.spp < jb.nslots ? s0 + jb.spp : jb.nslots;
    const u32 first = s0 + (sub + jb.nsub - s0 % jb.nsub) % jb.nsub;   // the part's first slot with slot % nsub == sub
    for (u32 slot = first + tid * jb.nsub; slot < s1; slot += 256u * jb.nsub)
        jb.slot_off[slot] = jb.slot_off[slot] - (u32)((u64)sub * subcap) + before;
    if (sub == 0 && tid == 0) jb.part_n[part] = total;
}

constexpr u32 SKM_PH_NT = 1024, SKM_PH_T = 4096, SKM_PH_T2 = 128, SKM_PH_MAXCH = 4096, SKM_PH_HBINS = 512;
// A phase holds a few dozen records and its time is the insertion's chain of LDS round trips (ablation: 64 % of the
// kernel): ONE k-mer per thread — twice the threads of the union's two-k-mer chunks, half the chain.
constexpr u32 SKM_PH_E = 1, SKM_PH_OB = 5;   // k-mers per chunk; a record has at most 1 << SKM_PH_OB chunks
constexpr u32 SKM_PH_ROUND = 3072;   // k-mer instances a round of the phased union takes (the table has 4096 entries)
constexpr u32 SKM_PH_STAGED = 32;    // pieces whose records of a slot are numbered in one go (more: a phase at a time)
size_t kh_skm_phased_lds_bytes() {
    return (size_t)SKM_PH_T * 16 + (size_t)SKM_PH_T2 * 16 + 128 + (size_t)SKM_PH_HBINS * 4 + (size_t)SKM_PH_MAXCH * 2 +
           (size_t)SKM_PH_NT * 2 + (size_t)SKM_PH_STAGED * (8 + 8 + 4 * 5) + 64;
}

// A slot's time is a chain of memory latencies, not work (a phase holds a few dozen records): the records of ALL
// pieces are numbered in one go — thread t takes record t of the concatenated pieces, one ordered block scan gives
// every phase its range of chunks — and the phases that follow only insert and fold (two barriers each).  The
// per-piece counts and offsets of the NEXT slot are loaded a slot ahead.  A slot with more than 1024 records or
// 3072 chunks over all pieces (or more than 32 pieces) is numbered a phase at a time instead.
__global__ __launch_bounds__(SKM_PH_NT, 8) void k_skm_phased(const KhSkmPhasedJob jb) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr u32 NT = SKM_PH_NT, T = SKM_PH_T, T2 = SKM_PH_T2, HBITS = 12, NP = SKM_PH_STAGED;
    constexpr int E = (int)SKM_PH_E;
    constexpr u64 EMPTY = ~0ull;
    u8* p = lds_raw;
    unsigned long long* tkey = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T * 8;
    u32* tmlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;   // the running phase's mask
    u32* tcnt = reinterpret_cast<u32*>(p);                                 p += (size_t)T * 4;   // tags of the phases before it
    unsigned long long* okey = reinterpret_cast<unsigned long long*>(p);   p += (size_t)T2 * 8;
    u32* omlo = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    u32* ocnt = reinterpret_cast<u32*>(p);                                 p += (size_t)T2 * 4;
    u32* scratch = reinterpret_cast<u32*>(p);                              p += 128;   // [0] chunks of the phase, [1] entries made, [2] k-mers of the slot, [3] staged?, [8..23] wave totals
    u32* lhist = reinterpret_cast<u32*>(p);                                p += (size_t)SKM_PH_HBINS * 4;
    u16* owner = reinterpret_cast<u16*>(p);                                p += (size_t)SKM_PH_MAXCH * 2;
    u16* rloc = reinterpret_cast<u16*>(p);                                 p += (size_t)NT * 2;   // staged record t -> piece << 10 | record of the piece's slot
    const uint4** prec = reinterpret_cast<const uint4**>(p);               p += (size_t)NP * 8;   // the pieces' arrays (loaded once)
    const u32** pmsk = reinterpret_cast<const u32**>(p);                   p += (size_t)NP * 8;
    u32* pcnt = reinterpret_cast<u32*>(p);                                 p += (size_t)NP * 4;   // this slot: records per piece,
    u32* poff = reinterpret_cast<u32*>(p);                                 p += (size_t)NP * 4;   //   where they start,
    u32* cbeg = reinterpret_cast<u32*>(p);                                 p += (size_t)NP * 4;   //   the piece's chunks [cbeg, cend)
    u32* cend = reinterpret_cast<u32*>(p);                                 p += (size_t)NP * 4;
    u32* pflag = reinterpret_cast<u32*>(p);                                // dup_row << 1 | join_next
    const u32 tid0 = threadIdx.x;
    u32 tid = tid0, lane = lane_id();
    const int k = jb.k;
    const u32 kml = (u32)kh_mask(2 * k), kmh = (u32)(kh_mask(2 * k) >> 32), fsh = 64u - 2u * (u32)k, tsh = 2u * (u32)k - 2u;
    const bool tsh_high = tsh >= 32u;
    const u32 tsh_sub = tsh_high ? tsh - 32u : tsh;
    const u32 hbins = jb.hist_len < SKM_PH_HBINS ? jb.hist_len : SKM_PH_HBINS;   // counts below this: LDS; above: global atomics
    const bool want_dup = jb.dup != nullptr;
    const bool staged = jb.npieces <= NP;
    const u32 npieces = jb.npieces;
    typedef const u32 __attribute__((address_space(4))) * ConstU32;
    if (tid < SKM_PH_HBINS) lhist[tid] = 0;
    if (tid < 32) scratch[tid] = 0;
    const u32* my_count = nullptr;   // thread p < npieces: piece p's per-slot arrays
    const u32* my_off = nullptr;
    if (staged && tid < npieces) {
        const KhSkmPiece pc = jb.pieces[tid];
        prec[tid] = pc.rec;
        pmsk[tid] = pc.mask;
        pflag[tid] = (pc.dup_row << 1) | (pc.join_next & 1u);
        my_count = pc.count;
        my_off = pc.off;
    }
    u32 nxt_cnt = 0, nxt_off = 0;   // (thread p < npieces) piece p's records of the next slot
    if (my_count && blockIdx.x < jb.nslots) { nxt_cnt = my_count[blockIdx.x]; nxt_off = my_off[blockIdx.x]; }
    __syncthreads();
    unsigned long long emptyv = EMPTY;
    // ---- one chunk: E k-mers of record r0 from its k-mer `first` on, with the genomes / groups `bits`, enter the table
    auto insert_chunk = [&](const uint4 r0, const u32 bits, const u32 first, const u32 ph, const u32 R, const u32 q, u32& fresh_n) {
        const u32 left = (r0.w >> 27) - first;
        const u32 cnt = left < (u32)E ? left : (u32)E;
        const u64 clo = ((u64)r0.y << 32) | r0.x, chi = ((u64)r0.w << 32) | r0.z;
        const u32 sh = 2u * first;
        const u64 lo = sh ? (clo >> sh) | (chi << (64u - sh)) : clo, hi = chi >> sh;
        const u32 xl = (u32)lo & kml, xh = (u32)(lo >> 32) & kmh;
        const u64 fw = kh_revpairs64(((u64)xh << 32) | xl) >> fsh;
        u32 fl = (u32)fw, fh = (u32)(fw >> 32), rl = ~xl & kml, rh = ~xh & kmh;
        const u32 t = (u32)((lo >> tsh) | (hi << (64u - tsh)));
        const u32 tc = ~t;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (e) {
                fh = __builtin_amdgcn_alignbit(fh, fl, 30) & kmh;
                fl = ((fl << 2) | ((t >> (2 * e)) & 3u)) & kml;
                rl = __builtin_amdgcn_alignbit(rh, rl, 2);
                rh >>= 2;
                if (tsh_high) rh |= ((tc >> (2 * e)) & 3u) << tsh_sub; else rl |= ((tc >> (2 * e)) & 3u) << tsh_sub;
            }
            if ((u32)e >= cnt) break;
            const bool fwd = fh < rh || (fh == rh && fl < rl);
            const u32 cl = fwd ? fl : rl, ch = fwd ? fh : rh;
            const unsigned long long K = ((u64)ch << 32) | cl;
            const u32 H = key_hash2(cl, ch);
            if (R != 1 && (((H >> 4) & 0xffffu) * R) >> 16 != q) continue;   // another round's key
            // main table: KH_TUNE_SKM_FULL_ROUNDS probes, then the second table (8), then the main one to the end
            u32 S = H >> (32 - HBITS), probes = 0, level = 0;
            while (true) {
                unsigned long long* kp = level == 1 ? okey : tkey;
                const unsigned long long o2 = atomicCAS(&kp[S], emptyv, K);
                if (o2 == emptyv || o2 == K) {
                    if (want_dup) {   // uniform: bits set already = a second instance under that tag (rare: global counters)
                        u32 d = atomicOr((level == 1 ? omlo : tmlo) + S, bits) & bits;
                        while (d) { atomicAdd(&jb.dup[ph * 32u + (u32)__builtin_ctz(d)], 1ull); d &= d - 1u; }
                    } else atomicOr((level == 1 ? omlo : tmlo) + S, bits);
                    if (o2 == emptyv) ++fresh_n;
                    break;
                }
                ++probes;
                if (level == 0 && probes >= (u32)KH_TUNE_SKM_FULL_ROUNDS) {
                    level = 1; probes = 0;
                    S = ((H ^ (H >> 15)) * 0x85EBCA77u) >> (32 - HBITS + 5);
                } else if (level == 1 && probes >= 8u) {
                    level = 2; probes = 0;
                    S = ((H >> (32 - HBITS)) + (u32)KH_TUNE_SKM_FULL_ROUNDS) & (T - 1u);
                } else if (level == 2 && probes >= T) {
                    atomicOr(jb.ctl, KH_ERR_CAPACITY);   // the table is full
                    break;
                } else {
                    S = (S + 1u) & (level == 1 ? T2 - 1u : T - 1u);
                }
            }
        }
    };
    // ---- behind a phase's insertions: its tags are counted, the mask plane is free for the next phase
    auto count_fresh = [&](const u32 fresh_n) {
        if (__builtin_amdgcn_ballot_w64(fresh_n != 0)) {
            const u32 tot = wave_scan_add(fresh_n);
            if (lane == KH_WAVE - 1) atomicAdd(&scratch[1], tot);
        }
    };
    auto fold_now = [&]() {
        __syncthreads();
#if defined(KH_PH_ABLATE) && KH_PH_ABLATE == 2   // (timing study 2: barriers without the fold's work)
        __syncthreads();
        return;
#endif
        uint4 m4 = reinterpret_cast<uint4*>(tmlo)[tid];
        if (m4.x | m4.y | m4.z | m4.w) {
            uint4 c4 = reinterpret_cast<uint4*>(tcnt)[tid];
            c4.x += (u32)__popc(m4.x); c4.y += (u32)__popc(m4.y); c4.z += (u32)__popc(m4.z); c4.w += (u32)__popc(m4.w);
            reinterpret_cast<uint4*>(tcnt)[tid] = c4;
            reinterpret_cast<uint4*>(tmlo)[tid] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (tid < T2) { const u32 m = omlo[tid]; if (m) { ocnt[tid] += (u32)__popc(m); omlo[tid] = 0u; } }
        if (tid == 0) {
            scratch[0] = 0;
            if (scratch[1] > T - T / 16) atomicOr(jb.ctl, KH_ERR_CAPACITY);   // nearly full: probing would crawl
        }
        __syncthreads();
    };
    for (u32 slot = blockIdx.x; slot < jb.nslots; slot += gridDim.x) {
        tid = tid0;
        asm volatile("" : "+v"(tid));   // (addresses formed from it are worked out where they are used)
        lane = tid & (KH_WAVE - 1u);
        asm volatile("" : "+v"(emptyv));
        // ---- the slot's records over all pieces, its k-mer instances (more than a table takes -> rounds of key subsets)
        bool all_staged = staged;   // uniform
        u32 my_nj = 0, my_nch = 0, my_cstart = 0, Ctot = 0;
        if (staged) {
            if (tid < npieces) {
                pcnt[tid] = nxt_cnt;
                poff[tid] = nxt_off;
                cbeg[tid] = 0;
                cend[tid] = 0;
                const u32 ns = slot + gridDim.x;
                nxt_cnt = 0;
                if (ns < jb.nslots) { nxt_cnt = my_count[ns]; nxt_off = my_off[ns]; }   // (arrives during this slot)
            }
            __syncthreads();
            u32 acc = 0, myph = NP, myidx = 0, mycnt = 0;
            for (u32 ph = 0; ph < npieces; ++ph) {
                const u32 cn = pcnt[ph];
                if (tid >= acc && tid - acc < cn) { myph = ph; myidx = tid - acc; mycnt = cn; }
                acc += cn;
            }
            if (acc > NT) {
                all_staged = false;
                if (tid == 0) atomicMax(jb.ctl + 1, acc);
            } else {
                if (myph < NP) {
                    const u64 at = (u64)poff[myph] + myidx;
                    my_nj = prec[myph][at].w >> 27;
                    const u32 touch = pmsk[myph][at];   // (the mask's cache line sets out now)
                    asm volatile("" ::"v"(touch));
                    rloc[tid] = (u16)((myph << 10) | myidx);
                }
                my_nch = (my_nj + (u32)E - 1u) / (u32)E;
                // ordered block scan of (chunks, instances): chunk numbers follow the record order, i.e. the phases
                const u32 both = my_nch | (my_nj << 16);
                const u32 incl = wave_scan_add(both);
                if (lane == KH_WAVE - 1) scratch[8 + (tid >> 6)] = incl;
                __syncthreads();
                u32 before = 0, total = 0;
#pragma unroll
                for (u32 w = 0; w < NT / KH_WAVE; ++w) {
                    const u32 v = scratch[8 + w];
                    if (w < (tid >> 6)) before += v;
                    total += v;
                }
                Ctot = total & 0xffffu;
                my_cstart = ((before + incl - both) & 0xffffu);
                if (tid == 0) scratch[2] = total >> 16;
                if (Ctot > SKM_PH_MAXCH) {
                    all_staged = false;
                } else if (myph < NP) {
#pragma unroll
                    for (u32 cc = 0; cc < (1u << SKM_PH_OB); ++cc)
                        if (cc < my_nch) owner[my_cstart + cc] = (u16)((tid << SKM_PH_OB) | cc);
                    if (myidx == 0) cbeg[myph] = my_cstart;
                    if (myidx + 1 == mycnt) cend[myph] = my_cstart + my_nch;
                }
            }
            if (!all_staged && tid == 0) scratch[2] = 0;
            __syncthreads();
        }
        if (!all_staged) {   // instances of the slot, a piece at a time
            u32 mine = 0;
            for (u32 ph = 0; ph < npieces; ++ph) {
                const KhSkmPiece pc = jb.pieces[ph];
                u32 nrec = ((ConstU32)(unsigned long long)pc.count)[slot];
                nrec = nrec < NT ? nrec : NT;
                if (tid < nrec) mine += (pc.rec + ((ConstU32)(unsigned long long)pc.off)[slot])[tid].w >> 27;
            }
            if (__builtin_amdgcn_ballot_w64(mine != 0)) {
                const u32 tot = wave_scan_add(mine);
                if (lane == KH_WAVE - 1) atomicAdd(&scratch[2], tot);
            }
            __syncthreads();
        }
        const u32 R = (((scratch[2] * jb.share_q8) >> 8) + SKM_PH_ROUND - 1u) / SKM_PH_ROUND;   // (an optimistic share_q8 that overfills a table is caught at the fold: the host runs the launch again with 256)
        for (u32 q = 0; q < R; ++q) {
            {   // a fresh table (the read-out before it is behind a barrier)
                uint4* k4 = reinterpret_cast<uint4*>(tkey);
#pragma unroll
                for (u32 e = 0; e < T / 2 / NT; ++e) k4[e * NT + tid] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
                reinterpret_cast<uint4*>(tmlo)[tid] = make_uint4(0u, 0u, 0u, 0u);
                reinterpret_cast<uint4*>(tcnt)[tid] = make_uint4(0u, 0u, 0u, 0u);
                if (tid < T2) { okey[tid] = emptyv; omlo[tid] = 0u; ocnt[tid] = 0u; }
                if (tid == 0) { scratch[0] = 0; scratch[1] = 0; scratch[3] = 0; }
            }
            __syncthreads();
            // A phase = a piece and the pieces joined to it (join_next: the same tags — a sub-batch's side list): the fold
            // comes behind the last of them that holds records of this slot.
            bool open = false;   // uniform: insertions since the last fold
#if defined(KH_PH_ABLATE) && KH_PH_ABLATE == 3   // (timing study 3: no phases at all)
            if (false) {
#else
            if (all_staged) {
#endif
                for (u32 ph = 0; ph < npieces; ++ph) {
                    const u32 c0 = cbeg[ph], c1 = cend[ph], fl = pflag[ph];
                    if (c0 != c1) {   // uniform
                        const uint4* __restrict__ rec = prec[ph] + poff[ph];
                        const u32* __restrict__ msk = pmsk[ph] + poff[ph];
                        u32 fresh_n = 0;
#if !defined(KH_PH_ABLATE) || KH_PH_ABLATE != 1   // (timing study 1: no insertion)
                        for (u32 c = c0 + tid; c < c1; c += NT) {
                            const u32 o = owner[c], ri = rloc[o >> SKM_PH_OB] & 1023u;
                            insert_chunk(rec[ri], msk[ri], (o & ((1u << SKM_PH_OB) - 1u)) * (u32)E, fl >> 1, R, q, fresh_n);
                        }
#endif
                        count_fresh(fresh_n);
                        open = true;
                    }
                    if (open && !((fl & 1u) && ph + 1 < npieces)) { fold_now(); open = false; }
                }
            } else {
                constexpr u32 BATCH = SKM_PH_MAXCH >> SKM_PH_OB;   // records numbered at a time: their chunks fit the owner table
                u32 par = 0;
                for (u32 ph = 0; ph < npieces; ++ph) {
                    const KhSkmPiece pc = jb.pieces[ph];
                    u32 nrec = ((ConstU32)(unsigned long long)pc.count)[slot];
                    if (nrec > NT) {   // uniform
                        if (tid == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
                        nrec = NT;
                    }
                    const u64 roff = ((ConstU32)(unsigned long long)pc.off)[slot];
                    const uint4* __restrict__ rec = pc.rec + roff;
                    const u32* __restrict__ msk = pc.mask + roff;
                    for (u32 b0 = 0; b0 < nrec; b0 += BATCH) {
                        // ---- number the chunks of the batch's records
                        const u32 nj = tid < BATCH && b0 + tid < nrec ? rec[b0 + tid].w >> 27 : 0u;
                        const u32 nch = (nj + (u32)E - 1u) / (u32)E;
                        if (__builtin_amdgcn_ballot_w64(nch != 0)) {
                            const u32 incl = wave_scan_add(nch);
                            u32 wbase = 0;
                            if (lane == KH_WAVE - 1 && incl) wbase = atomicAdd(&scratch[par], incl);
                            wbase = (u32)__builtin_amdgcn_readlane((int)wbase, KH_WAVE - 1);
                            const u32 cstart = wbase + incl - nch;
#pragma unroll
                            for (u32 cc = 0; cc < (1u << SKM_PH_OB); ++cc)
                                if (cc < nch) owner[cstart + cc] = (u16)((tid << SKM_PH_OB) | cc);
                        }
                        __syncthreads();
                        const u32 C = scratch[par];   // (<= BATCH << SKM_PH_OB = SKM_PH_MAXCH)
                        par ^= 3u;                    // the batches' chunk counters alternate: [0] and [3]
                        if (tid == 0) scratch[par] = 0;   // (the next batch's: nobody reads it before the barrier below)
                        u32 fresh_n = 0;   // entries this thread created
                        for (u32 c = tid; c < C; c += NT) {
                            const u32 o = owner[c], ri = b0 + (o >> SKM_PH_OB);
                            insert_chunk(rec[ri], msk[ri], (o & ((1u << SKM_PH_OB) - 1u)) * (u32)E, pc.dup_row, R, q, fresh_n);
                        }
                        count_fresh(fresh_n);
                        open = true;
                        __syncthreads();   // the owner table is written again
                    }
                    if (open && !((pc.join_next & 1u) && ph + 1 < npieces)) { fold_now(); open = false; }
                }
            }
            // ---- every occupied entry is one distinct k-mer of the slot; its counter: in how many (phase, tag) pairs
            {
                const uint4 c4 = reinterpret_cast<uint4*>(tcnt)[tid];
                const u32 cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (cv[e]) {
                        u32 c = cv[e] < jb.cs ? cv[e] : jb.cs;
                        c = c < jb.hist_len - 1u ? c : jb.hist_len - 1u;
                        if (c < hbins) atomicAdd(&lhist[c], 1u);
                        else atomicAdd(&jb.hist[c], 1ull);
                    }
                }
                if (tid < T2 && ocnt[tid]) {
                    u32 c = ocnt[tid] < jb.cs ? ocnt[tid] : jb.cs;
                    c = c < jb.hist_len - 1u ? c : jb.hist_len - 1u;
                    if (c < hbins) atomicAdd(&lhist[c], 1u);
                    else atomicAdd(&jb.hist[c], 1ull);
                }
                if (tid == 0 && q + 1 == R) scratch[2] = 0;   // (everybody has worked R out, barriers ago)
            }
            __syncthreads();
        }
        if (!R) __syncthreads();   // (an empty slot: scratch[2] is written again only behind the next barrier anyway)
    }
    __syncthreads();
    if (tid0 < hbins && lhist[tid0]) atomicAdd(&jb.hist[tid0], (unsigned long long)lhist[tid0]);
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
void kh_launch_skm_union(const KhSkmJob& job, u32 cs, u32 grid, hipStream_t st) {
    const size_t lds = kh_skm_union_lds_bytes(job.table);
    if (job.table == 2048) {
        skm_allow_lds(k_skm_union<512, 2048>, lds);
        hipLaunchKernelGGL((k_skm_union<512, 2048>), dim3(grid), dim3(512), lds, st, job, cs);
    } else if (job.table == 2560) {
        skm_allow_lds(k_skm_union<640, 2560>, lds);
        hipLaunchKernelGGL((k_skm_union<640, 2560>), dim3(grid), dim3(640), lds, st, job, cs);
    } else {
        skm_allow_lds(k_skm_union<1024, 4096>, lds);
        hipLaunchKernelGGL((k_skm_union<1024, 4096>), dim3(grid), dim3(1024), lds, st, job, cs);
    }
}
void kh_launch_skm_pack(const KhSkmPackJob& job, hipStream_t st) {
    if (!job.nslots) return;
    if (job.cap2 <= 256) {
        hipLaunchKernelGGL(k_skm_pack<256>, dim3(job.nslots), dim3(256), skm_pack_lds_bytes(256), st, job);
    } else if (job.cap2 <= 512) {
        hipLaunchKernelGGL(k_skm_pack<512>, dim3(job.nslots), dim3(512), skm_pack_lds_bytes(512), st, job);
    } else {
        const size_t lds = skm_pack_lds_bytes(1024);
        skm_allow_lds(k_skm_pack<1024>, lds);
        hipLaunchKernelGGL(k_skm_pack<1024>, dim3(job.nslots), dim3(1024), lds, st, job);
    }
}
void kh_launch_skm_pack_compact(const KhSkmCompactJob& job, hipStream_t st) {
    if (!job.nparts || !job.nsub) return;
    const u32 z = job.nparts >= 16 ? 1u : 16u / job.nparts;   // about a thousand workgroups
    hipLaunchKernelGGL(k_skm_pack_compact, dim3(job.nsub, job.nparts, z), dim3(256), 0, st, job);
}
void kh_launch_skm_phased(const KhSkmPhasedJob& job, u32 grid, hipStream_t st) {
    if (!job.nslots || !grid) return;
    const size_t lds = kh_skm_phased_lds_bytes();
    skm_allow_lds(k_skm_phased, lds);
    hipLaunchKernelGGL(k_skm_phased, dim3(grid), dim3(SKM_PH_NT), lds, st, job);
}
static u32 big_y() { const char* e = getenv("KHOICE_SKM_BIG_Y"); const int v = e ? atoi(e) : 4; return (u32)(v < 1 ? 1 : (v > 16 ? 16 : v)); }
void kh_launch_skm_big(const KhSkmJob& job, u32 cs, u32 nbig, hipStream_t st) {
    if (!nbig) return;
    const size_t lds = kh_skm_big_lds_bytes();
    skm_allow_lds(k_skm_big, lds);
    hipLaunchKernelGGL(k_skm_big, dim3(nbig, nbig < 2048u ? big_y() : 1u), dim3(SKM_BIG_NT), lds, st, job, cs);   // y: the rounds of a slot side by side
}
