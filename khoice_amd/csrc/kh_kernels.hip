// khoice_amd — hand-written gfx950 (CDNA4, wave64) kernels for the k-mer hot path.
//
// Pipeline of one batched build (K1; replaces `kmc -fm -k{k} -ci1`, reference call site
// workflow/rules/exp_type_1.smk:163):
//   pass A  k_extract<W,false>   bases -> canonical k-mer -> mixed key -> per-tile bucket histogram
//           k_col_totals / k_exscan / k_col_offsets      bucket starts and per-tile write cursors
//   pass B  k_extract<W,true>    same extraction, keys scattered to their bucket (LDS cursors)
//   pass C  k_bucket_sort_rle<W> one workgroup per bucket: LDS bitonic sort, run-length count,
//                                ordered single-pass output (decoupled look-back)
// Set operations (K3/K5/K6; `kmc_tools complex|simple`, exp_type_1.smk:182, exp_type_2.smk:363-379):
//   k_range_bounds<W>            binary-search the slot boundaries of every operand
//   k_setop<W,PAY>               one workgroup per slot: gather operand slices into LDS, sort,
//                                combine counters per key, fused counter histogram (K4),
//                                ordered single-pass output
// All integer work; bounded by HBM bandwidth, not by MFMA (none is used).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kh_launch.h"
#include "kh_device.h"

// Diagnostic build only (-DKH_STAMPS, never the shipped library): thread 0 of every sort
// workgroup stores the shader clock at phase boundaries into a buffer of its own.
#ifdef KH_STAMPS
__device__ u64* g_kh_stamps = nullptr;
void kh_debug_set_stamps(u64* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kh_stamps), &p, sizeof p); }
#define KH_STAMP(q, idx)                                                              \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        if (threadIdx.x == 0 && g_kh_stamps)                                          \
            g_kh_stamps[((u64)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (idx)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                            \
    } while (0)
#else
#define KH_STAMP(q, idx) do {} while (0)
#endif


// rank of this thread among flagged threads of the block (exclusive) and the block total.
// wave_tot: LDS scratch of blockDim/64 words.  Contains two barriers.
__device__ __forceinline__ u32 block_rank(bool flag, u32* wave_tot, u32& total) {
    const u64 b = __ballot(flag);
    const u32 lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) wave_tot[wid] = (u32)__popcll(b);
    __syncthreads();
    u32 pre = 0, tot = 0;
    for (u32 w = 0; w < nw; ++w) {
        const u32 v = wave_tot[w];
        pre += (w < wid) ? v : 0u;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return pre + (u32)__popcll(b & ((1ull << lane) - 1ull));
}

// ---- ordered single-pass output: decoupled look-back over 64-bit {status:2, value:62} words.
// One 8-byte relaxed agent-scope store/load carries status and value together, so no
// separate flag ordering is needed.  Tickets are handed out in start order, hence every
// predecessor of a resident workgroup is resident or finished: no deadlock.  Spins are
// bounded; on timeout an error bit is raised and the launch drains.
constexpr u64 KH_LB_AGG = 1ull << 62;
constexpr u64 KH_LB_PREFIX = 2ull << 62;
constexpr u64 KH_LB_VALUE = (1ull << 62) - 1ull;

__device__ __forceinline__ void lb_store(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lb_load(u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// pass A / pass B: base decoding, rolling canonical k-mer, mixing, bucket histogram / scatter
// ------------------------------------------------------------------------------------------
constexpr int KH_CODE_WORDS = (KH_SUBTILE + KH_HALO) / 16;    // words of 16 bases

size_t kh_extract_lds_bytes(u32 nb_alloc) {
    return (size_t)nb_alloc * 4 + (size_t)KH_CODE_WORDS * 4 + (size_t)KH_CODE_WORDS * 2 + 16;
}


// The same in two steps, so that the global load of the NEXT round can be in flight while the
// current one is processed: ki_fetch -> registers (up to two words per thread), ki_store -> LDS.
struct CodeFetch { uint4 v[2]; u32 left[2]; };   // left: valid bytes of the word (0: past the end, >= 16: whole)
template <u32 NT>
__device__ __forceinline__ void fetch_codes(const u8* __restrict__ sbase, const u64 len, const u64 p0, const u32 nwords,
                                            CodeFetch& f) {
#pragma unroll
    for (u32 r = 0; r < 2; ++r) {
        const u32 w = threadIdx.x + r * NT;
        f.left[r] = 0;
        f.v[r] = make_uint4(0, 0, 0, 0);
        if (w < nwords) {
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
template <u32 NT>
__device__ __forceinline__ void store_codes(const CodeFetch& f, u32* code, u16* bad16, const u32 nwords) {
#pragma unroll
    for (u32 r = 0; r < 2; ++r) {
        const u32 w = threadIdx.x + r * NT;
        if (w < nwords) {
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

template <int W> struct Roller;
template <> struct Roller<1> {
    u64 mask, top_shift;
    __device__ Roller(int k) : mask(kh_mask(2 * k)), top_shift(2 * k - 2) {}
    __device__ __forceinline__ void push(KmerKey<1>& f, KmerKey<1>& r, u32 c) const {
        f.lo = ((f.lo << 2) | c) & mask;
        r.lo = (r.lo >> 2) | ((u64)(3u - c) << top_shift);
    }
};
template <> struct Roller<2> {
    u64 mask_hi, top_shift;   // hi holds 2k-64 bits
    __device__ Roller(int k) : mask_hi(kh_mask(2 * k - 64)), top_shift(2 * k - 64 - 2) {}
    __device__ __forceinline__ void push(KmerKey<2>& f, KmerKey<2>& r, u32 c) const {
        f.hi = ((f.hi << 2) | (f.lo >> 62)) & mask_hi;
        f.lo = (f.lo << 2) | c;
        r.lo = (r.lo >> 2) | (r.hi << 62);
        r.hi = (r.hi >> 2) | ((u64)(3u - c) << top_shift);
    }
};
template <int W> __device__ __forceinline__ KmerKey<W> key_zero();
template <> __device__ __forceinline__ KmerKey<1> key_zero<1>() { return KmerKey<1>{0}; }
template <> __device__ __forceinline__ KmerKey<2> key_zero<2>() { return KmerKey<2>{0, 0}; }


// Roller state after the m = k-1 bases in front of a thread's first k-mer end, straight from the
// packed window X (base i of the window at bits 2i) instead of m pushes:
//   forward word  = the window with the order of its bases reversed (first base most significant),
//   reverse word  = the complemented window, two bits up (every later push shifts it down by one base).
__device__ __forceinline__ void window_state(const u32 (&xw)[2], int m, KmerKey<1>& f, KmerKey<1>& r) {
    const u64 x = (((u64)xw[1] << 32) | xw[0]) & kh_mask(2 * m);
    f.lo = m ? kh_revpairs64(x) >> (64 - 2 * m) : 0ull;
    r.lo = ((~x) & kh_mask(2 * m)) << 2;
}
__device__ __forceinline__ void window_state(const u32 (&xw)[4], int m, KmerKey<2>& f, KmerKey<2>& r) {
    // 32 <= m <= 63 bases: 2m bits in (xhi, xlo)
    const u64 xlo = ((u64)xw[1] << 32) | xw[0];
    const u64 xhi = (((u64)xw[3] << 32) | xw[2]) & kh_mask(2 * m - 64);
    const u64 revhi = kh_revpairs64(xlo), revlo = kh_revpairs64(xhi);   // 128-bit group reversal
    const int s = 128 - 2 * m;                                           // 2 .. 64
    if (s >= 64) { f.lo = revhi; f.hi = 0; }
    else { f.lo = (revlo >> s) | (revhi << (64 - s)); f.hi = revhi >> s; }
    r.lo = (~xlo) << 2;
    r.hi = (((~xhi) & kh_mask(2 * m - 64)) << 2) | ((~xlo) >> 62);
}

// Canonical mixed keys of the PPT k-mers that start at bases p .. p + PPT - 1 of the sub-tile held
// in code / bad16 (p = PPT * thread): emit(j, key) is called for every start p + j whose k bases are
// all valid.  One window lookup, then exactly PPT pushes with compile-time bit positions.
template <int W, int PPT, class Emit>
__device__ __forceinline__ void extract_positions(const u32* code, const u16* bad16, const u32 p, const int k,
                                                  const Roller<W>& roller, Emit emit) {
    constexpr int NPRE = W == 1 ? 2 : 4;            // code words that hold the k-1 leading bases
    constexpr int NNEW = (PPT + 15) / 16;           // code words of the PPT bases that end the k-mers
    const int m = k - 1;
    const u32 w0 = p >> 4, o = p & 15u;
    const u32 q = p + (u32)m, wq = q >> 4, oq = q & 15u;
    // all LDS reads of the thread are issued before the first is used
    u32 cw[NPRE + 1], nw[NNEW + 1], bw[5], bn[3];
#pragma unroll
    for (int i = 0; i <= NPRE; ++i) cw[i] = code[w0 + i];
#pragma unroll
    for (int i = 0; i <= NNEW; ++i) nw[i] = code[wq + i];
#pragma unroll
    for (int i = 0; i < 5; ++i) bw[i] = bad16[w0 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) bn[i] = bad16[wq + i];
    u32 xw[NPRE], newc[NNEW];
#pragma unroll
    for (int i = 0; i < NPRE; ++i) xw[i] = __builtin_amdgcn_alignbit(cw[i + 1], cw[i], 2 * o);
#pragma unroll
    for (int i = 0; i < NNEW; ++i) newc[i] = __builtin_amdgcn_alignbit(nw[i + 1], nw[i], 2 * oq);
    const u64 bpre = ((((u64)bw[4] << 48) | ((u64)bw[3] << 32) | ((u64)bw[2] << 16) | (u64)bw[1]) << (16 - o)) |
                     ((u64)bw[0] >> o);             // flags of bases p .. p+63
    const u64 newbad = (((u64)bn[2] << 32) | ((u64)bn[1] << 16) | (u64)bn[0]) >> oq;   // flags of bases q .. q+PPT-1
    KmerKey<W> f, r;
    window_state(xw, m, f, r);
    // valid bases since the last break inside the window (the window itself when it has none)
    const u64 bm = bpre & kh_mask(m);
    int run = bm ? m - 64 + (int)__builtin_clzll(bm) : m;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const u32 c = (newc[j >> 4] >> (2 * (j & 15))) & 3u;
        roller.push(f, r, c);
        run = ((newbad >> j) & 1ull) ? 0 : run + 1;
        if (run >= k) {
            KmerKey<W> can = key_lt(r, f) ? r : f;
            emit(j, kh_mix(can, k));
        }
    }
}

template <int W, bool SCATTER>
__global__ __launch_bounds__(256) void k_extract(const u8* __restrict__ seq,
                                                const KhSeg* __restrict__ segs,
                                                const KhTile* __restrict__ tiles, u32 nb_alloc,
                                                int k, u32* __restrict__ thist,
                                                const u64* __restrict__ bstart,
                                                KmerKey<W>* __restrict__ part, u32 tile_pos) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    u32* cur = reinterpret_cast<u32*>(lds_raw);
    u32* code = cur + nb_alloc;
    u16* bad16 = reinterpret_cast<u16*>(code + KH_CODE_WORDS);

    const u32 tid = threadIdx.x;
    const KhTile t = tiles[blockIdx.x];
    const KhSeg sg = segs[t.seg];
    const u32 nb = sg.nbuckets;
    u32* row = thist + sg.thist_base + (u64)t.tile_in_seg * nb;

    for (u32 i = tid; i < nb; i += 256) cur[i] = SCATTER ? row[i] : 0u;
    u64 part_base = 0;
    if (SCATTER) part_base = bstart[sg.bucket_base];

    const Roller<W> roller(k);
    const u64 tile_pos0 = (u64)t.tile_in_seg * tile_pos;
    constexpr int PPT = KH_SUBTILE / 256;   // 32 start positions per thread
    const int subtiles = (int)(tile_pos / KH_SUBTILE);   // host: tile_pos is a multiple of 2 * KH_SUBTILE

    for (int sub = 0; sub < subtiles; ++sub) {
        const u64 p0 = tile_pos0 + (u64)sub * KH_SUBTILE;
        if (p0 >= sg.npos) break;   // uniform over the block
        __syncthreads();            // cursor init done / previous sub-tile fully consumed
        load_codes<256>(sg.seq, sg.len, p0, code, bad16, (u32)KH_CODE_WORDS);
        __syncthreads();
        extract_positions<W, PPT>(code, bad16, (u32)PPT * tid, k, roller, [&](int, const KmerKey<W>& can) {
            const u32 slot = kh_slot<W>(can, k, sg.nb_virtual) - sg.b_first;
            if (slot >= nb) return;   // a key-range wave keeps its own slice of the key space only
            if (SCATTER) {
                const u32 pos = atomicAdd(&cur[slot], 1u);
                part[part_base + pos] = can;
            } else {
                atomicAdd(&cur[slot], 1u);
            }
        });
    }
    if (!SCATTER) {
        __syncthreads();
        for (u32 i = tid; i < nb; i += 256) row[i] = cur[i];
    }
}

// ------------------------------------------------------------------------------------------
// pass B with write combining.  The direct form above stores every key as a lone 8W-byte
// write; the lines are evicted from L2 before their neighbours arrive and HBM sees 3.3x the
// bytes (measured: WRITE_SIZE 6.6 GB for 2.0 GB of keys).  Here a round's keys are first
// counting-sorted by bucket inside LDS (through registers, in place), so that consecutive lanes
// then store consecutive keys of one bucket run: the same bytes reach HBM in far fewer, fuller
// sectors.  One workgroup of 1024 threads per CU stages 128 KiB of keys per round: 16384
// one-word keys (a bucket's run in a flush is ~12 keys), 8192 two-word keys.
// ------------------------------------------------------------------------------------------
constexpr u32 KH_ST_THREADS = 1024;
template <int W> struct StageGeo {
    static constexpr int SUB = W == 1 ? 2 * KH_SUBTILE : KH_SUBTILE;     // keys staged per round
    static constexpr int ROUNDS = KH_TILE / SUB;                          // rounds per tile
    static constexpr int PPT = SUB / (int)KH_ST_THREADS;                  // start positions per thread (16 / 8)
    static constexpr int CODE_WORDS = (SUB + KH_HALO) / 16;
};
size_t kh_extract_staged_lds_bytes(int W, u32 nb_alloc) {
    const size_t sub = W == 1 ? StageGeo<1>::SUB : StageGeo<2>::SUB;
    const size_t cw = W == 1 ? StageGeo<1>::CODE_WORDS : StageGeo<2>::CODE_WORDS;
    return sub * 8 * W + (size_t)nb_alloc * 4 + ((size_t)nb_alloc + 4) * 4 + cw * 4 + cw * 2 + 8 + 64;
}

template <int W>
__global__ __launch_bounds__(KH_ST_THREADS, 1) void k_extract_staged(const u8* __restrict__ seq,
                                                          const KhSeg* __restrict__ segs,
                                                          const KhTile* __restrict__ tiles,
                                                          u32 nb_alloc, int k,
                                                          const u32* __restrict__ thist,
                                                          const u64* __restrict__ bstart,
                                                          KmerKey<W>* __restrict__ part, u32 tile_pos) {
    using G = StageGeo<W>;
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    KmerKey<W>* stage = reinterpret_cast<KmerKey<W>*>(lds_raw);                      // [SUB]
    u32* cur = reinterpret_cast<u32*>(lds_raw + (size_t)G::SUB * 8 * W);             // [nb_alloc] global cursors
    u32* sub = cur + nb_alloc;                                                       // [nb_alloc + 4] round counts
    u32* code = sub + nb_alloc + 4;
    u16* bad16 = reinterpret_cast<u16*>(code + G::CODE_WORDS);
    u32* wsum = reinterpret_cast<u32*>(bad16 + G::CODE_WORDS + 4);                   // [16] scan scratch

    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const KhTile t = tiles[blockIdx.x];
    const KhSeg sg = segs[t.seg];
    const u32 nb = sg.nbuckets;
    const u32* row = thist + sg.thist_base + (u64)t.tile_in_seg * nb;
    constexpr u32 NT = KH_ST_THREADS;
    constexpr int PPT = G::PPT;
    for (u32 i = tid; i < nb; i += NT) cur[i] = row[i];
    const u64 part_base = bstart[sg.bucket_base];
    const Roller<W> roller(k);
    const u64 tile_pos0 = (u64)t.tile_in_seg * tile_pos;
    const u32 per = (nb + NT - 1) / NT;   // scan entries per thread (host guarantees <= 4)
    const int rounds = (int)(tile_pos / (u32)G::SUB);

    static_assert(G::CODE_WORDS <= 2 * (int)NT, "two code words per thread");
    CodeFetch pre;
    fetch_codes<NT>(sg.seq, sg.len, tile_pos0, (u32)G::CODE_WORDS, pre);
    for (int sb = 0; sb < rounds; ++sb) {
        const u64 p0 = tile_pos0 + (u64)sb * G::SUB;
        if (p0 >= sg.npos) break;   // uniform over the block
        __syncthreads();            // previous round fully flushed, cursors advanced
        KH_STAMP(0, 0);
        store_codes<NT>(pre, code, bad16, (u32)G::CODE_WORDS);   // fetched while the previous round was placed and flushed
        for (u32 i = tid; i <= nb; i += NT) sub[i] = 0;
        __syncthreads();
        KH_STAMP(0, 1);
        // ---- A: extract; the thread's keys stay in registers until they are placed (phase C/D)
        u32 vm = 0;   // which of the PPT start positions gave a key
        KmerKey<W> key[PPT];
#pragma unroll
        for (int j = 0; j < PPT; ++j) key[j] = key_zero<W>();
        const u32 nbv = sg.nb_virtual, b_first = sg.b_first;
        extract_positions<W, PPT>(code, bad16, (u32)PPT * tid, k, roller, [&](int j, const KmerKey<W>& can) {
            const u32 slot = kh_slot<W>(can, k, nbv) - b_first;
            if (slot >= nb) return;   // a key-range wave keeps its own slice of the key space only
            key[j] = can;
            vm |= 1u << j;
            atomicAdd(&sub[slot], 1u);
        });
        __syncthreads();
        KH_STAMP(0, 2);
        // the next round's bases: requested now, decoded at the top of the next round (a bare
        // HBM round trip at the start of every round was a fifth of the round)
        if (sb + 1 < rounds && p0 + G::SUB < sg.npos) fetch_codes<NT>(sg.seq, sg.len, p0 + G::SUB, (u32)G::CODE_WORDS, pre);
        // ---- B: exclusive scan of the bucket counts, in place (sub[b] = first staged index)
        {
            u32 c[4], sum = 0;
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {
                const u32 b = tid * per + j;
                c[j] = (j < per && b < nb) ? sub[b] : 0u;
                sum += c[j];
            }
            const u32 incl = wave_scan_add(sum);
            if (lane == KH_WAVE - 1) wsum[wid] = incl;
            __syncthreads();
            u32 run = incl - sum;
            for (u32 w = 0; w < wid; ++w) run += wsum[w];
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {
                const u32 b = tid * per + j;
                if (j < per && b < nb) sub[b] = run;
                run += c[j];
            }
            if (tid == NT - 1) sub[nb] = run;   // total staged keys (threads past nb add nothing)
        }
        __syncthreads();
        KH_STAMP(0, 3);
        // ---- C/D: counting sort by bucket, registers -> staging array
        {
            u32 at[PPT];
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                at[j] = 0;
                if (vm & (1u << j)) at[j] = atomicAdd(&sub[kh_slot<W>(key[j], k, nbv) - b_first], 1u);
            }
#pragma unroll
            for (int j = 0; j < PPT; ++j)
                if (vm & (1u << j)) stage[at[j]] = key[j];
        }
        __syncthreads();
        KH_STAMP(0, 4);
        // ---- E: flush; sub[b] now holds the END of bucket b's staged run.  Two batches of PPT / 2:
        // all key reads, then all cursor reads, then the stores (two dependent LDS round trips per
        // batch instead of per key)
        {
            const u32 nvalid = sub[nb];
            constexpr int HB = PPT / 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                KmerKey<W> fk[HB];
                u32 fb2[HB], first[HB], cb[HB];
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const u32 p = NT * (u32)(h * HB + j) + tid;
                    fk[j] = stage[p < nvalid ? p : 0];
                }
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const u32 p = NT * (u32)(h * HB + j) + tid;
                    fb2[j] = p < nvalid ? kh_slot<W>(fk[j], k, nbv) - b_first : 0u;
                    first[j] = fb2[j] ? sub[fb2[j] - 1] : 0u;
                    cb[j] = cur[fb2[j]];
                }
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const u32 p = NT * (u32)(h * HB + j) + tid;
#ifndef KH_DIAG_NO_SCATTER_STORE   // diagnostic builds only: how long pass B takes without its stores
                    if (p < nvalid) part[part_base + cb[j] + (p - first[j])] = fk[j];
#else
                    if (p < nvalid && fk[j].lo == 0x123456789abcdefull && fb2[j] == 0xffffffffu) part[0] = fk[j];
#endif
                }
            }
        }
        __syncthreads();
        KH_STAMP(0, 5);
        KH_STAMP(0, 6);
        KH_STAMP(0, 7);
        for (u32 b = tid; b < nb; b += NT) cur[b] += sub[b] - (b ? sub[b - 1] : 0u);
        KH_STAMP(0, 8);
    }
}

// Column totals of the (tile x bucket) histogram matrix and, after the bucket starts are known,
// the per-tile write cursors.  Block = 64 buckets x KH_COL_TY tile groups: lanes run over buckets
// (coalesced rows), the tile groups split a segment's tiles into contiguous ranges, so that a
// single genome cut into a few hundred small tiles (the per-call path) is not one long serial
// walk per bucket.
constexpr u32 KH_COL_TY = 8;
__global__ __launch_bounds__(64 * KH_COL_TY) void k_col_totals(const KhSeg* __restrict__ segs,
                                                              const u32* __restrict__ thist, u64* __restrict__ tot) {
    __shared__ u64 part_sum[KH_COL_TY][64];
    const KhSeg sg = segs[blockIdx.y];
    const u32 b = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    const u32 t0 = (u32)((u64)sg.ntiles * ty / KH_COL_TY), t1 = (u32)((u64)sg.ntiles * (ty + 1) / KH_COL_TY);
    u64 s = 0;
    if (b < sg.nbuckets)
        for (u32 t = t0; t < t1; ++t) s += thist[sg.thist_base + (u64)t * sg.nbuckets + b];
    part_sum[ty][threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && b < sg.nbuckets) {
        u64 a = 0;
        for (u32 y = 0; y < KH_COL_TY; ++y) a += part_sum[y][threadIdx.x];
        tot[sg.bucket_base + b] = a;
    }
}
// turn per-tile counts into per-tile write cursors, relative to the segment's first bucket
__global__ __launch_bounds__(64 * KH_COL_TY) void k_col_offsets(const KhSeg* __restrict__ segs, u32* __restrict__ thist,
                                                               const u64* __restrict__ bstart, const u32* __restrict__ rank,
                                                               const u64* __restrict__ seg_out_base,
                                                               KhBucketWork* __restrict__ work, u32* __restrict__ over,
                                                               u32 over_cap) {
    __shared__ u64 part_sum[KH_COL_TY][64];
    const KhSeg sg = segs[blockIdx.y];
    const u32 b = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    const bool live = b < sg.nbuckets;
    const u32 t0 = (u32)((u64)sg.ntiles * ty / KH_COL_TY), t1 = (u32)((u64)sg.ntiles * (ty + 1) / KH_COL_TY);
    u64 s = 0;
    if (live)
        for (u32 t = t0; t < t1; ++t) s += thist[sg.thist_base + (u64)t * sg.nbuckets + b];
    part_sum[ty][threadIdx.x] = s;
    __syncthreads();
    if (!live) return;
    const u32 gb = sg.bucket_base + b;
    const u64 lo = bstart[gb];
    // pass C's work item of this bucket, at its place in the interleaved start order
    if (ty == 0) {
        const u64 nkeys = bstart[gb + 1] - lo;
        work[rank[gb]] = KhBucketWork{lo, seg_out_base[blockIdx.y], (u32)nkeys, sg.nb_virtual, b, gb};
        // grid mode: buckets above the LDS capacity go on a list for the folding kernel (over[0] = count)
        if (over && nkeys > over_cap) over[1u + atomicAdd(&over[0], 1u)] = rank[gb];
    }
    u64 running = lo - bstart[sg.bucket_base];
    for (u32 y = 0; y < ty; ++y) running += part_sum[y][threadIdx.x];
    for (u32 t = t0; t < t1; ++t) {
        const u64 at = sg.thist_base + (u64)t * sg.nbuckets + b;
        const u32 c = thist[at];
        thist[at] = (u32)running;
        running += c;
    }
}

// Exclusive scan of n u64 in two launches of n/4096 workgroups (one workgroup walking the
// whole array took 100 us for the 67 K buckets of the benchmark batch): PHASE 0 writes each
// tile's sum, PHASE 1 adds up the sums of the tiles before its own (redundantly, from L2) and
// scans its tile.  out[n] = grand total.
constexpr u32 KH_SCAN_TILE = 4096;
template <int PHASE>
__global__ __launch_bounds__(1024) void k_exscan(const u64* __restrict__ in, u64* __restrict__ out,
                                                 u64 n, u64* __restrict__ tile_sum) {
    __shared__ u64 wsum[16];
    __shared__ u64 tile_base;
    const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const u64 i0 = (u64)blockIdx.x * KH_SCAN_TILE + 4ull * tid;
    u64 v[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = i0 + j < n ? in[i0 + j] : 0ull;
        sum += v[j];
    }
    u64 incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 u = __shfl_up(incl, off);
        if (lane >= (u32)off) incl += u;
    }
    if (lane == 63) wsum[wid] = incl;
    if (PHASE == 1) {
        u64 part = 0;
        for (u32 t = tid; t < blockIdx.x; t += 1024) part += tile_sum[t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (tid == 0) tile_base = 0;
        __syncthreads();
        if (lane == 0 && part) atomicAdd(reinterpret_cast<unsigned long long*>(&tile_base), (unsigned long long)part);
    }
    __syncthreads();
    u64 before = 0, total = 0;
#pragma unroll
    for (u32 w = 0; w < 16; ++w) {
        const u64 x = wsum[w];
        before += w < wid ? x : 0ull;
        total += x;
    }
    if (PHASE == 0) {
        if (tid == 0) tile_sum[blockIdx.x] = total;
        return;
    }
    u64 run = tile_base + before + incl - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j < n) out[i0 + j] = run;
        run += v[j];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 1023) out[n] = run;
}

// ------------------------------------------------------------------------------------------
// LDS bitonic sort (keys, optional 32-bit payload) of s[0..n), any n.
// All comparators put the smaller key at the lower index (the "flip" form of the network:
// the first step of every merge stage compares mirrored elements).  Indices >= n stand for
// +infinity keys; such an element can never move, so a comparator touching one is skipped
// and no padding (or sentinel key value) is needed.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 next_pow2(u32 v) {
    u32 p = 2;
    while (p < v) p <<= 1;
    return p;
}

template <int W, bool PAY>
__device__ __forceinline__ void cmp_exchange(KmerKey<W>* s, u32* pay, u32 i, u32 j) {
    const KmerKey<W> a = s[i], b = s[j];
    if (key_lt(b, a)) {
        s[i] = b;
        s[j] = a;
        if (PAY) {
            const u32 pa = pay[i];
            pay[i] = pay[j];
            pay[j] = pa;
        }
    }
}

template <int W, bool PAY>
__device__ void bitonic_sort_lds(KmerKey<W>* s, u32* pay, const u32 n) {
    const u32 tid = threadIdx.x, nt = blockDim.x;
    if (n < 2) return;
    const u32 P = next_pow2(n);
    for (u32 size = 2; size <= P; size <<= 1) {
        const u32 half = size >> 1;
        for (u32 t = tid; t < (P >> 1); t += nt) {
            const u32 off = t & (half - 1);
            const u32 blk = (t & ~(half - 1)) << 1;
            const u32 i = blk | off, j = blk + size - 1 - off;
            if (j < n) cmp_exchange<W, PAY>(s, pay, i, j);
        }
        __syncthreads();
        for (u32 stride = size >> 2; stride > 0; stride >>= 1) {
            for (u32 t = tid; t < (P >> 1); t += nt) {
                const u32 i = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                const u32 j = i | stride;
                if (j < n) cmp_exchange<W, PAY>(s, pay, i, j);
            }
            __syncthreads();
        }
    }
}

// Mark run heads of the sorted keys s[0..n) and record their start indices:
// hstart[r] = index of the r-th distinct key, hstart[d] = n.  Returns d (block-uniform).
template <int W>
__device__ u32 find_runs(const KmerKey<W>* s, u32 n, u16* hstart, u32* wave_tot) {
    const u32 tid = threadIdx.x, nt = blockDim.x;
    u32 base = 0;
    for (u32 i0 = 0; i0 < n; i0 += nt) {
        const u32 i = i0 + tid;
        const bool head = (i < n) && (i == 0 || !key_eq(s[i], s[i - 1]));
        u32 tot;
        const u32 rk = block_rank(head, wave_tot, tot);
        if (head) hstart[base + rk] = (u16)i;
        base += tot;
    }
    if (tid == 0) hstart[base] = (u16)n;
    __syncthreads();
    return base;
}

// LDS carve shared by k_bucket_sort_rle and k_setop (everything lives in dynamic LDS so that
// its base stays 16-B aligned):  keys[cap] | pay[cap] (optional) | aux | lhist[KH_LHIST_BINS] |
// tab[128 + KH_FINE_BINS/32 + KH_WORKLIST] (ballot table, dirty-bin bitmap, repair work list) |
// scratch[32] u32 | bcast[4] u64, where aux holds the fine-bin table
// bins[KH_FINE_BINS+1] u32 while sorting and hstart[cap+2] u16 afterwards.
struct SortLds {
    u8* base;
    u32 cap;
    int W;
    bool pay;
    __host__ __device__ size_t keys_off() const { return 0; }
    __host__ __device__ size_t pay_off() const { return (size_t)cap * 8 * W; }
    __host__ __device__ size_t hstart_off() const { return pay_off() + (pay ? (size_t)cap * 4 : 0); }
    __host__ __device__ size_t lhist_off() const {
        size_t aux = ((size_t)cap + 2) * 2;
        if (aux < (size_t)(KH_FINE_BINS / 2 + 1) * 4) aux = (size_t)(KH_FINE_BINS / 2 + 1) * 4;
        return (hstart_off() + aux + 15) & ~(size_t)15;
    }
    __host__ __device__ size_t tab_off() const { return lhist_off() + KH_LHIST_BINS * 4; }
    __host__ __device__ size_t scratch_off() const {
        return tab_off() + (128 + KH_FINE_BINS / 32 + KH_WORKLIST) * 4;
    }
    __host__ __device__ size_t bcast_off() const { return scratch_off() + 32 * 4; }
    __host__ __device__ size_t total() const { return bcast_off() + 4 * 8; }
};
size_t kh_sort_lds_bytes(int W, u32 cap, bool pay) {
    SortLds L{nullptr, cap, W, pay};
    return L.total();
}

// ------------------------------------------------------------------------------------------
// Block-wide decoupled look-back: all KH_SORT_THREADS threads inspect one predecessor each, so
// the usual case costs one global round trip instead of a chain of 64-wide windows.
// scratch: >= 24 u32 words (8-byte aligned at word 8).  Returns the exclusive prefix to every
// thread.  Contains barriers: every thread of the block must call it.
// ------------------------------------------------------------------------------------------
__device__ u64 lookback_block(u64* desc, const u32 q, const u64 mine, u32* err, u32* scratch) {
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6, nw = blockDim.x >> 6;
    u32* first_in_wave = scratch;                               // [nw]
    u64* wave_sum = reinterpret_cast<u64*>(scratch + 8);         // [nw]
    if (q == 0) {
        if (tid == 0) lb_store(&desc[0], KH_LB_PREFIX | mine);
        return 0;
    }
    if (tid == 0) lb_store(&desc[q], KH_LB_AGG | mine);
    u64 excl = 0;
    long long base = (long long)q - 1;
    bool timed_out = false;
    while (true) {
        const long long idx = base - (long long)tid;
        u64 dsc = KH_LB_PREFIX;   // "before the first part": prefix 0
        if (idx >= 0) {
            dsc = lb_load(&desc[idx]);
            u32 spins = 0;
            while ((dsc >> 62) == 0) {
                __builtin_amdgcn_s_sleep(2);
                dsc = lb_load(&desc[idx]);
                if (++spins > (1u << 25)) { timed_out = true; dsc = KH_LB_PREFIX; break; }
            }
        }
        const u64 pm = __ballot((dsc >> 62) == 2);
        if (lane == 0) first_in_wave[wid] = pm ? (u32)__ffsll((unsigned long long)pm) - 1u : 64u;
        __syncthreads();
        u32 tp = 0xffffffffu;   // nearest predecessor that already knows its inclusive prefix
        for (u32 w = 0; w < nw; ++w) {
            const u32 f = first_in_wave[w];
            if (f < 64u) { tp = w * KH_WAVE + f; break; }
        }
        // aggregates in front of the nearest prefix are per-part counts (far below 2^32 for a
        // whole wave): summed on the DPP path; the prefix itself, one 64-bit value, is added by
        // the thread that holds it
        const u32 small = wave_scan_add((tid < tp) ? (u32)(dsc & KH_LB_VALUE) : 0u);
        if (lane == KH_WAVE - 1) wave_sum[wid] = small;
        if (tid == tp) wave_sum[nw] = dsc & KH_LB_VALUE;
        __syncthreads();
        for (u32 w = 0; w < nw; ++w) excl += wave_sum[w];
        if (tp != 0xffffffffu) excl += wave_sum[nw];
        if (tp != 0xffffffffu) break;
        base -= (long long)blockDim.x;
        __syncthreads();
    }
    if (__ballot(timed_out) && lane == 0) atomicOr(err, KH_ERR_SPIN_TIMEOUT);
    if (tid == 0) lb_store(&desc[q], KH_LB_PREFIX | ((excl + mine) & KH_LB_VALUE));
    return excl;
}

// ------------------------------------------------------------------------------------------
// Distribution sort of one slot's keys, from registers into LDS.
// Mixed keys are uniform inside a slot, so one counting pass over KH_FINE_BINS order-preserving
// fine bins (about 0.4 keys per bin) leaves the array sorted up to tiny per-bin permutations.
// The first key to arrive in a bin is its leader; a leader whose bin holds more than one key
// tidies the bin with a serial insertion pass, which is linear when the bin holds copies of
// one key (set unions, repeats) and touches 2-3 keys otherwise.  Everything else is
// key-parallel; about 20x less LDS traffic than the bitonic network, to which a block whose
// fullest bin exceeds KH_FINE_LIMIT falls back (always correct).
//   kreg/preg : this thread's keys (element e*NT + tid) and payloads
//   bins      : LDS u32[KH_FINE_BINS / 2 + 1], two 16-bit bins per word (counts, then bases);
//               aliases hstart, which is only used afterwards
//   dirty     : LDS bitmap u32[KH_FINE_BINS / 32] of bins that arrived out of order
//   wl        : LDS u32[KH_WORKLIST] work list of the keys of those bins
// ------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ u32 fine_bin(const KmerKey<W>& key, int k, u32 nslots) {
    const u32 frac = (u32)((u64)kh_top32(key, k) * (u64)nslots);   // position inside the slot
    return frac >> (32 - KH_FINE_BITS);
}
__device__ __forceinline__ u32 bin_base(const u32* bins, u32 f) {
    return (bins[f >> 1] >> (16 * (f & 1))) & 0xffffu;
}

__device__ __forceinline__ void distribute_clear(u32* bins, u32* dirty, u32* scratch) {
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i <= (u32)KH_FINE_BINS / 2; i += KH_SORT_THREADS) bins[i] = 0;
    for (u32 i = tid; i < (u32)KH_FINE_BINS / 32; i += KH_SORT_THREADS) dirty[i] = 0;
    if (tid == 0) scratch[20] = 0;   // work-list length
}

// Phases 1-3 of the distribution sort: count the keys into the fine bins (the returning LDS add
// hands every key its arrival rank inside its bin), scan the bins, scatter the keys.  On return
// (a barrier has been passed) s[] holds the keys grouped by fine bin, in arrival order inside a
// bin; fr[e] = bin << 16 | arrival rank, at[e] = position of element e, bmax = fullest bin.
// `binfn(key)` must be monotone in the key over the slot and < KH_FINE_BINS.
template <int W, bool PAY, int E, class BF>
__device__ __forceinline__ void distribute_place(const KmerKey<W> (&kreg)[E], const u32 (&preg)[E], const u32 n,
                                                 KmerKey<W>* s, u32* pay, u32* bins, u32* scratch, BF binfn, u32 q,
                                                 u32 (&fr)[E], u32 (&at)[E], u32& vmask, u32& bmax) {
    // precondition: distribute_clear() ran and a barrier followed (the kernels fold it into the
    // barrier that broadcasts the ticket)
    constexpr u32 NT = KH_SORT_THREADS;
    constexpr u32 WORDS = KH_FINE_BINS / 2;
    constexpr u32 PER = WORDS / NT;   // packed words scanned per thread
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    // Every phase below is written as "issue all LDS operations of the thread, then consume":
    // used inside one predicated block, a result makes hipcc wait (lgkmcnt(0)) before the next
    // block, which chains the E LDS latencies of a thread instead of overlapping them.
    vmask = 0;   // bit e: element e*NT + tid exists
    {
        u32 old[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            fr[e] = binfn(kreg[e]);
            old[e] = 0;
            if ((u32)e * NT + tid < n) {
                vmask |= 1u << e;
                old[e] = atomicAdd(&bins[fr[e] >> 1], 1u << (16 * (fr[e] & 1)));
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) fr[e] = (fr[e] << 16) | ((old[e] >> (16 * (fr[e] & 1))) & 0xffffu);
    }
    __syncthreads();
    KH_STAMP(q, 2);
    // exclusive scan of the bin counts (PER consecutive words per thread) + fullest bin
    u32 c[PER], sum = 0, mx = 0;
#pragma unroll
    for (u32 j = 0; j < PER; ++j) {
        c[j] = bins[tid * PER + j];
        const u32 c0 = c[j] & 0xffffu, c1 = c[j] >> 16;
        mx = c0 > mx ? c0 : mx;
        mx = c1 > mx ? c1 : mx;
        sum += c0 + c1;
    }
    const u32 incl = wave_scan_add(sum);
    mx = wave_scan_max(mx);                      // the wave's maximum arrives in lane 63
    if (lane == KH_WAVE - 1) { scratch[wid] = incl; scratch[8 + wid] = mx; }
    __syncthreads();
    u32 wbase = 0;
    bmax = 0;
#pragma unroll
    for (u32 w = 0; w < NT / KH_WAVE; ++w) {
        wbase += (w < wid) ? scratch[w] : 0u;
        bmax = scratch[8 + w] > bmax ? scratch[8 + w] : bmax;
    }
    u32 run = wbase + incl - sum;
#pragma unroll
    for (u32 j = 0; j < PER; ++j) {
        const u32 c0 = c[j] & 0xffffu, c1 = c[j] >> 16;
        bins[tid * PER + j] = run | ((run + c0) << 16);
        run += c0 + c1;
    }
    if (tid == NT - 1) bins[WORDS] = run;   // base of the bin past the last one = n
    __syncthreads();
    KH_STAMP(q, 3);
#pragma unroll
    for (int e = 0; e < E; ++e) at[e] = bin_base(bins, fr[e] >> 16) + (fr[e] & 0xffffu);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (vmask & (1u << e)) {
            s[at[e]] = kreg[e];
            if (PAY) pay[at[e]] = preg[e];
        }
    }
    __syncthreads();
    KH_STAMP(q, 4);
}

template <int W, bool PAY, int E, class BF>
__device__ void distribute_sort_bf(const KmerKey<W> (&kreg)[E], const u32 (&preg)[E], const u32 n,
                                   KmerKey<W>* s, u32* pay, u32* bins, u32* dirty, u32* wl,
                                   u32* scratch, BF binfn, u32 q) {
    constexpr u32 NT = KH_SORT_THREADS;
    const u32 tid = threadIdx.x, lane = lane_id();
    u32 fr[E], at[E], vmask, bmax;
    distribute_place<W, PAY, E>(kreg, preg, n, s, pay, bins, scratch, binfn, q, fr, at, vmask, bmax);
    if (bmax > (u32)KH_FINE_LIMIT) {
        bitonic_sort_lds<W, PAY>(s, pay, n);
        KH_STAMP(q, 5);
        return;
    }
    if (bmax > 1) {
        // (a) a bin is out of order iff some key is smaller than its predecessor in the bin
        {
            KmerKey<W> pred[E];
#pragma unroll
            for (int e = 0; e < E; ++e) pred[e] = s[at[e] ? at[e] - 1 : 0];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if ((vmask & (1u << e)) && (fr[e] & 0xffffu) != 0 && key_lt(kreg[e], pred[e])) {
                    const u32 f = fr[e] >> 16;
                    atomicOr(&dirty[f >> 5], 1u << (f & 31));
                }
            }
        }
        __syncthreads();
        // (b) keys of out-of-order bins (a few per cent of all keys) enter a work list, so that
        // the repair below runs with dense lanes instead of diverging over all keys
        // (one list-length atomic per wave: the wave's entries are ranked by ballots)
        {
            u32 dw[E];
#pragma unroll
            for (int e = 0; e < E; ++e) dw[e] = dirty[fr[e] >> 21];
            const u64 lt_mask = (1ull << lane) - 1ull;
            u32 listed = 0, rank[E], mine = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const u32 f = fr[e] >> 16;
                const bool on = (vmask & (1u << e)) && ((dw[e] >> (f & 31)) & 1u);
                const u64 bal = __ballot(on);
                rank[e] = listed + (u32)__popcll(bal & lt_mask);
                listed += (u32)__popcll(bal);
                mine |= on ? (1u << e) : 0u;
            }
            u32 base = 0;
            if (listed) {
                if (lane == 0) base = atomicAdd(&scratch[20], listed);
                base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            }
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((mine & (1u << e)) && base + rank[e] < (u32)KH_WORKLIST)
                    wl[base + rank[e]] = at[e] | ((fr[e] >> 16) << 12);
        }
        __syncthreads();
        const u32 wlc = scratch[20];
        if (wlc > (u32)KH_WORKLIST) {   // too disordered for the list: sort everything
            bitonic_sort_lds<W, PAY>(s, pay, n);
            KH_STAMP(q, 5);
            return;
        }
        // (c) final place of a listed key: smaller keys of its bin + equal keys before it
        constexpr int IT = KH_WORKLIST / NT;
        KmerKey<W> wk[IT];
        u32 wp[IT], wdst[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            wdst[it] = 0xffffffffu;
            wp[it] = 0;
            wk[it] = key_zero<W>();
            const u32 w = (u32)it * NT + tid;
            if (w < wlc) {
                const u32 item = wl[w];
                const u32 at = item & 0xfffu, f = item >> 12;
                const u32 b0 = bin_base(bins, f), cnt = bin_base(bins, f + 1) - b0, r = at - b0;
                wk[it] = s[at];
                if (PAY) wp[it] = pay[at];
                u32 pos = 0;
                for (u32 t = 0; t < cnt; ++t) {
                    const KmerKey<W> o = s[b0 + t];
                    pos += (key_lt(o, wk[it]) || (t < r && key_eq(o, wk[it]))) ? 1u : 0u;
                }
                wdst[it] = b0 + pos;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (wdst[it] != 0xffffffffu) {
                s[wdst[it]] = wk[it];
                if (PAY) pay[wdst[it]] = wp[it];
            }
        }
        __syncthreads();
    }
    KH_STAMP(q, 5);
}

template <int W, bool PAY, int E>
__device__ __forceinline__ void distribute_sort(const KmerKey<W> (&kreg)[E], const u32 (&preg)[E], const u32 n,
                                                KmerKey<W>* s, u32* pay, u32* bins, u32* dirty, u32* wl,
                                                u32* scratch, int k, u32 nslots, u32 q) {
    distribute_sort_bf<W, PAY, E>(kreg, preg, n, s, pay, bins, dirty, wl, scratch,
                                  [=](const KmerKey<W>& key) { return fine_bin<W>(key, k, nslots); }, q);
}

// ------------------------------------------------------------------------------------------
// Run-length pass + ordered output over sorted s[0..n), n <= E * KH_SORT_THREADS.
// Key-parallel throughout: (1) run heads by comparing neighbours, ranked with one table of
// per-wave ballots (a single barrier), hstart[r] = start of run r; (2) eval(h0, h1) gives the
// output counter of run r (0 = drop the key), kept runs are ranked the same way; (3) the slot's
// kept count goes through the block-wide look-back; (4) consecutive lanes write consecutive
// outputs (coalesced).  All threads of the block must call it.
// ------------------------------------------------------------------------------------------
// tab[e * NW + w] holds the number of flagged lanes of wave w in pass e (E * NW <= 64 entries,
// NW = waves per workgroup).
// One wave turns the table into its exclusive prefix (pass-major order) and stores the grand
// total behind it; two barriers inside.
template <int E>
__device__ __forceinline__ void table_scan(u32* tab) {
    static_assert(E * KH_SORT_NW <= KH_WAVE, "one wave scans the ballot table");
    __syncthreads();
    if (threadIdx.x < KH_WAVE) {
        const u32 lane = threadIdx.x;
        const u32 v = lane < (u32)E * KH_SORT_NW ? tab[lane] : 0u;
        const u32 incl = wave_scan_add(v);
        if (lane < (u32)E * KH_SORT_NW) tab[lane] = incl - v;
        if (lane == KH_WAVE - 1) tab[E * KH_SORT_NW] = incl;
    }
    __syncthreads();
}

template <int W, int E, class Eval, class Sink>
__device__ u64 rle_emit(const KmerKey<W>* s, const u32 n, u16* hstart, u32* tab, Eval eval, Sink sink,
                         KhLookback lb, const u32 q, u32* scratch, const bool all_kept,
                         const bool keys_only = false) {
    constexpr u32 NT = KH_SORT_THREADS;
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const u64 lt_mask = (1ull << lane) - 1ull;
    u32 lr[E];
    u32 flags = 0;
    // ---- (1) run heads (all LDS reads of the thread are issued before the first is used)
    KmerKey<W> cur[E];
    {
        KmerKey<W> prv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u32 i = (u32)e * NT + tid;
            cur[e] = s[i < n ? i : 0];
            prv[e] = s[(i < n && i) ? i - 1 : 0];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u32 i = (u32)e * NT + tid;
            const bool head = (i < n) && (i == 0 || !key_eq(cur[e], prv[e]));
            const u64 bal = __ballot(head);
            if (lane == 0) tab[e * KH_SORT_NW + wid] = (u32)__popcll(bal);
            lr[e] = (u32)__popcll(bal & lt_mask);
            flags |= head ? (1u << e) : 0u;
        }
    }
    table_scan<E>(tab);
    const u32 d = tab[E * KH_SORT_NW];
    // when no run can be dropped the slot's output count is already known: publish it now so
    // that it is visible to the successors by the time they look back
    if (all_kept && q != 0 && tid == 0) lb_store(&lb.desc[q], KH_LB_AGG | (u64)d);
    if (all_kept && keys_only) {
        // plain sets (no counter array, nothing dropped): every run head is an output and its
        // rank is already known, so the run table and the second ranking are skipped
        u32 base[E];
#pragma unroll
        for (int e = 0; e < E; ++e) base[e] = tab[e * KH_SORT_NW + wid] + lr[e];
        KH_STAMP(q, 6);
        const u64 ob = lookback_block(lb.desc, q, (u64)d, lb.err, scratch);
        KH_STAMP(q, 7);
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (flags & (1u << e)) sink(ob + base[e], cur[e], 1u);
        KH_STAMP(q, 8);
        return ob + d;
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (flags & (1u << e)) hstart[tab[e * KH_SORT_NW + wid] + lr[e]] = (u16)((u32)e * NT + tid);
    if (tid == 0) hstart[d] = (u16)n;
    __syncthreads();
    KH_STAMP(q, 6);
    if (all_kept) {
        // every run is kept: output rank = run index, only the counters remain to be evaluated
        u32 h0[E], h1[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u32 r = (u32)e * NT + tid;
            h0[e] = hstart[r < d ? r : 0];
            h1[e] = hstart[r < d ? r + 1 : 0];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = s[h0[e]];
        const u64 ob = lookback_block(lb.desc, q, (u64)d, lb.err, scratch);
        KH_STAMP(q, 7);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u32 r = (u32)e * NT + tid;
            if (r < d) sink(ob + r, cur[e], eval(h0[e], h1[e]));
        }
        KH_STAMP(q, 8);
        return ob + d;
    }
    // ---- (2) counters of the runs, kept runs ranked
    u32 cnt[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u32 r = (u32)e * NT + tid;
        cnt[e] = (r < d) ? eval((u32)hstart[r], (u32)hstart[r + 1]) : 0u;
        const u64 bal = __ballot(cnt[e] != 0);
        if (lane == 0) tab[e * KH_SORT_NW + wid] = (u32)__popcll(bal);
        lr[e] = (u32)__popcll(bal & lt_mask);
    }
    table_scan<E>(tab);
    const u32 kept = tab[E * KH_SORT_NW];
#pragma unroll
    for (int e = 0; e < E; ++e) lr[e] += tab[e * KH_SORT_NW + wid];
    // ---- (3) slot prefix
    const u64 obase = lookback_block(lb.desc, q, (u64)kept, lb.err, scratch);
    KH_STAMP(q, 7);
    // ---- (4) output
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (cnt[e]) {
            const u32 r = (u32)e * NT + tid;
            sink(obase + lr[e], s[hstart[r]], cnt[e]);
        }
    }
    KH_STAMP(q, 8);
    return obase + kept;   // all outputs of the chain up to and including this part
}

// ------------------------------------------------------------------------------------------
// Grid mode of pass C (fused experiment-type-1 path, KhGrid in kh_launch.h): the distinct keys
// of the sorted bucket s[0..n) go to out[0..d) — the bucket's own place, known before the launch,
// so nothing is waited for — together with the bucket's sub-range index off[0..S].
// ------------------------------------------------------------------------------------------
// Sub-range of a key inside its bucket.  Sub-ranges are made of whole fine bins of the bucket
// (sub-range f = fine bins [kh_first_bin(f), kh_first_bin(f + 1))), so that keys grouped by fine bin
// are grouped by sub-range too.
template <int W>
__device__ __forceinline__ u32 kh_sub(const KmerKey<W>& key, int k, u32 nb, u32 S) {
    return (fine_bin<W>(key, k, nb) * S) >> KH_FINE_BITS;
}
__host__ __device__ __forceinline__ u32 kh_first_bin(u32 f, u32 S) { return (f * (u32)KH_FINE_BINS + S - 1u) / S; }

// In-bin search for the first copy of each of this thread's keys (its "leader") after
// distribute_place: keys sharing a fine bin sit next to each other in arrival order, equal keys
// always share a bin, and bins hold a key or two, so a key finds an earlier copy of itself — or
// learns that it is the first — with one LDS read in nearly every case.  lead[e] = position of the
// first copy (== at[e] for a leader).  No repair of the in-bin order is needed by callers that
// only want equal keys to meet.
template <int W, int E>
__device__ __forceinline__ void find_leaders(const KmerKey<W> (&kreg)[E], const KmerKey<W>* s, const u32 (&fr)[E],
                                             const u32 (&at)[E], const u32 vmask, u32 (&lead)[E]) {
    KmerKey<W> first[E];
#pragma unroll
    for (int e = 0; e < E; ++e) first[e] = s[at[e] - (fr[e] & 0xffffu)];   // the bin's first arrival (itself for rank 0)
    u32 pos[E], act = 0;   // act bit e: element e still looks for an earlier copy, next candidate pos[e]
#pragma unroll
    for (int e = 0; e < E; ++e) {
        lead[e] = at[e];
        const u32 rk = fr[e] & 0xffffu;
        pos[e] = at[e] - rk + 1u;
        if ((vmask & (1u << e)) && rk) {
            if (key_eq(first[e], kreg[e])) lead[e] = at[e] - rk;
            else if (rk >= 2u) act |= 1u << e;
        }
    }
    KH_STAMP(0, 9);
    // The rare keys that are neither first in their bin nor copies of its first key probe on, in
    // ROUNDS: every round issues one candidate read for each of the thread's searching elements
    // together (one LDS round trip per round, not one per element and step).
    while (__builtin_amdgcn_ballot_w64(act != 0)) {
        KmerKey<W> cand[E];
#pragma unroll
        for (int e = 0; e < E; ++e) cand[e] = s[(act & (1u << e)) ? pos[e] : 0u];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (act & (1u << e)) {
                if (key_eq(cand[e], kreg[e])) { lead[e] = pos[e]; act &= ~(1u << e); }
                else if (++pos[e] >= at[e]) act &= ~(1u << e);
            }
        }
    }
    KH_STAMP(0, 10);
}

// Grid mode of pass C for a bucket that fits LDS: place the keys by fine bin, drop every key that
// has an earlier copy, write the others in position (= fine bin) order where the bucket's input
// starts, and derive the sub-range index from the bin table.  No in-bin repair: the output is
// grouped by sub-range, not sorted inside one (its only reader, the tagged union, re-bins it).
// lbits: 64 u64 words, zeroed (the dirty-bin bitmap of the full sort, cleared by distribute_clear).
template <int W, int E>
__device__ void grid_bucket(const KmerKey<W> (&kreg)[E], const u32 n, KmerKey<W>* s, u32* bins, u32* tab,
                            u32* scratch, u32* lbits, int k, u32 nb, KmerKey<W>* __restrict__ out,
                            u16* __restrict__ off, const u32 S, unsigned long long* __restrict__ distinct, u32 q) {
    constexpr u32 NT = KH_SORT_THREADS;
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    u32 fr[E], at[E], vmask, bmax;
    {
        u32 none[E];
#pragma unroll
        for (int e = 0; e < E; ++e) none[e] = 0;
        distribute_place<W, false, E>(kreg, none, n, s, nullptr, bins, scratch,
                                      [=](const KmerKey<W>& key) { return fine_bin<W>(key, k, nb); }, q, fr, at,
                                      vmask, bmax);
    }
    {
        u32 lead[E];
        find_leaders<W, E>(kreg, s, fr, at, vmask, lead);
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((vmask & (1u << e)) && lead[e] == at[e]) atomicOr(&lbits[at[e] >> 5], 1u << (at[e] & 31u));
    }
    KH_STAMP(q, 11);
    __syncthreads();
    KH_STAMP(q, 5);
    // position-major from here on: element p = e * NT + tid, consecutive lanes = consecutive outputs
    const unsigned long long* lb64 = reinterpret_cast<const unsigned long long*>(lbits);
    u32 lr[E], flags = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u64 word = lb64[e * KH_SORT_NW + wid];          // leaders among this wave's 64 positions
        if (lane == 0) tab[e * KH_SORT_NW + wid] = (u32)__popcll(word);
        lr[e] = (u32)__popcll(word & ((1ull << lane) - 1ull));
        flags |= ((word >> lane) & 1ull) ? (1u << e) : 0u;
    }
    table_scan<E>(tab);
    KH_STAMP(q, 6);
    KH_STAMP(q, 7);
    const u32 d = tab[E * KH_SORT_NW];
    {
        KmerKey<W> cur[E];
#pragma unroll
        for (int e = 0; e < E; ++e) cur[e] = s[(flags & (1u << e)) ? (u32)e * NT + tid : 0u];
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (flags & (1u << e)) out[tab[e * KH_SORT_NW + wid] + lr[e]] = cur[e];
    }
    // off[f] = leaders in front of the first fine bin of sub-range f
    for (u32 t = tid; t <= S; t += NT) {
        u32 v = d;
        if (t < S) {
            const u32 P = bin_base(bins, kh_first_bin(t, S));
            if (P < n) v = tab[P >> 6] + (u32)__popcll(lb64[P >> 6] & ((1ull << (P & 63u)) - 1ull));
        }
        off[t] = (u16)v;
    }
    if (tid == 0 && d) atomicAdd(distinct, (unsigned long long)d);
    KH_STAMP(q, 8);
}

template <int W, int E>
__device__ void grid_emit(const KmerKey<W>* s, const u32 n, u32* tab, KmerKey<W>* __restrict__ out,
                          u16* __restrict__ off, const u32 S, int k, u32 nb,
                          unsigned long long* __restrict__ distinct) {
    constexpr u32 NT = KH_SORT_THREADS;
    const u32 tid = threadIdx.x, lane = lane_id(), wid = tid >> 6;
    const u64 lt_mask = (1ull << lane) - 1ull;
    KmerKey<W> cur[E], prv[E];
    u32 lr[E], flags = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u32 i = (u32)e * NT + tid;
        cur[e] = s[i < n ? i : 0];
        prv[e] = s[(i < n && i) ? i - 1 : 0];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u32 i = (u32)e * NT + tid;
        const bool head = (i < n) && (i == 0 || !key_eq(cur[e], prv[e]));
        const u64 bal = __ballot(head);
        if (lane == 0) tab[e * KH_SORT_NW + wid] = (u32)__popcll(bal);
        lr[e] = (u32)__popcll(bal & lt_mask);
        flags |= head ? (1u << e) : 0u;
    }
    table_scan<E>(tab);
    KH_STAMP(0, 6);
    KH_STAMP(0, 7);
    const u32 d = tab[E * KH_SORT_NW];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (flags & (1u << e)) {
            const u32 i = (u32)e * NT + tid;
            const u32 rank = tab[e * KH_SORT_NW + wid] + lr[e];
            out[rank] = cur[e];
            // this key opens every sub-range after its predecessor's up to its own
            const u32 sc = kh_sub<W>(cur[e], k, nb, S);
            for (u32 f = i ? kh_sub<W>(prv[e], k, nb, S) + 1u : 0u; f <= sc; ++f) off[f] = (u16)rank;
        }
    }
    // sub-ranges behind the last key (and the end marker off[S]) start at d
    const u32 first_open = n ? kh_sub<W>(s[n - 1], k, nb, S) + 1u : 0u;
    for (u32 t = first_open + tid; t <= S; t += NT) off[t] = (u16)d;
    if (tid == 0 && d) atomicAdd(distinct, (unsigned long long)d);
    KH_STAMP(0, 8);
}

// ------------------------------------------------------------------------------------------
// pass C: per-bucket sort + run-length count + ordered output
// ------------------------------------------------------------------------------------------
// Normal buckets (n <= cap): keys-only distribution sort, counter = run length.
// Oversize buckets (duplicate-heavy input, e.g. small k or low-complexity sequence): the
// same LDS is re-carved as (key, counter) pairs and the bucket is folded in chunks:
// [accumulated distinct pairs | next raw chunk] -> sort -> sum per key.
// Only a bucket with more DISTINCT keys than the pair capacity cannot be handled; that
// raises KH_ERR_CAPACITY (the host then re-plans the batch with more buckets).
template <int W>
__global__ __launch_bounds__(KH_SORT_THREADS, KH_SORT_WAVES_PER_SIMD) void k_bucket_sort_rle(
    const KmerKey<W>* __restrict__ part, const KhBucketWork* __restrict__ work, u32 cap, int k,
    KmerKey<W>* __restrict__ out_keys, u32* __restrict__ out_counts, KhLookback lb, u32 ci, u32 cx,
    u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const SortLds L{lds_raw, cap, W, false};
    KmerKey<W>* s = reinterpret_cast<KmerKey<W>*>(lds_raw + L.keys_off());
    u16* hstart = reinterpret_cast<u16*>(lds_raw + L.hstart_off());
    u32* tab = reinterpret_cast<u32*>(lds_raw + L.tab_off());
    u32* scratch = reinterpret_cast<u32*>(lds_raw + L.scratch_off());
    // pair-mode carve (oversize path): keys[capp] | pay[capp] inside the key region
    const u32 capp = ((cap * 8u * W) / (8u * W + 4u)) & ~63u;
    u32* pay = reinterpret_cast<u32*>(lds_raw + (size_t)capp * 8 * W);
    constexpr int E = ((W == 1 ? KH_SORT_CAP_W1 : KH_SORT_CAP_W2) + KH_SORT_THREADS - 1) / KH_SORT_THREADS;

    const u32 tid = threadIdx.x, nt = blockDim.x;
    // index order: the bucket is known at once and its bounds are fetched while the block clears
    // its bins (the barrier the sort needs after the clear then comes after the key loads have
    // been issued); ticket order: everything waits for the ticket
    const bool early = lb.dynamic == 0;
    KhBucketWork wk{0, 0, 0, 0, 0, 0};
    if (early) wk = work[blockIdx.x];
    distribute_clear(reinterpret_cast<u32*>(hstart), tab + 128, scratch);
    if (!early) {
        if (tid == 0) scratch[16] = atomicAdd(lb.ticket, 1u);
        __syncthreads();
        wk = work[scratch[16]];
    }
    const u32 seg_nb = wk.nb;
    const u64 lo = wk.lo;
    const u64 n64 = wk.n;
    // the segment's own chain: descriptors [gb - b, ...), this bucket is its part number b
    lb.desc += wk.gb - wk.b;
    const u32 q = wk.b;
    out_keys += wk.out_base;
    if (out_counts) out_counts += wk.out_base;
    KH_STAMP(q, 0);

    auto sink = [&](u64 o, const KmerKey<W>& key, u32 c) {
        out_keys[o] = key;
        if (out_counts) out_counts[o] = c < cs ? c : cs;
    };

    if (n64 <= cap) {
        const u32 n = (u32)n64;
        KmerKey<W> kreg[E];
        u32 preg[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            preg[e] = 0;
            kreg[e] = key_zero<W>();
            const u32 i = (u32)e * KH_SORT_THREADS + tid;
            if (i < n) kreg[e] = part[lo + i];
        }
        if (early) __syncthreads();   // bins cleared by every thread before the sort counts into them
#ifdef KH_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        KH_STAMP(q, 1);
        distribute_sort<W, false, E>(kreg, preg, n, s, nullptr, reinterpret_cast<u32*>(hstart),
                                     tab + 128, tab + 128 + KH_FINE_BINS / 32, scratch, k, seg_nb, q);
        auto eval = [&](u32 h0, u32 h1) -> u32 {
            const u32 c = h1 - h0;
            return (c >= ci && c <= cx) ? c : 0u;
        };
        rle_emit<W, E>(s, n, hstart, tab, eval, sink, lb, q, scratch, ci <= 1u && cx == 0xffffffffu,
                       out_counts == nullptr);
        return;
    }

    // ---- oversize bucket: fold chunk by chunk into (key, counter) pairs
    u32 acc = 0;
    u64 consumed = 0;
    bool fail = false;
    while (consumed < n64) {
        if (acc >= capp) { fail = true; break; }
        const u64 left = n64 - consumed;
        const u32 take = left < (u64)(capp - acc) ? (u32)left : (capp - acc);
        const u32 m = acc + take;
        for (u32 i = acc + tid; i < m; i += nt) {
            s[i] = part[lo + consumed + (i - acc)];
            pay[i] = 1u;
        }
        __syncthreads();
        bitonic_sort_lds<W, true>(s, pay, m);
        const u32 dd = find_runs<W>(s, m, hstart, scratch);
        // fold runs in place: slot r <- (key, saturating sum of payloads)
        for (u32 r0 = 0; r0 < dd; r0 += nt) {
            const u32 r = r0 + tid;
            KmerKey<W> kv = key_zero<W>();
            u64 sum = 0;
            if (r < dd) {
                const u32 h0 = hstart[r], h1 = hstart[r + 1];
                kv = s[h0];
                for (u32 j = h0; j < h1; ++j) sum += pay[j];
                if (sum > 0xffffffffull) sum = 0xffffffffull;
            }
            __syncthreads();
            if (r < dd) { s[r] = kv; pay[r] = (u32)sum; }
            __syncthreads();
        }
        acc = dd;
        consumed += take;
    }
    if (fail) {
        if (tid == 0) atomicOr(lb.err, KH_ERR_CAPACITY);
        acc = 0;
    }
    // s[0..acc) now holds distinct keys with their counters in pay[] (acc <= capp < cap)
    auto eval = [&](u32 h0, u32 h1) -> u32 {
        const u32 c = pay[h0];
        return (c >= ci && c <= cx) ? c : 0u;
    };
    rle_emit<W, E>(s, acc, hstart, tab, eval, sink, lb, q, scratch, ci <= 1u && cx == 0xffffffffu);
}

// ------------------------------------------------------------------------------------------
// Grid-mode pass C as a kernel of its own (buckets that fit LDS; oversize buckets are left to
// k_bucket_sort_rle, which folds them).  Without the general path's sort, repair and look-back it
// needs 51 KB of LDS and few enough registers for THREE workgroups per CU instead of two.
// LDS: keys[cap] | bins[KH_FINE_BINS/2 + 1] u32 | tab[64 + 1] | lbits[128] | scratch[32]
// ------------------------------------------------------------------------------------------
size_t kh_grid_bucket_lds_bytes(int W, u32 cap) {
    return (size_t)cap * 8 * W + (size_t)(KH_FINE_BINS / 2 + 4) * 4 + 80 * 4 + 128 * 4 + 32 * 4;
}
template <int W>
__global__ __launch_bounds__(KH_SORT_THREADS, 6) void k_grid_bucket(const KmerKey<W>* __restrict__ part,
                                                                  const KhBucketWork* __restrict__ work, u32 cap, int k,
                                                                  KmerKey<W>* __restrict__ out_keys, const KhGrid grid) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    KmerKey<W>* s = reinterpret_cast<KmerKey<W>*>(lds_raw);
    u32* bins = reinterpret_cast<u32*>(lds_raw + (size_t)cap * 8 * W);
    u32* tab = bins + KH_FINE_BINS / 2 + 4;
    u32* lbits = tab + 80;          // 8-byte aligned: read as 64-bit words
    u32* scratch = lbits + 128;
    constexpr int E = ((W == 1 ? KH_SORT_CAP_W1 : KH_SORT_CAP_W2) + KH_SORT_THREADS - 1) / KH_SORT_THREADS;
    const u32 tid = threadIdx.x;
    const KhBucketWork wk = work[blockIdx.x];
    if (wk.n > cap) return;         // on k_col_offsets' list for k_grid_oversize
    for (u32 i = tid; i <= (u32)KH_FINE_BINS / 2; i += KH_SORT_THREADS) bins[i] = 0;
    if (tid < 128) lbits[tid] = 0;
    const u32 n = wk.n;
    KmerKey<W> kreg[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        kreg[e] = key_zero<W>();
        const u32 i = (u32)e * KH_SORT_THREADS + tid;
        if (i < n) kreg[e] = part[wk.lo + i];
    }
    __syncthreads();
    grid_bucket<W, E>(kreg, n, s, bins, tab, scratch, lbits, k, wk.nb, out_keys + wk.lo,
                      grid.off + (u64)wk.gb * (grid.S + 1), grid.S, grid.distinct + wk.gb / grid.nb, wk.b);
}

// Buckets with more keys than fit LDS (duplicate-heavy input: tiny k, low-complexity sequence) in
// grid mode: k_col_offsets listed them; a small grid walks the list and folds each bucket chunk by
// chunk — [distinct keys so far | next raw chunk] -> sort -> keep the first key of every run —
// then writes the distinct keys and the sub-range index like grid_bucket does (grid_emit: the fold
// leaves them sorted).  Only a bucket with more DISTINCT keys than fit LDS cannot be
// handled: KH_ERR_CAPACITY, the caller falls back to the general path.
template <int W>
__global__ __launch_bounds__(KH_SORT_THREADS, KH_SORT_WAVES_PER_SIMD) void k_grid_oversize(
    const KmerKey<W>* __restrict__ part, const KhBucketWork* __restrict__ work, const u32* __restrict__ over, u32 cap,
    int k, KmerKey<W>* __restrict__ out_keys, u32* __restrict__ err, const KhGrid grid) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const SortLds L{lds_raw, cap, W, false};
    KmerKey<W>* s = reinterpret_cast<KmerKey<W>*>(lds_raw + L.keys_off());
    u16* hstart = reinterpret_cast<u16*>(lds_raw + L.hstart_off());
    u32* tab = reinterpret_cast<u32*>(lds_raw + L.tab_off());
    u32* scratch = reinterpret_cast<u32*>(lds_raw + L.scratch_off());
    const u32 capp = cap;      // plain sets: only the distinct keys are kept while folding
    constexpr int E = ((W == 1 ? KH_SORT_CAP_W1 : KH_SORT_CAP_W2) + KH_SORT_THREADS - 1) / KH_SORT_THREADS;
    const u32 tid = threadIdx.x, nt = blockDim.x;
    const u32 count = over[0];
    for (u32 it = blockIdx.x; it < count; it += gridDim.x) {
        const KhBucketWork wk = work[over[1u + it]];
        const u64 lo = wk.lo, n64 = wk.n;
        u32 acc = 0;
        u64 consumed = 0;
        bool fail = false;
        __syncthreads();   // the previous bucket's readers of s / pay / tab are done
        while (consumed < n64) {
            if (acc >= capp) { fail = true; break; }
            const u64 left = n64 - consumed;
            const u32 take = left < (u64)(capp - acc) ? (u32)left : (capp - acc);
            const u32 m = acc + take;
            for (u32 i = acc + tid; i < m; i += nt) {
                s[i] = part[lo + consumed + (i - acc)];
            }
            __syncthreads();
            bitonic_sort_lds<W, false>(s, nullptr, m);
            const u32 dd = find_runs<W>(s, m, hstart, scratch);
            for (u32 r0 = 0; r0 < dd; r0 += nt) {      // fold runs in place: slot r <- first key of run r
                const u32 r = r0 + tid;
                KmerKey<W> kv = key_zero<W>();
                if (r < dd) kv = s[hstart[r]];
                __syncthreads();
                if (r < dd) s[r] = kv;
                __syncthreads();
            }
            acc = dd;
            consumed += take;
        }
        if (fail) {
            if (tid == 0) atomicOr(err, KH_ERR_CAPACITY);
            acc = 0;
        }
        grid_emit<W, E>(s, acc, tab, out_keys + lo, grid.off + (u64)wk.gb * (grid.S + 1), grid.S, k, wk.nb,
                        grid.distinct + wk.gb / grid.nb);
    }
}

// ------------------------------------------------------------------------------------------
// set operations on sorted sets
// ------------------------------------------------------------------------------------------
// bounds[g * (nranges + 1) + r] = first index of set g whose key falls in slot >= r
template <int W>
__global__ void k_range_bounds(const KhSetView* __restrict__ sets, u32 nsets, u32 nranges, int k,
                               u64* __restrict__ bounds, u64* __restrict__ zero, u64 zero_words) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    // the set operation that follows needs its look-back descriptors, control words and
    // histogram zeroed: done here instead of by separate fill launches
    for (u64 i = idx; i < zero_words; i += (u64)gridDim.x * blockDim.x) zero[i] = 0;
    const u64 per = (u64)nranges + 1;
    if (idx >= per * nsets) return;
    const u32 g = (u32)(idx / per), r = (u32)(idx % per);
    const KhSetView sv = sets[g];
    const KmerKey<W>* keys = reinterpret_cast<const KmerKey<W>*>(sv.keys);
    u64 lo = 0, hi = sv.n;
    if (r == 0) hi = 0;
    if (r >= nranges) lo = hi;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (kh_slot<W>(keys[mid], k, nranges) < r) lo = mid + 1; else hi = mid;
    }
    bounds[idx] = lo;
}

// the same for several independent set operations in one launch (blockIdx.y = operation): the
// group unions of one wave are planned together, and ten latency-bound little launches become one
template <int W>
__global__ void k_range_bounds_batch(const KhBoundsJob* __restrict__ jobs, int k) {
    const KhBoundsJob jb = jobs[blockIdx.y];
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 i = idx; i < jb.zero_words; i += (u64)gridDim.x * blockDim.x) jb.zero[i] = 0;
    const u64 per = (u64)jb.nranges + 1;
    if (idx >= per * jb.nsets) return;
    const u32 g = (u32)(idx / per), r = (u32)(idx % per);
    const KhSetView sv = jb.sets[g];
    const KmerKey<W>* keys = reinterpret_cast<const KmerKey<W>*>(sv.keys);
    u64 lo = 0, hi = sv.n;
    if (r == 0) hi = 0;
    if (r >= jb.nranges) lo = hi;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (kh_slot<W>(keys[mid], k, jb.nranges) < r) lo = mid + 1; else hi = mid;
    }
    jb.bounds[idx] = lo;
}

__device__ __forceinline__ long long combine_counters(int mode, long long a, long long b) {
    switch (mode) {
        case KH_OC_MIN: return a < b ? a : b;
        case KH_OC_MAX: return a > b ? a : b;
        case KH_OC_SUM: return a + b;
        case KH_OC_DIFF: return a - b;
        case KH_OC_LEFT: return a;
        default: return b;
    }
}

// PAY=false: n-ary union of sets whose counters are all 1, counters summed: the counter of a
// key is simply the length of its run, no payload array needed (the step_3 / step_7 case,
// exp_type_1.smk:182,250).  PAY=true: payload = counter | (operand index > 0) << 31.
template <int W, bool PAY>
__global__ __launch_bounds__(KH_SORT_THREADS, KH_SORT_WAVES_PER_SIMD) void k_setop(
    const KhSetopBatch batch, u32 njobs, u32 cap, int k, int op, int mode, u32 cs, u32 hist_len,
    u32 dynamic) {
    // The operations of a batch are INTERLEAVED in start order (workgroup i: operation i % njobs,
    // slot i / njobs; the grid is njobs times the widest operation): a slot's look-back
    // predecessor was started njobs workgroups earlier and has usually published its count by
    // the time it is needed, instead of running in lock step with its successor.
    const u32 job = blockIdx.x % njobs, slot_idx = blockIdx.x / njobs;
    const KhSetopJob& jb = batch.job[job];
    const u32 nranges = jb.nranges;
    if (slot_idx >= nranges) return;
    const KhSetView* __restrict__ sets = jb.sets;
    const u32 nsets = jb.nsets;
    const u64* __restrict__ bounds = jb.bounds;
    KmerKey<W>* __restrict__ out_keys = reinterpret_cast<KmerKey<W>*>(jb.out_keys);
    u32* __restrict__ out_counts = jb.out_counts;
    unsigned long long* __restrict__ hist = jb.hist;
    KhLookback lb;
    lb.desc = jb.desc;
    lb.ticket = jb.ctl;
    lb.err = jb.ctl + 1;
    lb.dynamic = dynamic;
    const u32 slot0 = jb.slot0, nslots = jb.nslots;
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const SortLds L{lds_raw, cap, W, PAY};
    KmerKey<W>* s = reinterpret_cast<KmerKey<W>*>(lds_raw + L.keys_off());
    u32* pay = reinterpret_cast<u32*>(lds_raw + L.pay_off());
    u16* hstart = reinterpret_cast<u16*>(lds_raw + L.hstart_off());
    u32* lhist = reinterpret_cast<u32*>(lds_raw + L.lhist_off());
    u32* tab = reinterpret_cast<u32*>(lds_raw + L.tab_off());
    u32* scratch = reinterpret_cast<u32*>(lds_raw + L.scratch_off());
    u64* bcast = reinterpret_cast<u64*>(lds_raw + L.bcast_off());
    constexpr int CAPC = PAY ? (W == 1 ? KH_SORT_CAP_PAY_W1 : KH_SORT_CAP_PAY_W2)
                             : (W == 1 ? KH_SORT_CAP_W1 : KH_SORT_CAP_W2);
    constexpr int E = (CAPC + KH_SORT_THREADS - 1) / KH_SORT_THREADS;
    const u32 tid = threadIdx.x, nt = blockDim.x, lane = lane_id();
    const u64 per = (u64)nslots + 1;
    // index order: the slot is known at once, so the first wave's descriptor loads are in
    // flight while the block clears its bins; ticket order: they wait for the ticket
    const bool early = lb.dynamic == 0;
    u32 q = slot_idx;             // part number inside this chain; the slot itself is slot0 + q
    // With at most 64 operands every wave describes the slot for ITSELF, lane g holding operand g
    // (redundant loads, served by the caches): no LDS descriptors, no "wave 0 scans, everyone
    // waits", no chain of dependent LDS reads in front of the key loads.
    const bool wave_local = nsets <= (u32)KH_WAVE;
    u64 pre_b0 = 0, pre_b1 = 0;
    KhSetView pre_sv{nullptr, nullptr, 0, 0, 0};
    if (early && (wave_local || tid < KH_WAVE) && lane < nsets) {
        pre_b0 = bounds[lane * per + slot0 + q];
        pre_b1 = bounds[lane * per + slot0 + q + 1];
        pre_sv = sets[lane];
    }
    distribute_clear(reinterpret_cast<u32*>(hstart), tab + 128, scratch);
    if (!early) {
        if (tid == 0) scratch[16] = atomicAdd(lb.ticket, 1u);
        __syncthreads();
        q = scratch[16];
    }
    KH_STAMP(q, 0);

    KmerKey<W> kreg[E];
    u32 preg[E];
    u32 n = 0;
    if (wave_local) {
        const bool have = lane < nsets;
        if (!early && have) {
            pre_b0 = bounds[lane * per + slot0 + q];
            pre_b1 = bounds[lane * per + slot0 + q + 1];
            pre_sv = sets[lane];
        }
        u64 chain_in = (slot0 && have) ? bounds[lane * per + slot0] : 0ull;   // inputs in front of this chain
        // slice lengths are scanned as 32-bit values on the DPP path; a slice of 2^24 records or
        // more is far beyond any slot capacity and only needs to be reported as "too full"
        // bounds that run backwards or past the operand's end (an operand that is not sorted by
        // mixed key: kh_set_wrap_device / kh_set_from_device trust their caller) must never reach the
        // gather: the slot is dropped and KH_ERR_ORDER raised
        const bool bad_bounds = have && (pre_b1 < pre_b0 || pre_b1 > pre_sv.n);
        if (__builtin_amdgcn_ballot_w64(bad_bounds) && tid == 0) atomicOr(lb.err, KH_ERR_ORDER);
        const u64 len64 = (have && !bad_bounds) ? pre_b1 - pre_b0 : (bad_bounds ? 0xffffffffull : 0ull);
        const u32 len = len64 > 0xffffffull ? 0xffffffu : (u32)len64;
        const u32 incl = wave_scan_add(len);
        const u64 n64 = (u32)__builtin_amdgcn_readlane((int)incl, KH_WAVE - 1);
        if (slot0) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) chain_in += __shfl_xor(chain_in, off);
        }
        out_keys += chain_in;
        if (out_counts) out_counts += chain_in;
        if (n64 > (u64)CAPC || n64 > (u64)cap) {
            if (tid == 0) {
                atomicOr(lb.err, KH_ERR_CAPACITY);
                atomicMax(lb.err + 1, n64 > 0xffffffffull ? 0xffffffffu : (u32)n64);   // fullest slot seen
            }
        } else {
            n = (u32)n64;
        }
        // operands that do not exist start "after everything"
        const u32 soff = have ? incl - len : 0xffffffffu;
        const u64 sbeg = pre_b0, skey = reinterpret_cast<u64>(pre_sv.keys), scnt = reinterpret_cast<u64>(pre_sv.counts);
        const u32 suni = pre_sv.uniform;
        auto lane64 = [](u64 v, u32 src) -> u64 {   // value of lane `src` (uniform), via scalar reads
            const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, (int)src);
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), (int)src);
            return ((u64)hi << 32) | lo;
        };
#pragma unroll
        for (int e = 0; e < E; ++e) {
            kreg[e] = key_zero<W>();
            preg[e] = 0;
            // this wave's 64 elements of pass e are [B, B + 64): nearly always inside ONE operand
            const u32 B = (u32)e * KH_SORT_THREADS + (tid & ~(u32)(KH_WAVE - 1));
            const u32 i = B + lane;
            const u32 g_lo = (u32)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(soff <= B))) - 1u;
            const u32 g_hi = (u32)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(soff <= B + (KH_WAVE - 1)))) - 1u;
            u32 ga = g_lo, my_soff;
            u64 my_sbeg, my_skey, my_scnt = 0;
            u32 my_suni = 0;
            if (g_lo == g_hi) {
                my_soff = (u32)__builtin_amdgcn_readlane((int)soff, (int)g_lo);
                my_sbeg = lane64(sbeg, g_lo);
                my_skey = lane64(skey, g_lo);
                if (PAY) { my_scnt = lane64(scnt, g_lo); my_suni = (u32)__builtin_amdgcn_readlane((int)suni, (int)g_lo); }
            } else {   // an operand boundary inside the 64 elements: per-lane choice
                for (u32 gg = g_lo + 1; gg <= g_hi; ++gg)
                    ga += ((u32)__builtin_amdgcn_readlane((int)soff, (int)gg) <= i) ? 1u : 0u;
                my_soff = __shfl(soff, ga);
                my_sbeg = __shfl(sbeg, ga);
                my_skey = __shfl(skey, ga);
                if (PAY) { my_scnt = __shfl(scnt, ga); my_suni = __shfl(suni, ga); }
            }
            if (i < n) {
                const u64 idx = my_sbeg + (i - my_soff);
                kreg[e] = reinterpret_cast<const KmerKey<W>*>(my_skey)[idx];
                if (PAY) {
                    const u32* cp = reinterpret_cast<const u32*>(my_scnt);
                    u32 c = cp ? cp[idx] : my_suni;
                    if (c > 0x7fffffffu) c = 0x7fffffffu;
                    // binary operations tag the second operand; n-ary unions only ever sum
                    preg[e] = c | ((nsets == 2 && ga == 1) ? 0x80000000u : 0u);
                }
            }
        }
    } else {

    // operand slices of this slot, described in LDS while gathering:
    //   soff[g] u32  first gathered index of operand g's slice (exclusive scan)
    //   suni[g] u32  uniform counter of operand g
    //   sbeg[g] u64  first element of the slice inside operand g
    //   skey[g] u64  operand g's key array
    //   scnt[g] u64  operand g's counter array (0 = uniform)
    // (they borrow the head of the key array, which is not written before the gather is over)
    constexpr u32 MAXG = KH_MAX_INPUT_SETS;
    u32* dsc = reinterpret_cast<u32*>(s);
    u32* soff = dsc;
    u32* suni = dsc + MAXG;
    u64* sbeg = reinterpret_cast<u64*>(dsc + 2 * MAXG);
    u64* skey = reinterpret_cast<u64*>(dsc + 4 * MAXG);
    u64* scnt = reinterpret_cast<u64*>(dsc + 6 * MAXG);
    if (tid < KH_WAVE) {   // one wave scans the slice lengths, 64 operands per round
        u64 carry = 0, chain_in = 0;
        for (u32 g0 = 0; g0 < nsets; g0 += KH_WAVE) {
            const u32 g = g0 + tid;
            u64 len = 0;
            if (g < nsets) {
                const bool pre = early && g0 == 0;
                const u64 b0 = pre ? pre_b0 : bounds[g * per + slot0 + q];
                const u64 b1 = pre ? pre_b1 : bounds[g * per + slot0 + q + 1];
                const KhSetView sv = pre ? pre_sv : sets[g];
                if (slot0) chain_in += bounds[g * per + slot0];   // inputs in front of this chain
                sbeg[g] = b0;
                skey[g] = reinterpret_cast<u64>(sv.keys);
                scnt[g] = reinterpret_cast<u64>(sv.counts);
                suni[g] = sv.uniform;
                if (b1 < b0 || b1 > sv.n) {   // see the wave-local form: never a wrapped length
                    atomicOr(lb.err, KH_ERR_ORDER);
                    len = 0xffffffffull;
                } else {
                    len = b1 - b0;
                }
            }
            u64 incl = len;
#pragma unroll
            for (int off = 1; off < KH_WAVE; off <<= 1) {
                const u64 u = __shfl_up(incl, off);
                if (lane >= (u32)off) incl += u;
            }
            if (g < nsets) {
                const u64 ex = carry + incl - len;
                soff[g] = ex > 0xffffffffull ? 0xffffffffu : (u32)ex;
            }
            carry += __shfl(incl, KH_WAVE - 1);
        }
        if (slot0) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) chain_in += __shfl_xor(chain_in, off);
        }
        if (tid == 0) { bcast[1] = carry; bcast[2] = chain_in; }
    }
    __syncthreads();
    const u64 n64 = bcast[1];
    // a chain that does not start at slot 0 writes where its inputs start: no output of an
    // earlier chain can reach that far (every output key is one of the inputs)
    out_keys += bcast[2];
    if (out_counts) out_counts += bcast[2];
    if (n64 > (u64)CAPC || n64 > (u64)cap) {
        if (tid == 0) {
            atomicOr(lb.err, KH_ERR_CAPACITY);
            atomicMax(lb.err + 1, n64 > 0xffffffffull ? 0xffffffffu : (u32)n64);   // fullest slot seen
        }
    } else {
        n = (u32)n64;
    }
    // gather straight into registers: element i of the concatenated slices belongs to the last
    // operand whose slice starts at or before i; all loads of a thread are independent
    u32 ga = 0;   // operand of this thread's current element: found once, then only advanced
    {
        u32 gb = nsets;
        while (gb - ga > 1) {
            const u32 m = (ga + gb) >> 1;
            if (soff[m] <= tid) ga = m; else gb = m;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        kreg[e] = key_zero<W>();
        preg[e] = 0;
        const u32 i = (u32)e * KH_SORT_THREADS + tid;
        if (i < n) {
            while (ga + 1 < nsets && soff[ga + 1] <= i) ++ga;
            const u64 idx = sbeg[ga] + (i - soff[ga]);
            kreg[e] = reinterpret_cast<const KmerKey<W>*>(skey[ga])[idx];
            if (PAY) {
                const u32* cp = reinterpret_cast<const u32*>(scnt[ga]);
                u32 c = cp ? cp[idx] : suni[ga];
                if (c > 0x7fffffffu) c = 0x7fffffffu;
                // binary operations tag the second operand; n-ary unions only ever sum
                preg[e] = c | ((nsets == 2 && ga == 1) ? 0x80000000u : 0u);
            }
        }
    }
    }
    __syncthreads();   // the slice descriptors are no longer needed
    for (u32 i = tid; i < KH_LHIST_BINS; i += nt) lhist[i] = 0;
#ifdef KH_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    KH_STAMP(q, 1);
    distribute_sort<W, PAY, E>(kreg, preg, n, s, pay, reinterpret_cast<u32*>(hstart), tab + 128,
                               tab + 128 + KH_FINE_BINS / 32, scratch, k, nslots, q);

    // counter of run s[h0..h1) under the requested operation (0 = key dropped)
    auto eval = [&](u32 h0, u32 h1) -> u32 {
        long long c;
        if (!PAY) {
            c = (long long)(h1 - h0);
        } else {
            long long ca = 0, cb = 0;
            bool ha = false, hb = false;
            for (u32 t = h0; t < h1; ++t) {
                const u32 p = pay[t];
                if (p >> 31) { cb += p & 0x7fffffffu; hb = true; } else { ca += p; ha = true; }
            }
            switch (op) {
                case KH_OP_UNION:
                    c = (ha && hb) ? combine_counters(mode, ca, cb) : (ca + cb);
                    break;
                case KH_OP_INTERSECT:
                    c = (ha && hb) ? combine_counters(mode, ca, cb) : 0;
                    break;
                case KH_OP_KMERS_SUBTRACT:
                    c = (ha && !hb) ? ca : 0;
                    break;
                default:   // counters subtract
                    c = ha ? ca - cb : 0;
                    break;
            }
        }
        if (c <= 0) return 0u;
        return c > (long long)cs ? cs : (u32)c;
    };
    // Fused histogram: almost every counter of a union is tiny (1 for the across-group sum,
    // 1..G within a group), so a plain LDS histogram would serialise a whole wave on one
    // address.  Counters below 16 go to 64 lane-private copies (the sort's work-list region is
    // free by now) that are folded into lhist afterwards.
    u32* stripe = tab + 128 + KH_FINE_BINS / 32;            // [16][64]
    static_assert(KH_WORKLIST >= 16 * 64, "lane-private histogram needs the work-list region");
    if (hist) {
        __syncthreads();                                    // the sort is done with its work list
        for (u32 i = tid; i < 16 * 64; i += nt) stripe[i] = 0;
    }
    auto sink = [&](u64 o, const KmerKey<W>& key, u32 c) {
        out_keys[o] = key;
        if (out_counts) out_counts[o] = c;
        if (hist) {
            if (c < 16u) atomicAdd(&stripe[c * 64 + lane], 1u);
            else if (c < KH_LHIST_BINS) atomicAdd(&lhist[c], 1u);
            else atomicAdd(&hist[c < hist_len ? c : hist_len - 1], 1ull);
        }
    };
    const u64 chain_out = rle_emit<W, E>(s, n, hstart, tab, eval, sink, lb, q, scratch,
                                         op == KH_OP_UNION && mode != KH_OC_DIFF);
    if (q == nranges - 1 && tid == 0)   // the chain is complete: its outputs join the operation's total
        atomicAdd(reinterpret_cast<unsigned long long*>(jb.ctl + 4), (unsigned long long)chain_out);
    if (hist) {
        __syncthreads();
        for (u32 c = tid >> 6; c < 16; c += nt >> 6) {      // one wave folds two counters
            u32 v = stripe[c * 64 + lane];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && v) atomicAdd(&lhist[c], v);
        }
        __syncthreads();
        for (u32 i = tid; i < KH_LHIST_BINS; i += nt) {
            const u32 v = lhist[i];
            if (v) atomicAdd(&hist[i < hist_len ? i : hist_len - 1], (unsigned long long)v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tagged n-ary union over gridded genome sets (KhTagJob in kh_launch.h): steps 3+4 and 7+8 of
// exp_type_1.smk in one pass, with no per-group database in between.  One workgroup per slot
// r = b * S + f (looping over slots when nothing is emitted): lane g of every wave looks operand
// g's slice up in the bucket index, the slices are gathered into registers with the operand
// number as payload, sorted in LDS like any other set operation, and every run of equal keys is
// turned into a genome mask by the thread that holds its first element.
// ------------------------------------------------------------------------------------------
template <int W, bool EMIT>
__global__ __launch_bounds__(KH_SORT_THREADS, KH_SORT_WAVES_PER_SIMD) void k_union_tagged(
    const KhTagJob jb, u32 cap, int k, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    const SortLds L{lds_raw, cap, W, true};
    KmerKey<W>* s = reinterpret_cast<KmerKey<W>*>(lds_raw + L.keys_off());
    u32* pay = reinterpret_cast<u32*>(lds_raw + L.pay_off());
    u16* hstart = reinterpret_cast<u16*>(lds_raw + L.hstart_off());
    // Carve (kh_tag_lds_bytes): the sort's carve, ginfo[64] in its lhist region, then the compact
    // histogram hstripe[nbins][8] (8 copies per bin, copy = lane & 7).
    u32* ginfo = reinterpret_cast<u32*>(lds_raw + L.lhist_off());
    u32* tab = reinterpret_cast<u32*>(lds_raw + L.tab_off());
    u32* scratch = reinterpret_cast<u32*>(lds_raw + L.scratch_off());
    u32* hstripe = reinterpret_cast<u32*>(lds_raw + L.total());   // [nbins][8]
    constexpr int CAPC = W == 1 ? KH_SORT_CAP_PAY_W1 : KH_SORT_CAP_PAY_W2;
    constexpr int E = (CAPC + KH_SORT_THREADS - 1) / KH_SORT_THREADS;
    constexpr u32 NT = KH_SORT_THREADS;
    const u32 tid = threadIdx.x, lane = lane_id();
    const u32 nb = jb.nb, S = jb.S, nops = jb.nops, nbins = jb.nbins;
    const u32 nslots = nb * S;
    const KmerKey<W>* __restrict__ keys = reinterpret_cast<const KmerKey<W>*>(jb.keys);
    constexpr bool emit = EMIT;
    KmerKey<W>* __restrict__ out_keys = reinterpret_cast<KmerKey<W>*>(jb.out_keys);
    u32* __restrict__ out_counts = jb.out_counts;
    KhLookback lb;
    lb.desc = jb.desc;
    lb.ticket = jb.ctl + 2;
    lb.err = jb.ctl;
    lb.dynamic = 0;

    for (u32 i = tid; i < (u32)KH_TAG_MAX_OPS; i += NT) ginfo[i] = jb.ginfo[i];
    for (u32 i = tid; i < nbins * 8u; i += NT) hstripe[i] = 0;

    // One slot per workgroup.  (A workgroup looping over slots kept the histogram in LDS longer, but
    // the loop invariants hipcc hoisted out of it were spilled, and every scratch reload is an
    // s_waitcnt vmcnt(0) in the middle of the gather: the eight key loads of a thread ran one
    // after the other, 17 K cycles instead of 4 K.  A spill-free persistent form that also kept the
    // keys of the next slot and the index of the one after in flight was measured too: 1.73-1.82 ms
    // against 1.67 — with two workgroups per CU the hash-set form is bound by the LDS atomic rate,
    // not by the two memory round trips in front of it.)
    {
        const u32 r = blockIdx.x;
        const u32 b = r / S, f = r - b * S;
        KH_STAMP(r, 0);
        // ---- operand slices of this slot: lane g of every wave describes operand g
        const bool have = lane < nops;
        u64 sbeg = 0;
        u32 len = 0;
        if (have) {
            const u32 gb = lane * nb + b;
            const u16* __restrict__ o = jb.off + (u64)gb * (S + 1) + f;
            const u32 o0 = o[0], o1 = o[1];
            sbeg = jb.bstart[gb] + o0;
            len = o1 >= o0 ? o1 - o0 : 0xffffffu;   // a corrupt index reads as "too full", never as a wrap
            if (o1 < o0) atomicOr(jb.ctl, KH_ERR_ORDER);
        }
        distribute_clear(reinterpret_cast<u32*>(hstart), tab + 128, scratch);
        const u32 incl = wave_scan_add(len);
        const u32 n64 = (u32)__builtin_amdgcn_readlane((int)incl, KH_WAVE - 1);
        u32 n = 0;
        if (n64 > (u32)CAPC || n64 > cap) {
            if (tid == 0) {
                atomicOr(lb.err, KH_ERR_CAPACITY);
                atomicMax(lb.err + 1, n64);
            }
        } else {
            n = n64;
        }
        const u32 soff = have ? incl - len : 0xffffffffu;
        auto lane64 = [](u64 v, u32 src) -> u64 {
            const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, (int)src);
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), (int)src);
            return ((u64)hi << 32) | lo;
        };
        KmerKey<W> kreg[E];
        u32 preg[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            kreg[e] = key_zero<W>();
            preg[e] = 0;
            const u32 B = (u32)e * NT + (tid & ~(u32)(KH_WAVE - 1));
            const u32 i = B + lane;
            if (B >= n) continue;   // wave-uniform
            const u32 g_lo = (u32)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(soff <= B))) - 1u;
            const u32 g_hi = (u32)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(soff <= B + (KH_WAVE - 1)))) - 1u;
            u32 ga = g_lo, my_soff;
            u64 my_sbeg;
            if (g_lo == g_hi) {
                my_soff = (u32)__builtin_amdgcn_readlane((int)soff, (int)g_lo);
                my_sbeg = lane64(sbeg, g_lo);
            } else {   // operand boundaries inside the 64 elements: per-lane choice
                for (u32 gg = g_lo + 1; gg <= g_hi; ++gg)
                    ga += ((u32)__builtin_amdgcn_readlane((int)soff, (int)gg) <= i) ? 1u : 0u;
                my_soff = __shfl(soff, ga);
                my_sbeg = __shfl(sbeg, ga);
            }
            if (i < n) {
                kreg[e] = keys[my_sbeg + (i - my_soff)];
                preg[e] = ga;
            }
        }
        __syncthreads();   // bins cleared
#ifdef KH_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        KH_STAMP(r, 1);
        // fine bin of a key inside this slot: the slot is fine bins [fb0, fb0 + width) of bucket b
        // (kh_first_bin), i.e. positions [fb0 << 19, ...) of the bucket's 32-bit position scale;
        // binmul stretches the widest slot over the KH_FINE_BINS bins.  A key from outside the slot
        // (corrupt index) lands in the last bin instead of outside the table.
        const u32 rel0 = kh_first_bin(f, S) << (32 - KH_FINE_BITS);
        const u32 binmul = jb.binmul, nbv = jb.nbv;
        auto binfn = [=](const KmerKey<W>& key) -> u32 {
            const u32 frac = (u32)((u64)kh_top32(key, k) * (u64)nbv);
            const u32 fb = (u32)(((u64)(frac - rel0) * (u64)binmul) >> 32);
            return fb < (u32)KH_FINE_BINS ? fb : (u32)KH_FINE_BINS - 1u;
        };
        // a genome mask -> per-group counts (step_4 bins) + number of groups (returned)
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
        if constexpr (!emit) {
            // (one-word keys normally take the hash-set form, k_union_hash below; this is the form for
            // two-word keys, which have no 128-bit compare-and-swap to meet in a table with)
            // Nothing is written, so nothing has to be sorted: equal keys share a fine bin, every
            // key finds the first copy of itself there (find_leaders) and ORs its genome bit into
            // that copy's 64-bit mask.  The masks overlay the payload + bin-table carve (both free
            // once the keys are placed); tags stay in registers.
            u32 fr[E], at[E], vmask, bmax;
            distribute_place<W, false, E>(kreg, preg, n, s, nullptr, reinterpret_cast<u32*>(hstart), scratch, binfn,
                                          r, fr, at, vmask, bmax);
            unsigned long long* rmask = reinterpret_cast<unsigned long long*>(pay);
            u32 lead[E];
            find_leaders<W, E>(kreg, s, fr, at, vmask, lead);
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((vmask & (1u << e)) && lead[e] == at[e]) rmask[at[e]] = 1ull << (preg[e] & 63u);
            KH_STAMP(r, 11);
            __syncthreads();
            KH_STAMP(r, 5);
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((vmask & (1u << e)) && lead[e] != at[e]) atomicOr(&rmask[lead[e]], 1ull << (preg[e] & 63u));
            __syncthreads();
            KH_STAMP(r, 6);
            KH_STAMP(r, 7);
            u64 mk[E];
#pragma unroll
            for (int e = 0; e < E; ++e) mk[e] = ((vmask & (1u << e)) && lead[e] == at[e]) ? rmask[at[e]] : 0ull;
            u32 gi[E];   // first group of every mask (nearly always the only one): table reads issued together
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
            // the across-group bin "1" would otherwise take one LDS atomic per key: per-wave sum
            ones = wave_scan_add(ones);
            if (lane == KH_WAVE - 1 && ones) atomicAdd(&hstripe[(jb.abase + 1u) * 8u], ones);
            __syncthreads();
            KH_STAMP(r, 8);
        } else {
            // the across-group set is written (multi-GPU exchange): full sort, ordered output
            distribute_sort_bf<W, true, E>(kreg, preg, n, s, pay, reinterpret_cast<u32*>(hstart), tab + 128,
                                           tab + 128 + KH_FINE_BINS / 32, scratch, binfn, r);
            auto eval = [&](u32 h0, u32 h1) -> u32 {
                u64 mask = 0;
                for (u32 t = h0; t < h1; ++t) mask |= 1ull << (pay[t] & 63u);
                return eval_mask(mask, ginfo[__ffsll((unsigned long long)mask) - 1]);
            };
            auto sink = [&](u64 o, const KmerKey<W>& key, u32 c) {
                out_keys[o] = key;
                out_counts[o] = c;
                atomicAdd(&hstripe[(jb.abase + c) * 8u + (lane & 7u)], 1u);
            };
            const u64 chain_out = rle_emit<W, E>(s, n, hstart, tab, eval, sink, lb, r, scratch, true);
            if (r == nslots - 1 && tid == 0) *jb.out_n = chain_out;
            __syncthreads();
        }
    }
    __syncthreads();
    unsigned long long* __restrict__ rep = jb.hist + (u64)(blockIdx.x % jb.reps) * nbins;
    for (u32 i = tid; i < nbins; i += NT) {
        u32 v = 0;
#pragma unroll
        for (u32 j = 0; j < 8; ++j) v += hstripe[i * 8u + j];
        if (v) atomicAdd(&rep[i], (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------
// The tagged union of one-word keys when nothing is emitted (the benchmark path), as a kernel of
// its own: LDS hash set of T entries {key, genome mask}, NT threads, 8 keys per thread, one slot
// per workgroup (see the hash-set notes in k_union_tagged, whose other forms keep serving two-word
// keys and emitting runs).  Templated on the geometry, which was measured both ways: <256, 2048>
// (four workgroups per CU, 39 KB each) and <512, 4096> (two) run the headline step in the same
// 1.70 ms — the per-slot chain of round trips does not shorten with the slot, so the throughput is
// (keys resident in LDS per CU) / (length of that chain) either way.  <512, 4096> is what runs.
//
// Why a hash set: with nothing to emit, nothing has to be ordered — equal keys only have to MEET.
// Home entry = the key's in-slot fine bin / 2 (keys are uniform inside a slot), linear probing.  A
// 64-bit LDS compare-and-swap claims an empty entry or returns the key living there: one round
// trip tells a key where its first copy is, and its genome bit is ORed in there.  Linear probing
// has long clusters (max ~30 entries at this load) and the whole workgroup would wait for the one
// wave that walks the longest: a key gets KH_HASH_ROUNDS probes in the main table, then moves to a
// small second table with an independent hash, and from a full second table back to unbounded
// probing of the first.  Occupied entries stay occupied, so every copy of a key takes the same
// decisions as the first.  The all-ones key (the empty marker; k = 32 only) is carried beside.
// ------------------------------------------------------------------------------------------
template <u32 T> size_t kh_union_hash_lds(u32 nbins) {
    return (size_t)T * 16 + 256 + 128 + (((size_t)nbins * 32 + 15) & ~(size_t)15) + (size_t)(T / 8) * 16;
}

template <u32 NT, u32 T>
__global__ __launch_bounds__(NT, (T == 2048 ? 4 : 2) * (NT / 64) / 4) void k_union_hash(const KhTagJob jb, int k, u32 cs) {
    extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];
    constexpr int E = (int)(T / NT);
    constexpr u32 T2 = T / 8;                 // second table
    constexpr u32 HBITS = T == 4096 ? 12 : 11;
    constexpr u64 EMPTY = ~0ull;
    struct alignas(16) Ent { unsigned long long key, mask; };
    Ent* tbl = reinterpret_cast<Ent*>(lds_raw);                                               // [T]
    u32* ginfo = reinterpret_cast<u32*>(lds_raw + (size_t)T * 16);                           // [64]
    u32* scratch = ginfo + 64;                                                               // [32]
    unsigned long long* special = reinterpret_cast<unsigned long long*>(scratch + 24);       // mask of the key ~0
    u32* hstripe = scratch + 32;                                                             // [nbins][8]
    const u32 nbins = jb.nbins;
    Ent* ovf = reinterpret_cast<Ent*>(hstripe + ((nbins * 8u + 3u) & ~3u));                  // [T2]
    const u32 tid = threadIdx.x, lane = lane_id();
    const u32 nb = jb.nb, S = jb.S, nops = jb.nops;
    const KmerKey<1>* __restrict__ keys = reinterpret_cast<const KmerKey<1>*>(jb.keys);
    const u32 r = blockIdx.x;
    const u32 b = r / S, f = r - b * S;
    // ---- operand slices of this slot: lane g of every wave describes operand g
    const bool have = lane < nops;
    u64 sbeg = 0;
    u32 len = 0;
    if (have) {
        const u32 gb = lane * nb + b;
        const u16* __restrict__ o = jb.off + (u64)gb * (S + 1) + f;
        const u32 o0 = o[0], o1 = o[1];
        sbeg = jb.bstart[gb] + o0;
        len = o1 >= o0 ? o1 - o0 : 0xffffffu;   // a corrupt index reads as "too full", never as a wrap
        if (o1 < o0) atomicOr(jb.ctl, KH_ERR_ORDER);
    }
    // tables and tables' neighbours, written while the index loads are in flight
    for (u32 i = tid; i < (u32)KH_TAG_MAX_OPS; i += NT) ginfo[i] = jb.ginfo[i];
    for (u32 i = tid; i < nbins * 8u; i += NT) hstripe[i] = 0;
    {
        uint4* t4 = reinterpret_cast<uint4*>(lds_raw);
#pragma unroll
        for (int e = 0; e < E; ++e) t4[(u32)e * NT + tid] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
        uint4* o4 = reinterpret_cast<uint4*>(ovf);
        for (u32 i = tid; i < T2; i += NT) o4[i] = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
        if (tid == 0) *special = 0ull;
    }
    const u32 incl = wave_scan_add(len);
    const u32 n64 = (u32)__builtin_amdgcn_readlane((int)incl, KH_WAVE - 1);
    u32 n = 0;
    if (n64 > T) {
        if (tid == 0) {
            atomicOr(jb.ctl, KH_ERR_CAPACITY);
            atomicMax(jb.ctl + 1, n64);
        }
    } else {
        n = n64;
    }
    const u32 soff = have ? incl - len : 0xffffffffu;
    u64 kreg[E];
    u32 tagp[(E + 3) / 4];
#pragma unroll
    for (int w = 0; w < (E + 3) / 4; ++w) tagp[w] = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        kreg[e] = EMPTY;
        const u32 B = (u32)e * NT + (tid & ~(u32)(KH_WAVE - 1));
        const u32 i = B + lane;
        if (B >= n) continue;   // wave-uniform
        const u32 g_lo = (u32)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(soff <= B))) - 1u;
        const u32 g_hi = (u32)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(soff <= B + (KH_WAVE - 1)))) - 1u;
        u32 ga = g_lo, my_soff;
        u64 my_sbeg;
        if (g_lo == g_hi) {
            my_soff = (u32)__builtin_amdgcn_readlane((int)soff, (int)g_lo);
            const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)sbeg, (int)g_lo);
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(sbeg >> 32), (int)g_lo);
            my_sbeg = ((u64)hi << 32) | lo;
        } else {   // operand boundaries inside the 64 elements: per-lane choice
            for (u32 gg = g_lo + 1; gg <= g_hi; ++gg)
                ga += ((u32)__builtin_amdgcn_readlane((int)soff, (int)gg) <= i) ? 1u : 0u;
            my_soff = __shfl(soff, ga);
            my_sbeg = __shfl(sbeg, ga);
        }
        if (i < n) {
            kreg[e] = keys[my_sbeg + (i - my_soff)].lo;
            tagp[e >> 2] |= ga << (8 * (e & 3));
        }
    }
    __syncthreads();   // tables initialised
    auto tag = [&](int e) -> u32 { return (tagp[e >> 2] >> (8 * (e & 3))) & 63u; };
    const u32 rel0 = kh_first_bin(f, S) << (32 - KH_FINE_BITS);
    const u32 binmul = jb.binmul, nbv = jb.nbv;
    auto home = [=](u64 key) -> u32 {
        const u32 frac = (u32)((u64)kh_top32(KmerKey<1>{key}, k) * (u64)nbv);
        u32 fb = (u32)(((u64)(frac - rel0) * (u64)binmul) >> 32);
        fb = fb < (u32)KH_FINE_BINS ? fb : (u32)KH_FINE_BINS - 1u;
        return fb >> (KH_FINE_BITS - HBITS);
    };
    u32 slot[E], act = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        slot[e] = home(kreg[e]);
        if ((u32)e * NT + tid < n) {
            if (kreg[e] == EMPTY) atomicOr(special, 1ull << tag(e));
            else act |= 1u << e;
        }
    }
#define KH_PROBE_ROUNDS(TBL, TMASK, ROUNDS)                                                                           \
    for (u32 round = 0; round < (ROUNDS) && __builtin_amdgcn_ballot_w64(act != 0); ++round) {                        \
        unsigned long long old[E];                                                                                    \
        _Pragma("unroll") for (int e = 0; e < E; ++e)                                                                 \
            old[e] = (act & (1u << e)) ? atomicCAS(&(TBL)[slot[e]].key, EMPTY, (unsigned long long)kreg[e]) : 0ull;   \
        _Pragma("unroll") for (int e = 0; e < E; ++e) {                                                               \
            if (act & (1u << e)) {                                                                                    \
                if (old[e] == EMPTY || old[e] == kreg[e]) {                                                           \
                    atomicOr(&(TBL)[slot[e]].mask, 1ull << tag(e));                                                   \
                    act &= ~(1u << e);                                                                                \
                } else {                                                                                              \
                    slot[e] = (slot[e] + 1u) & (TMASK);                                                               \
                }                                                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    }
    KH_PROBE_ROUNDS(tbl, T - 1u, (u32)KH_HASH_ROUNDS)
    if (__builtin_amdgcn_ballot_w64(act != 0)) {
#pragma unroll
        for (int e = 0; e < E; ++e) slot[e] = (u32)((kreg[e] * KH_C3) >> 40) & (T2 - 1u);
        KH_PROBE_ROUNDS(ovf, T2 - 1u, T2)
        if (__builtin_amdgcn_ballot_w64(act != 0)) {   // second table full of other keys: on in the main table
#pragma unroll
            for (int e = 0; e < E; ++e) slot[e] = (home(kreg[e]) + (u32)KH_HASH_ROUNDS) & (T - 1u);
            KH_PROBE_ROUNDS(tbl, T - 1u, T)
            if (__builtin_amdgcn_ballot_w64(act != 0) && lane == 0) atomicOr(jb.ctl, KH_ERR_CAPACITY);
        }
    }
#undef KH_PROBE_ROUNDS
    __syncthreads();
    // ---- every occupied entry is one distinct key of the slot: genome mask -> histogram bins
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
    if (tid == 0 && *special) {   // the one key that cannot live in the tables (all bits set, k = 32 only)
        const u64 m = *special;
        const u32 ng = eval_mask(m, ginfo[__ffsll((unsigned long long)m) - 1]);
        if (ng == 1u) ++ones;
        else atomicAdd(&hstripe[(jb.abase + ng) * 8u], 1u);
    }
    ones = wave_scan_add(ones);
    if (lane == KH_WAVE - 1 && ones) atomicAdd(&hstripe[(jb.abase + 1u) * 8u], ones);
    __syncthreads();
    unsigned long long* __restrict__ rep = jb.hist + (u64)(blockIdx.x % jb.reps) * nbins;
    for (u32 i = tid; i < nbins; i += NT) {
        u32 v = 0;
#pragma unroll
        for (u32 j = 0; j < 8; ++j) v += hstripe[i * 8u + j];
        if (v) atomicAdd(&rep[i], (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------
// Block-private histogram whose low bins (where nearly all counters of a k-mer database fall)
// are kept once per lane, so that a wave does not serialise on one LDS address.
struct BlockHist {
    u32* lh;       // [KH_LHIST_BINS]
    u32* stripe;   // [16][64]
    __device__ void clear() {
        for (u32 i = threadIdx.x; i < KH_LHIST_BINS; i += blockDim.x) lh[i] = 0;
        for (u32 i = threadIdx.x; i < 16 * 64; i += blockDim.x) stripe[i] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void add(u32 c, unsigned long long* hist, u32 hist_len) {
        if (c < 16u) atomicAdd(&stripe[c * 64 + (threadIdx.x & 63)], 1u);
        else if (c < KH_LHIST_BINS) atomicAdd(&lh[c], 1u);
        else atomicAdd(&hist[c < hist_len ? c : hist_len - 1], 1ull);
    }
    __device__ void flush(unsigned long long* hist, u32 hist_len) {
        __syncthreads();
        for (u32 i = threadIdx.x; i < 16; i += blockDim.x) {
            u32 v = 0;
            for (u32 l = 0; l < 64; ++l) v += stripe[i * 64 + ((l + i) & 63)];
            lh[i] += v;
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < KH_LHIST_BINS; i += blockDim.x) {
            const u32 v = lh[i];
            if (v) atomicAdd(&hist[i < hist_len ? i : hist_len - 1], (unsigned long long)v);
        }
    }
};

__global__ __launch_bounds__(256) void k_histogram(const u32* __restrict__ counts, u64 n,
                                                  unsigned long long* __restrict__ hist,
                                                  u32 hist_len) {
    __shared__ u32 lh[KH_LHIST_BINS];
    __shared__ u32 stripe[16 * 64];
    BlockHist bh{lh, stripe};
    bh.clear();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bh.add(counts[i], hist, hist_len);
    bh.flush(hist, hist_len);
}

template <int W, bool UNMIX>
__global__ void k_remix(const KmerKey<W>* __restrict__ in, KmerKey<W>* __restrict__ out, u64 n,
                        int k) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = UNMIX ? kh_unmix(in[i], k) : kh_mix(in[i], k);
}

__global__ void k_fill_u32(u32* __restrict__ p, u64 n, u32 v) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// Direct-addressed occurrence table for small k (k <= 16, 4^k cells).
// A set holds DISTINCT keys, so one launch touches every cell at most once: a plain byte (or
// dword) read-modify-write is race-free; different sets are added in stream order.
// ------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(256) void k_table_add(const KmerKey<1>* __restrict__ keys, u64 n, int k,
                                                  C* __restrict__ table, u32 cmax) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 v = kh_unmix(keys[i], k).lo;
        const u32 c = table[v];
        if (c < cmax) table[v] = (C)(c + 1);
    }
}

// hist[min(c, cs, hist_len-1)] += 1 for every non-zero cell of table[lo, hi); cells are read
// 16 bytes at a time (lo and hi are multiples of 16 cells except at the very end of the table).
template <class C>
__global__ __launch_bounds__(256) void k_table_hist(const C* __restrict__ table, u64 lo, u64 hi, u32 cs,
                                                   unsigned long long* __restrict__ hist, u32 hist_len) {
    __shared__ u32 lh[KH_LHIST_BINS];
    __shared__ u32 stripe[16 * 64];
    BlockHist bh{lh, stripe};
    bh.clear();
    constexpr u32 PER = 16 / sizeof(C);
    const u64 nvec = (hi - lo) / PER;
    const uint4* __restrict__ vp = reinterpret_cast<const uint4*>(table + lo);
    const u64 stride = (u64)gridDim.x * blockDim.x;
    auto put = [&](u32 c) {
        if (!c) return;
        bh.add(c > cs ? cs : c, hist, hist_len);
    };
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const uint4 q = vp[i];
        const u32 w4[4] = {q.x, q.y, q.z, q.w};
        if (!(q.x | q.y | q.z | q.w)) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (sizeof(C) == 4) put(w4[j]);
            else if (w4[j]) {
                put(w4[j] & 0xffu); put((w4[j] >> 8) & 0xffu);
                put((w4[j] >> 16) & 0xffu); put(w4[j] >> 24);
            }
        }
    }
    if (blockIdx.x == 0)      // tail cells (fewer than one vector)
        for (u64 i = lo + nvec * PER + threadIdx.x; i < hi; i += blockDim.x) put((u32)table[i]);
    bh.flush(hist, hist_len);
}

// ------------------------------------------------------------------------------------------
// Membership matrix (experiment type 4, src/merge_lists.py:14-33): for every key of the pivot
// set, bit d of its mask = "set d holds the key".  One thread per pivot key; every operand is
// sorted by MIXED key and mixed keys are uniform, so the search starts from an interpolated
// window of +-(4 sqrt(n) + 64) records and falls back to the whole set when the window misses.
// ------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void k_membership(const KmerKey<W>* __restrict__ pivot, u64 n,
                                                   const KhSetView* __restrict__ sets, u32 nsets, int k,
                                                   u32 nwords, u64* __restrict__ masks) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KmerKey<W> key = pivot[i];
    const u64 top = (u64)kh_top32(key, k) << 32;
    for (u32 w = 0; w < nwords; ++w) {
        u64 m = 0;
        const u32 dend = min(nsets, (w + 1) * 64);
        for (u32 d = w * 64; d < dend; ++d) {
            const KhSetView sv = sets[d];
            if (!sv.n) continue;
            const KmerKey<W>* __restrict__ keys = reinterpret_cast<const KmerKey<W>*>(sv.keys);
            const u64 est = __umul64hi(top, sv.n);
            const u64 rad = (u64)(4.0f * sqrtf((float)sv.n)) + 64;
            u64 lo = est > rad ? est - rad : 0;
            u64 hi = est + rad < sv.n ? est + rad : sv.n;
            if (lo > 0 && !key_lt(keys[lo - 1], key)) lo = 0;        // answer lies left of the window
            if (hi < sv.n && key_lt(keys[hi], key)) hi = sv.n;       // ... or right of it
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (key_lt(keys[mid], key)) lo = mid + 1; else hi = mid;
            }
            if (lo < sv.n && key_eq(keys[lo], key)) m |= 1ull << (d & 63);
        }
        masks[i * nwords + w] = m;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static u32 grid_for(u64 n, u32 block, u32 cap_blocks = 2048) {
    u64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap_blocks) g = cap_blocks;
    return (u32)g;
}

template <class K> static void allow_lds(K kern, size_t bytes) {
    static thread_local size_t granted = 0;   // per kernel instantiation
    if (bytes > 48 * 1024 && bytes > granted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        granted = bytes;
    }
}

void kh_launch_extract(int W, bool scatter, const u8* seq, const KhSeg* segs, const KhTile* tiles,
                       u32 ntiles, u32 nb_alloc, int k, u32* thist, const u64* bstart, void* part,
                       u32 tile_pos, hipStream_t st) {
    if (!ntiles) return;
    if (scatter && nb_alloc <= 4 * KH_ST_THREADS && kh_extract_staged_lds_bytes(W, nb_alloc) <= 160 * 1024 &&
        !getenv("KHOICE_DIRECT_SCATTER")) {
        const size_t lds2 = kh_extract_staged_lds_bytes(W, nb_alloc);
        if (W == 1) {
            allow_lds(k_extract_staged<1>, lds2);
            hipLaunchKernelGGL(k_extract_staged<1>, dim3(ntiles), dim3(KH_ST_THREADS), lds2, st, seq, segs, tiles,
                               nb_alloc, k, thist, bstart, reinterpret_cast<KmerKey<1>*>(part), tile_pos);
        } else {
            allow_lds(k_extract_staged<2>, lds2);
            hipLaunchKernelGGL(k_extract_staged<2>, dim3(ntiles), dim3(KH_ST_THREADS), lds2, st, seq, segs, tiles,
                               nb_alloc, k, thist, bstart, reinterpret_cast<KmerKey<2>*>(part), tile_pos);
        }
        return;
    }
    const size_t lds = kh_extract_lds_bytes(nb_alloc);
#define KH_EX(WW, SC)                                                                           \
    do {                                                                                        \
        allow_lds(k_extract<WW, SC>, lds);                                                      \
        hipLaunchKernelGGL((k_extract<WW, SC>), dim3(ntiles), dim3(256), lds, st, seq, segs,    \
                           tiles, nb_alloc, k, thist, bstart,                                   \
                           reinterpret_cast<KmerKey<WW>*>(part), tile_pos);                     \
    } while (0)
    if (W == 1) { if (scatter) KH_EX(1, true); else KH_EX(1, false); }
    else        { if (scatter) KH_EX(2, true); else KH_EX(2, false); }
#undef KH_EX
}

void kh_launch_col_totals(const KhSeg* segs, u32 nseg, u32 max_nb, const u32* thist, u64* tot,
                          hipStream_t st) {
    if (!nseg || !max_nb) return;
    hipLaunchKernelGGL(k_col_totals, dim3((max_nb + 63) / 64, nseg), dim3(64, KH_COL_TY), 0, st, segs,
                       thist, tot);
}
void kh_launch_col_offsets(const KhSeg* segs, u32 nseg, u32 max_nb, u32* thist, const u64* bstart,
                           const u32* rank, const u64* seg_out_base, KhBucketWork* work, u32* over, u32 over_cap,
                           hipStream_t st) {
    if (!nseg || !max_nb) return;
    hipLaunchKernelGGL(k_col_offsets, dim3((max_nb + 63) / 64, nseg), dim3(64, KH_COL_TY), 0, st, segs,
                       thist, bstart, rank, seg_out_base, work, over, over_cap);
}
size_t kh_exscan_tmp_words(u64 n) { return (size_t)((n + KH_SCAN_TILE - 1) / KH_SCAN_TILE) + 1; }
void kh_launch_exscan(const u64* in, u64* out, u64 n, u64* tmp, hipStream_t st) {
    const u32 grid = (u32)std::max<u64>(1, (n + KH_SCAN_TILE - 1) / KH_SCAN_TILE);
    hipLaunchKernelGGL((k_exscan<0>), dim3(grid), dim3(1024), 0, st, in, out, n, tmp);
    hipLaunchKernelGGL((k_exscan<1>), dim3(grid), dim3(1024), 0, st, in, out, n, tmp);
}

void kh_launch_bucket_sort(int W, const void* part, const KhBucketWork* work,
                           u32 nbuckets, int k, void* out_keys, u32* out_counts, KhLookback lb,
                           u32 ci, u32 cx, u32 cs, hipStream_t st) {
    if (!nbuckets) return;
    const u32 cap = W == 1 ? KH_SORT_CAP_W1 : KH_SORT_CAP_W2;
    const size_t lds = kh_sort_lds_bytes(W, cap, false);
    if (W == 1) {
        allow_lds(k_bucket_sort_rle<1>, lds);
        hipLaunchKernelGGL((k_bucket_sort_rle<1>), dim3(nbuckets), dim3(KH_SORT_THREADS), lds, st,
                           reinterpret_cast<const KmerKey<1>*>(part), work, cap, k,
                           reinterpret_cast<KmerKey<1>*>(out_keys), out_counts, lb, ci, cx, cs);
    } else {
        allow_lds(k_bucket_sort_rle<2>, lds);
        hipLaunchKernelGGL((k_bucket_sort_rle<2>), dim3(nbuckets), dim3(KH_SORT_THREADS), lds, st,
                           reinterpret_cast<const KmerKey<2>*>(part), work, cap, k,
                           reinterpret_cast<KmerKey<2>*>(out_keys), out_counts, lb, ci, cx, cs);
    }
}

size_t kh_tag_lds_bytes(int W, u32 cap, u32 nbins, bool emit) {
    (void)emit;
    return kh_sort_lds_bytes(W, cap, true) + (((size_t)nbins * 32 + 15) & ~(size_t)15);
}
void kh_launch_union_tagged(int W, const KhTagJob& job, u32 grid, int k, u32 cs, hipStream_t st) {
    if (!grid) return;
    const u32 cap = W == 1 ? KH_SORT_CAP_PAY_W1 : KH_SORT_CAP_PAY_W2;
    const size_t lds = kh_tag_lds_bytes(W, cap, job.nbins, job.desc != nullptr);
#define KH_UT(WW, EE)                                                                                   \
    do {                                                                                                \
        allow_lds(k_union_tagged<WW, EE>, lds);                                                         \
        hipLaunchKernelGGL((k_union_tagged<WW, EE>), dim3(grid), dim3(KH_SORT_THREADS), lds, st, job, cap, k, cs); \
    } while (0)
    const bool emit = job.desc != nullptr;
    if (W == 1) { if (emit) KH_UT(1, true); else KH_UT(1, false); }
    else        { if (emit) KH_UT(2, true); else KH_UT(2, false); }
#undef KH_UT
}

void kh_launch_range_bounds(int W, const KhSetView* sets, u32 nsets, u32 nranges, int k,
                            u64* bounds, u64* zero, u64 zero_words, hipStream_t st) {
    const u64 total = ((u64)nranges + 1) * nsets;
    const u32 grid = (u32)((total + 255) / 256);
    if (W == 1)
        hipLaunchKernelGGL((k_range_bounds<1>), dim3(grid), dim3(256), 0, st, sets, nsets, nranges,
                           k, bounds, zero, zero_words);
    else
        hipLaunchKernelGGL((k_range_bounds<2>), dim3(grid), dim3(256), 0, st, sets, nsets, nranges,
                           k, bounds, zero, zero_words);
}

void kh_launch_range_bounds_batch(int W, const KhBoundsJob* jobs, u32 njobs, u64 max_threads, int k,
                                  hipStream_t st) {
    if (!njobs) return;
    const u32 grid = (u32)((max_threads + 255) / 256);
    if (W == 1)
        hipLaunchKernelGGL((k_range_bounds_batch<1>), dim3(grid, njobs), dim3(256), 0, st, jobs, k);
    else
        hipLaunchKernelGGL((k_range_bounds_batch<2>), dim3(grid, njobs), dim3(256), 0, st, jobs, k);
}

void kh_launch_setop(int W, bool pay, u32 cap, const KhSetopBatch& batch, u32 njobs, int k, int op, int mode,
                     u32 cs, u32 hist_len, bool dynamic_order, hipStream_t st) {
    u32 width = 0;
    for (u32 i = 0; i < njobs; ++i) width = std::max(width, batch.job[i].nranges);
    if (!njobs || !width) return;
    const size_t lds = kh_sort_lds_bytes(W, cap, pay);
#define KH_SO(WW, PP)                                                                            \
    do {                                                                                         \
        allow_lds(k_setop<WW, PP>, lds);                                                         \
        hipLaunchKernelGGL((k_setop<WW, PP>), dim3(width * njobs), dim3(KH_SORT_THREADS), lds, st, \
                           batch, njobs, cap, k, op, mode, cs, hist_len, dynamic_order ? 1u : 0u); \
    } while (0)
    if (W == 1) { if (pay) KH_SO(1, true); else KH_SO(1, false); }
    else        { if (pay) KH_SO(2, true); else KH_SO(2, false); }
#undef KH_SO
}

void kh_launch_histogram(const u32* counts, u64 n, unsigned long long* hist, u32 hist_len,
                         hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_histogram, dim3(grid_for(n, 256 * 8)), dim3(256), 0, st, counts, n, hist,
                       hist_len);
}
void kh_launch_unmix(int W, const void* in, void* out, u64 n, int k, hipStream_t st) {
    if (!n) return;
    if (W == 1)
        hipLaunchKernelGGL((k_remix<1, true>), dim3(grid_for(n, 256)), dim3(256), 0, st,
                           reinterpret_cast<const KmerKey<1>*>(in),
                           reinterpret_cast<KmerKey<1>*>(out), n, k);
    else
        hipLaunchKernelGGL((k_remix<2, true>), dim3(grid_for(n, 256)), dim3(256), 0, st,
                           reinterpret_cast<const KmerKey<2>*>(in),
                           reinterpret_cast<KmerKey<2>*>(out), n, k);
}
void kh_launch_fill_u32(u32* p, u64 n, u32 v, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n, 256)), dim3(256), 0, st, p, n, v);
}

void kh_launch_table_add(const void* keys, u64 n, int k, void* table, u32 cell_bytes, hipStream_t st) {
    if (!n) return;
    if (cell_bytes == 1)
        hipLaunchKernelGGL((k_table_add<u8>), dim3(grid_for(n, 256, 8192)), dim3(256), 0, st,
                           reinterpret_cast<const KmerKey<1>*>(keys), n, k, reinterpret_cast<u8*>(table), 255u);
    else
        hipLaunchKernelGGL((k_table_add<u32>), dim3(grid_for(n, 256, 8192)), dim3(256), 0, st,
                           reinterpret_cast<const KmerKey<1>*>(keys), n, k, reinterpret_cast<u32*>(table),
                           0xFFFFFFFFu);
}
void kh_launch_table_hist(const void* table, u32 cell_bytes, u64 lo, u64 hi, u32 cs,
                          unsigned long long* hist, u32 hist_len, hipStream_t st) {
    if (hi <= lo) return;
    const u64 nvec = (hi - lo) / (16 / cell_bytes);
    if (cell_bytes == 1)
        hipLaunchKernelGGL((k_table_hist<u8>), dim3(grid_for(nvec, 256 * 4, 4096)), dim3(256), 0, st,
                           reinterpret_cast<const u8*>(table), lo, hi, cs, hist, hist_len);
    else
        hipLaunchKernelGGL((k_table_hist<u32>), dim3(grid_for(nvec, 256 * 4, 4096)), dim3(256), 0, st,
                           reinterpret_cast<const u32*>(table), lo, hi, cs, hist, hist_len);
}

void kh_launch_membership(int W, const void* pivot, u64 n, const KhSetView* sets, u32 nsets, int k,
                          u32 nwords, u64* masks, hipStream_t st) {
    if (!n) return;
    const u32 grid = (u32)((n + 255) / 256);
    if (W == 1)
        hipLaunchKernelGGL((k_membership<1>), dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const KmerKey<1>*>(pivot), n, sets, nsets, k, nwords, masks);
    else
        hipLaunchKernelGGL((k_membership<2>), dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const KmerKey<2>*>(pivot), n, sets, nsets, k, nwords, masks);
}

u32 kh_union_hash_capacity() { return 4096u; }
void kh_launch_union_hash(const KhTagJob& job, u32 grid, int k, u32 cs, hipStream_t st) {
    if (!grid) return;
    const size_t lds = kh_union_hash_lds<4096>(job.nbins);
    allow_lds(k_union_hash<512, 4096>, lds);
    hipLaunchKernelGGL((k_union_hash<512, 4096>), dim3(grid), dim3(512), lds, st, job, k, cs);
}

u32 kh_grid_bucket_capacity(int W) { return W == 1 ? KH_SORT_CAP_W1 : KH_SORT_CAP_W2; }
void kh_launch_grid_bucket(int W, const void* part, const KhBucketWork* work, u32 nbuckets, int k, void* out_keys,
                           const u32* over, u32* err, const KhGrid& grid, hipStream_t st) {
    if (!nbuckets) return;
    const u32 cap = kh_grid_bucket_capacity(W);
    {   // the listed oversize buckets (usually none: the walkers find an empty list and leave)
        const size_t lds2 = kh_sort_lds_bytes(W, cap, false);
        const u32 walkers = std::min<u32>(nbuckets, 128);
        if (W == 1) {
            allow_lds(k_grid_oversize<1>, lds2);
            hipLaunchKernelGGL((k_grid_oversize<1>), dim3(walkers), dim3(KH_SORT_THREADS), lds2, st,
                               reinterpret_cast<const KmerKey<1>*>(part), work, over, cap, k,
                               reinterpret_cast<KmerKey<1>*>(out_keys), err, grid);
        } else {
            allow_lds(k_grid_oversize<2>, lds2);
            hipLaunchKernelGGL((k_grid_oversize<2>), dim3(walkers), dim3(KH_SORT_THREADS), lds2, st,
                               reinterpret_cast<const KmerKey<2>*>(part), work, over, cap, k,
                               reinterpret_cast<KmerKey<2>*>(out_keys), err, grid);
        }
    }
    const size_t lds = kh_grid_bucket_lds_bytes(W, cap);
    if (W == 1) {
        allow_lds(k_grid_bucket<1>, lds);
        hipLaunchKernelGGL((k_grid_bucket<1>), dim3(nbuckets), dim3(KH_SORT_THREADS), lds, st,
                           reinterpret_cast<const KmerKey<1>*>(part), work, cap, k, reinterpret_cast<KmerKey<1>*>(out_keys), grid);
    } else {
        allow_lds(k_grid_bucket<2>, lds);
        hipLaunchKernelGGL((k_grid_bucket<2>), dim3(nbuckets), dim3(KH_SORT_THREADS), lds, st,
                           reinterpret_cast<const KmerKey<2>*>(part), work, cap, k, reinterpret_cast<KmerKey<2>*>(out_keys), grid);
    }
}
