// khoice_amd — device side of the FASTA ingest (SURVEY.md §8f "next" #2; inputs
// data/dataset_N/*.fna.gz of workflow/rules/exp_type_1.smk:44-47,158).
//
// The host inflates a (gz) multi-FASTA file into pinned memory and ships the RAW bytes; these
// kernels turn them into the cleaned sequence text kh_read_fasta produces on the CPU, byte for
// byte, directly in HBM:
//   * a line whose first character (ignoring '\r') is '>' is a header line and is dropped;
//   * '\n' and '\r' are dropped, every other byte of a sequence line is kept verbatim
//     (N / IUPAC / lower case still reach the k-mer extraction, which breaks runs on them);
//   * one '\n' is put in front of a header line when sequence bytes have been written since the
//     previous header (records are never joined: SURVEY App. A.1).
// It is a stream compaction over tiles of 4096 bytes: pass 1 counts what every tile keeps, pass 2
// scans the tile counts (and carries "bytes kept since the last header" across tiles), pass 3 writes.
// The one sequential dependency — is the line that runs into a tile a header line? — is resolved
// by the host while the file is still in cache (a memchr sweep over line starts, one flag per tile).
#include <hip/hip_runtime.h>

#include "kh_launch.h"

constexpr u32 KI_TILE = 4096;          // bytes per workgroup
constexpr u32 KI_THREADS = 256;
constexpr u32 KI_PER = KI_TILE / KI_THREADS;   // 16 bytes per thread

struct KiTileSum {      // what pass 2 needs to know about a tile
    u32 keep;           // bytes kept
    u32 nsep;           // separators decided inside the tile (for its 2nd, 3rd, ... header line)
    u32 kept_before_first_hdr;
    u32 kept_after_last_hdr;
    u32 has_hdr;
    u32 pad;
};
struct KiTileBase {     // what pass 3 needs to know: written by pass 2
    u64 out;            // output offset of the tile's first emitted byte
    u32 carry;          // sequence bytes were written since the last header (or the start of the file)
    u32 pad;
};

__device__ __forceinline__ u32 ki_lane() { return threadIdx.x & 63u; }
// inclusive scans over the 256 threads of the block (4 waves); scratch: 8 u32
__device__ __forceinline__ u32 ki_scan_add(u32 v, u32* scratch, u32& total) {
    u32 x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off);
        if (ki_lane() >= (u32)off) x += y;
    }
    const u32 wid = threadIdx.x >> 6;
    __syncthreads();
    if (ki_lane() == 63) scratch[wid] = x;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (u32 w = 0; w < KI_THREADS / 64; ++w) {
        const u32 s = scratch[w];
        base += w < wid ? s : 0u;
        tot += s;
    }
    total = tot;
    return x + base;
}
__device__ __forceinline__ u32 ki_scan_max(u32 v, u32* scratch) {
    u32 x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off);
        if (ki_lane() >= (u32)off) x = x > y ? x : y;
    }
    const u32 wid = threadIdx.x >> 6;
    __syncthreads();
    if (ki_lane() == 63) scratch[wid] = x;
    __syncthreads();
    u32 base = 0;
    for (u32 w = 0; w < wid; ++w) base = scratch[w] > base ? scratch[w] : base;
    return x > base ? x : base;
}

// The 16 bytes of a thread, classified.  line_code: for every byte that STARTS a line, the value
// ((position in tile + 1) << 1) | (the line is a header line); a max-scan of it tells every byte
// which line it is on.  A line is a header line when its first byte that is not '\r' is '>' (the
// CPU reader skips '\r' without leaving "start of line").
struct KiChunk {
    u8 b[KI_PER];
    u32 n;            // valid bytes
    u32 code[KI_PER]; // line code of a byte that starts a line, else 0
};

__device__ __forceinline__ void ki_load(const u8* __restrict__ raw, u64 n, u64 p0, KiChunk& c) {
    c.n = p0 < n ? (u32)((n - p0) < KI_PER ? (n - p0) : KI_PER) : 0u;
    if (c.n == KI_PER) {
        const uint4 v = *reinterpret_cast<const uint4*>(raw + p0);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (u32 j = 0; j < KI_PER; ++j) c.b[j] = (u8)(w[j >> 2] >> (8 * (j & 3)));
    } else {
#pragma unroll
        for (u32 j = 0; j < KI_PER; ++j) c.b[j] = j < c.n ? raw[p0 + j] : (u8)'\n';
    }
}

// is byte at absolute position p the start of a line?  (previous byte '\n', looking through '\r's;
// position 0 starts a line)
__device__ __forceinline__ bool ki_line_start(const u8* __restrict__ raw, u64 p) {
    while (p > 0) {
        const u8 prev = raw[p - 1];
        if (prev == '\n') return true;
        if (prev != '\r') return false;
        --p;
    }
    return true;
}
// first byte of the line starting at p that is not '\r' (0 when the data ends first)
__device__ __forceinline__ u8 ki_first_char(const u8* __restrict__ raw, u64 n, u64 p) {
    while (p < n && raw[p] == '\r') ++p;
    return p < n ? raw[p] : (u8)0;
}

__device__ __forceinline__ void ki_classify(const u8* __restrict__ raw, u64 n, u64 p0, u32 tile_off, KiChunk& c) {
#pragma unroll
    for (u32 j = 0; j < KI_PER; ++j) {
        c.code[j] = 0;
        if (j < c.n) {
            // a byte starts a line when the byte before it is '\n' (in-chunk for j > 0); a '\r' right
            // after a line start leaves the line start where it is, which ki_first_char accounts for
            const bool ls = j ? (c.b[j - 1] == '\n') : ki_line_start(raw, p0);
            const bool real = j ? true : (p0 == 0 || raw[p0 - 1] == '\n');   // exact start, not one seen through '\r's
            if (ls && real) {
                const u8 fc = c.b[j] == '\r' ? ki_first_char(raw, n, p0 + j) : c.b[j];
                c.code[j] = ((tile_off + j + 1u) << 1) | (fc == '>' ? 1u : 0u);
            }
        }
    }
}

// Walk a chunk: hdr[j] = byte j lies on a header line; keep[j] = it is written out.
// `before` = line code in force at the chunk's first byte (0: the line that ran into the tile,
// whose kind is `entry_hdr`).
__device__ __forceinline__ void ki_flags(const KiChunk& c, u32 before, bool entry_hdr, u32& keepmask, u32& hdrstart) {
    keepmask = 0;
    hdrstart = 0;   // bit j: byte j is the first byte of a header line
    u32 cur = before;
#pragma unroll
    for (u32 j = 0; j < KI_PER; ++j) {
        if (j >= c.n) break;
        if (c.code[j]) {
            cur = c.code[j];
            if (cur & 1u) hdrstart |= 1u << j;
        }
        const bool hdr = cur ? (cur & 1u) : entry_hdr;
        if (!hdr && c.b[j] != '\n' && c.b[j] != '\r') keepmask |= 1u << j;
    }
}

// ---------------------------------------------------------------- pass 1: per-tile counts
__global__ __launch_bounds__(KI_THREADS) void k_fasta_summary(const u8* __restrict__ raw, u64 n,
                                                            const u8* __restrict__ entry_hdr,
                                                            KiTileSum* __restrict__ sums) {
    __shared__ u32 scratch[8];
    __shared__ u32 red[4];
    const u32 t = blockIdx.x, tid = threadIdx.x;
    const u64 p0 = (u64)t * KI_TILE + (u64)tid * KI_PER;
    KiChunk c;
    ki_load(raw, n, p0, c);
    ki_classify(raw, n, p0, tid * KI_PER, c);
    u32 last = 0;
#pragma unroll
    for (u32 j = 0; j < KI_PER; ++j) last = c.code[j] > last ? c.code[j] : last;
    const u32 incl = ki_scan_max(last, scratch);
    u32 before = __shfl_up(incl, 1);
    if (ki_lane() == 0) before = 0;
    {   // exclusive max over the threads before this one
        __syncthreads();
        if (ki_lane() == 63) scratch[tid >> 6] = incl;
        __syncthreads();
        u32 b2 = ki_lane() ? before : 0u;
        for (u32 w = 0; w < (tid >> 6); ++w) b2 = scratch[w] > b2 ? scratch[w] : b2;
        before = b2;
    }
    u32 keepmask, hdrstart;
    ki_flags(c, before, entry_hdr[t] != 0, keepmask, hdrstart);
    // kept bytes in front of every byte of the tile
    u32 total_keep;
    const u32 kincl = ki_scan_add((u32)__popc(keepmask), scratch, total_keep);
    const u32 kbase = kincl - (u32)__popc(keepmask);
    // K at the header starts: the tile's first and last header start, separators between header lines
    u32 first_k = 0xffffffffu, last_k1 = 0;   // K at the first header start; K + 1 at the last one of this thread
    u32 hm = hdrstart;
    while (hm) {
        const u32 j = (u32)__ffs(hm) - 1u;
        hm &= hm - 1u;
        const u32 kh = kbase + (u32)__popc(keepmask & ((1u << j) - 1u));
        if (first_k == 0xffffffffu) first_k = kh;
        last_k1 = kh + 1u;
    }
    // previous header start's K + 1 for this thread's header starts (0: none before in the tile)
    const u32 pincl = ki_scan_max(last_k1, scratch);
    u32 prevk1 = __shfl_up(pincl, 1);
    {
        __syncthreads();
        if (ki_lane() == 63) scratch[tid >> 6] = pincl;
        __syncthreads();
        u32 b2 = ki_lane() ? prevk1 : 0u;
        for (u32 w = 0; w < (tid >> 6); ++w) b2 = scratch[w] > b2 ? scratch[w] : b2;
        prevk1 = b2;
    }
    u32 nsep = 0;
    hm = hdrstart;
    while (hm) {
        const u32 j = (u32)__ffs(hm) - 1u;
        hm &= hm - 1u;
        const u32 kh = kbase + (u32)__popc(keepmask & ((1u << j) - 1u));
        if (prevk1 && kh + 1u > prevk1) ++nsep;    // sequence bytes since the previous header line of this tile
        prevk1 = kh + 1u;
    }
    u32 tot_sep;
    ki_scan_add(nsep, scratch, tot_sep);
    // block-wide: K at the first header start (min) and K + 1 at the last (max)
    if (tid == 0) { red[0] = 0xffffffffu; red[1] = 0; }
    __syncthreads();
    if (first_k != 0xffffffffu) atomicMin(&red[0], first_k);
    if (last_k1) atomicMax(&red[1], last_k1);
    __syncthreads();
    if (tid == 0) {
        KiTileSum s;
        s.keep = total_keep;
        s.nsep = tot_sep;
        s.has_hdr = red[1] ? 1u : 0u;
        s.kept_before_first_hdr = red[1] ? red[0] : total_keep;
        s.kept_after_last_hdr = red[1] ? total_keep - (red[1] - 1u) : total_keep;
        s.pad = 0;
        sums[t] = s;
    }
}

// ---------------------------------------------------------------- pass 2: scan over the tiles
// One workgroup per file; the carry ("sequence bytes since the last header") makes it a serial
// recurrence, which one thread walks over summaries staged in LDS (a few thousand tiles).
__global__ __launch_bounds__(KI_THREADS) void k_fasta_scan(const KiTileSum* __restrict__ sums, u32 ntiles,
                                                         KiTileBase* __restrict__ base,
                                                         unsigned long long* __restrict__ out_len) {
    __shared__ KiTileSum s[1024];
    __shared__ KiTileBase b[1024];
    u64 out = 0;
    u32 carry = 0;
    for (u32 t0 = 0; t0 < ntiles; t0 += 1024) {
        const u32 m = ntiles - t0 < 1024u ? ntiles - t0 : 1024u;
        __syncthreads();
        for (u32 i = threadIdx.x; i < m; i += KI_THREADS) s[i] = sums[t0 + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            for (u32 i = 0; i < m; ++i) {
                b[i].out = out;
                b[i].carry = carry;
                b[i].pad = 0;
                const KiTileSum x = s[i];
                if (x.has_hdr) {
                    const u32 first_sep = (carry || x.kept_before_first_hdr) ? 1u : 0u;
                    out += (u64)x.keep + x.nsep + first_sep;
                    carry = x.kept_after_last_hdr ? 1u : 0u;
                } else {
                    out += x.keep;
                    carry = (carry || x.keep) ? 1u : 0u;
                }
            }
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < m; i += KI_THREADS) base[t0 + i] = b[i];
    }
    if (threadIdx.x == 0) *out_len = out;
}

// ---------------------------------------------------------------- pass 3: write
__global__ __launch_bounds__(KI_THREADS) void k_fasta_emit(const u8* __restrict__ raw, u64 n,
                                                         const u8* __restrict__ entry_hdr,
                                                         const KiTileBase* __restrict__ base,
                                                         u8* __restrict__ out) {
    __shared__ u32 scratch[8];
    const u32 t = blockIdx.x, tid = threadIdx.x;
    const u64 p0 = (u64)t * KI_TILE + (u64)tid * KI_PER;
    KiChunk c;
    ki_load(raw, n, p0, c);
    ki_classify(raw, n, p0, tid * KI_PER, c);
    u32 last = 0;
#pragma unroll
    for (u32 j = 0; j < KI_PER; ++j) last = c.code[j] > last ? c.code[j] : last;
    const u32 incl = ki_scan_max(last, scratch);
    u32 before = __shfl_up(incl, 1);
    {
        __syncthreads();
        if (ki_lane() == 63) scratch[tid >> 6] = incl;
        __syncthreads();
        u32 b2 = ki_lane() ? before : 0u;
        for (u32 w = 0; w < (tid >> 6); ++w) b2 = scratch[w] > b2 ? scratch[w] : b2;
        before = b2;
    }
    u32 keepmask, hdrstart;
    ki_flags(c, before, entry_hdr[t] != 0, keepmask, hdrstart);
    u32 total_keep;
    const u32 kincl = ki_scan_add((u32)__popc(keepmask), scratch, total_keep);
    const u32 kbase = kincl - (u32)__popc(keepmask);
    const KiTileBase tb = base[t];
    // K + 1 at the previous header start (0: none before in the tile -> the carried state decides)
    u32 last_k1 = 0;
    u32 hm = hdrstart;
    while (hm) {
        const u32 j = (u32)__ffs(hm) - 1u;
        hm &= hm - 1u;
        last_k1 = kbase + (u32)__popc(keepmask & ((1u << j) - 1u)) + 1u;
    }
    const u32 pincl = ki_scan_max(last_k1, scratch);
    u32 prevk1 = __shfl_up(pincl, 1);
    {
        __syncthreads();
        if (ki_lane() == 63) scratch[tid >> 6] = pincl;
        __syncthreads();
        u32 b2 = ki_lane() ? prevk1 : 0u;
        for (u32 w = 0; w < (tid >> 6); ++w) b2 = scratch[w] > b2 ? scratch[w] : b2;
        prevk1 = b2;
    }
    u32 sepmask = 0;   // bit j: a separator goes in front of the header line starting at byte j
    hm = hdrstart;
    while (hm) {
        const u32 j = (u32)__ffs(hm) - 1u;
        hm &= hm - 1u;
        const u32 kh = kbase + (u32)__popc(keepmask & ((1u << j) - 1u));
        const bool sep = prevk1 ? (kh + 1u > prevk1) : (tb.carry || kh > 0u);
        if (sep) sepmask |= 1u << j;
        prevk1 = kh + 1u;
    }
    u32 tot_sep;
    const u32 sincl = ki_scan_add((u32)__popc(sepmask), scratch, tot_sep);
    u32 srun = sincl - (u32)__popc(sepmask);
    u32 krun = kbase;
    u8* __restrict__ o = out + tb.out;
#pragma unroll
    for (u32 j = 0; j < KI_PER; ++j) {
        if (j >= c.n) break;
        if (sepmask & (1u << j)) { o[krun + srun] = (u8)'\n'; ++srun; }
        if (keepmask & (1u << j)) { o[krun + srun] = c.b[j]; ++krun; }
    }
}

// ---------------------------------------------------------------- launcher
size_t kh_fasta_clean_workspace(u64 raw_len) {
    const u64 ntiles = (raw_len + KI_TILE - 1) / KI_TILE;
    return (size_t)(ntiles * (sizeof(KiTileSum) + sizeof(KiTileBase)) + 64);
}
u32 kh_fasta_tile_bytes() { return KI_TILE; }

// raw[0..n) (device, 16-byte aligned), entry_hdr[ntiles] (device), out (device, >= n bytes),
// ws (device, kh_fasta_clean_workspace(n) bytes), out_len (device u64)
void kh_launch_fasta_clean(const u8* raw, u64 n, const u8* entry_hdr, u8* out, void* ws,
                           unsigned long long* out_len, hipStream_t st) {
    const u32 ntiles = (u32)((n + KI_TILE - 1) / KI_TILE);
    if (!ntiles) {
        (void)hipMemsetAsync(out_len, 0, 8, st);
        return;
    }
    KiTileSum* sums = reinterpret_cast<KiTileSum*>(ws);
    KiTileBase* base = reinterpret_cast<KiTileBase*>(sums + ntiles);
    hipLaunchKernelGGL(k_fasta_summary, dim3(ntiles), dim3(KI_THREADS), 0, st, raw, n, entry_hdr, sums);
    hipLaunchKernelGGL(k_fasta_scan, dim3(1), dim3(KI_THREADS), 0, st, sums, ntiles, base, out_len);
    hipLaunchKernelGGL(k_fasta_emit, dim3(ntiles), dim3(KI_THREADS), 0, st, raw, n, entry_hdr, base, out);
}
