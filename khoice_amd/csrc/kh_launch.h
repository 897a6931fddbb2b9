// Host-callable launchers for the gfx950 kernels in kh_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "kh_common.h"

// one input sequence (genome) of a batched build
struct KhSeg {
    const u8* seq;    // device pointer to base 0 (16-B aligned; bytes past len are never read)
    u64 len;          // bases (bytes)
    u64 npos;         // k-mer start positions: len >= k ? len-k+1 : 0
    u64 thist_base;   // index of tile 0 / bucket 0 in the tile-histogram matrix
    u32 nbuckets;     // B_s: buckets this build keeps (cursor rows, bucket arrays)
    u32 bucket_base;  // index of its first bucket in the global bucket arrays
    u32 tile_base;    // global index of its first tile
    u32 ntiles;
    // Key-range waves (a build that keeps only one slice of the key space, kh_exp1_run under a memory
    // budget): the key space is cut into nb_virtual equal-width buckets, of which this build keeps
    // [b_first, b_first + nbuckets); keys of other buckets are dropped in passes A and B.  A build of
    // the whole key space has nb_virtual == nbuckets and b_first == 0.
    u32 nb_virtual;
    u32 b_first;
};
struct KhTile { u32 seg; u32 tile_in_seg; };

// What one pass-C workgroup needs, in the order the workgroups are started (written by
// k_col_offsets).  Buckets of different segments are INTERLEAVED in that order and every
// segment has its own look-back chain and output region: a workgroup's chain predecessor was
// started (number of interleaved segments) workgroups earlier, so its aggregate is almost
// always published by the time it is looked at (the single chain of before spent a quarter of a
// workgroup's life waiting for the slowest of its 512 concurrent predecessors).
struct KhBucketWork {
    u64 lo;         // first key of the bucket in the partition array
    u64 out_base;   // first output record of the bucket's segment
    u32 n;          // keys in the bucket
    u32 nb;         // equal-width buckets of the whole key space (the fine-bin scale; KhSeg::nb_virtual)
    u32 b;          // index of the bucket inside its segment = index in the segment's chain
    u32 gb;         // global bucket index (segment's chain starts at gb - b)
};

struct KhSetView {    // one operand of a set operation (device-resident, sorted by mixed key)
    const void* keys;
    const u32* counts;   // nullptr => every counter == uniform
    u64 n;
    u32 uniform;
    u32 pad;
};

struct KhBoundsJob {   // one set operation's share of a batched range-bounds launch
    const KhSetView* sets;
    u64* bounds;
    u64* zero;          // workspace to clear (look-back descriptors, control words, histogram)
    u64 zero_words;
    u32 nsets, nranges;
};

// Several independent set operations of the same kind (W, payload mode, operation, cs) run as ONE
// launch: blockIdx.y picks the operation from this table, which travels in the kernel arguments
// (scalar loads, no extra global round trip in front of a workgroup's first useful load).
constexpr int KH_SETOP_BATCH = 16;
struct KhSetopJob {
    const KhSetView* sets;
    const u64* bounds;          // [nsets][nslots + 1] of the whole operation
    void* out_keys;
    u32* out_counts;
    u64* desc;                  // this chain's look-back descriptors [nranges]
    u32* ctl;                   // the operation's control words: ticket, err, fullest slot, pad, u64 outputs
    unsigned long long* hist;   // nullptr: no fused histogram
    u32 nsets;
    u32 nslots;                 // slots of the whole operation (the key -> slot scale)
    u32 slot0, nranges;         // this chain covers slots [slot0, slot0 + nranges)
};
// An operation whose output set nobody reads as ONE array (histogram-only unions) is run as several
// chains: consecutive slot ranges with their own look-back chain, each writing at the offset its
// inputs start at (sum over operands of bounds[slot0]) — same bytes written, no global order to
// wait for.  A chain adds its output count to ctl's u64 when its last slot finishes.
struct KhSetopBatch { KhSetopJob job[KH_SETOP_BATCH]; };

// ---- fused experiment-type-1 path (kh_exp1_run when no per-group set is wanted) ----
// Pass C in GRID mode: every genome of the batch uses the SAME bucket grid (nb equal-width slots
// of the mixed key space), a bucket's distinct keys are written where its input starts (index
// `lo` of an output array laid out like the partition array: no compaction, hence no look-back
// chain), and for every bucket an index of S + 1 u16 offsets is emitted: off[f] = number of the
// bucket's keys in front of sub-range f (S equal-width sub-ranges of the bucket), off[S] = d.
// Slot r = b * S + f of the key space then is, in EVERY genome g, the contiguous record range
// [bstart[g*nb+b] + off[f], bstart[g*nb+b] + off[f+1]) — what a set operation otherwise finds by
// binary search (k_range_bounds) is a table lookup.
struct KhGrid {
    u16* off;                       // [nseg * nb][S + 1]; nullptr: normal (compact, look-back) mode
    unsigned long long* distinct;   // [nseg]: distinct keys per genome, added bucket by bucket
    u32 S;
    u32 nb;
};
// One tagged n-ary union over the gridded genome sets of up to 64 genomes: operand g carries tag
// g, a run of equal keys becomes a 64-bit genome mask, and from the mask come, in ONE pass,
//   * for every group holding the key, "in how many of its genomes" -> step_4 histograms,
//   * "in how many groups"                                          -> step_8 histogram
// (exp_type_1.smk:175-191, :243-259).  Histograms are compact: bin = first bin of the group +
// count; across-group bins follow at `abase`.  A workgroup (one slot) keeps its histogram in LDS and
// adds it to one of `reps` replicas in device memory (the host sums them): the replicas spread
// what would otherwise be ~10^6 atomics per step on a handful of cache lines.
constexpr int KH_TAG_MAX_OPS = 64;
constexpr int KH_TAG_MAX_BINS = 255;
struct KhTagJob {
    const void* keys;           // gapped key arrays of all operands (one allocation)
    const u64* bstart;          // [nops * nb + 1]
    const u16* off;             // [nops * nb][S + 1]
    const u32* ginfo;           // [64] per operand: first operand of its group | group size << 8 | first bin << 16
    unsigned long long* hist;   // [reps][nbins]
    u32* ctl;                   // [0] error bits, [1] fullest slot seen
    u32 nb, S, nops, nbins, abase, ngroups, reps;
    u32 nbv;                    // buckets of the whole key space (== nb unless the build was a key-range wave)
    u32 binmul;                 // 2^26 / (fine bins of the widest sub-range): in-slot fine-bin scale
    // optional ordered output (multi-GPU: the local across-group set, counter = groups holding the key)
    void* out_keys;
    u32* out_counts;
    u64* desc;                  // look-back descriptors [nb * S], zeroed; nullptr: histograms only
    unsigned long long* out_n;  // number of records written
};

// ---- super-k-mer form of the fused path (kh_skm.hip): records of n consecutive k-mers that share their
// minimizer slot, partitioned in two levels (coarse bucket, then slot = coarse * S + fine), then one LDS
// hash set per slot.  A record is 16 bytes: bits [0, 2(n+k-1)) the bases (base j at bits 2j, A0 C1 G2 T3),
// bits 108..116 the fine index of its slot, 117..122 the genome (operand) number, 123..127 n.
#ifndef KH_TUNE_SKM_STAGE
#define KH_TUNE_SKM_STAGE 2048   // (1024 / 1536 / 2048 / 2304 / 2560: scatter 0.82 / 0.68 / 0.645 / 0.69 / 0.735 ms; above 2048 only two workgroups fit a CU)
#endif
constexpr u32 KH_SKM_STAGE = KH_TUNE_SKM_STAGE;   // records counting-sorted in LDS per flush of the scatter
constexpr u32 KH_SKM_MAX_COARSE = 256;    // coarse buckets (LDS counters of the scatter); 512 (KH_SKM2_MAX_COARSE) for inputs too large for 256
constexpr u32 KH_SKM_MAX_FINE = 512;      // slots per coarse bucket (9 bits in the record)
constexpr u32 KH_SKM_MAX_CAP2 = 1024;     // records of one slot with one-word keys: one per thread of the union (512 with the 2048-entry table)
constexpr int KH_SKM_MIN_K = 17, KH_SKM_MAX_K = 32;   // (below 17 the key arrays are faster; the kernels take k >= 15)
#ifndef KH_TUNE_SKM_CUR1_STRIDE
#define KH_TUNE_SKM_CUR1_STRIDE 1088
#endif
// u32 words between the cursors of two coarse buckets: every flush of every workgroup adds to all of them,
// so they are kept on different memory channels (4352 bytes apart) instead of sixteen to a cache line
constexpr u32 KH_SKM_CUR1_STRIDE = KH_TUNE_SKM_CUR1_STRIDE;
struct KhSkmJob {
    const KhSeg* segs;
    const KhTile* tiles;
    const u8* seg_tag;              // [nseg] tag of a segment's records; nullptr: the segment number (operand = genome)
    uint4* reg1;                    // [nb1][cap1] records by coarse bucket
    uint4* reg2;                    // [nslots][cap2] records by slot
    u32* cur1;                      // [nb1 * KH_SKM_CUR1_STRIDE] zeroed: cursor of bucket b at b * KH_SKM_CUR1_STRIDE
    u32* cur2;                      // [nslots] zeroed
    unsigned long long* inst;       // [nops] valid k-mer instances per genome
    unsigned long long* dup;        // [64] instances whose (k-mer, genome) pair was seen before
    const u32* ginfo;               // as KhTagJob
    unsigned long long* hist;       // [reps][nbins]
    u32* ctl;                       // [0] error bits, [1] fullest slot seen, [2] records written by the scatter
    u32 tile_pos;
    int k, m;                       // m-mer length of the minimizer (<= 16)
    u32 w;                          // m-mers per k-mer: k - m + 1
    u32 nmax;                       // k-mers per record at most
    u32 nslots, S, nb1, cap1, cap2;
    u32 nbins, abase, reps, nops;
    u32 table;                      // entries of the union's hash set: 4096 (1024 threads) or 2048 (512 threads); one-word keys
    // one-word keys: records of slots whose region is full (regroup) and the slots the union leaves to k_skm_big
    uint4* spill_rec;               // [spill_cap] records (one or two uint4 each); nullptr: a full region is an error
    u32* spill_slot;                // [spill_cap]
    u32* big_list;                  // [big_cap] slots with more records than their region holds
    u32 spill_cap, big_cap;         // counters: ctl[5] records spilled, ctl[6] slots listed
};
bool kh_skm_supports_w(u32 w);   // m-mers per k-mer the scatter kernel is instantiated for
size_t kh_skm_scatter_lds_bytes(u32 nb1);
size_t kh_skm_regroup_lds_bytes(u32 S);
size_t kh_skm_union_lds_bytes(u32 table);
u32 kh_skm_union_max_cap2(u32 table);   // records of a slot the union with that table takes
void kh_launch_skm_scatter(const KhSkmJob& job, u32 ntiles, hipStream_t st);
void kh_launch_skm_regroup(const KhSkmJob& job, hipStream_t st);
void kh_launch_skm_union(const KhSkmJob& job, u32 cs, u32 grid, hipStream_t st);   // persistent: grid workgroups walk the slots
u32 kh_skm_union_per_cu(u32 table);   // workgroups of the union that fit a CU
void kh_launch_skm_big(const KhSkmJob& job, u32 cs, u32 nbig, hipStream_t st);   // the slots of big_list, one workgroup each
// ---- the exchange form of the across-group step (multi-GPU; kh_skm.hip k_skm_pack / k_skm_phased)
struct KhSkmPackJob {
    const uint4* reg2;              // [nslots][cap2] this rank's records by slot (tag = local group, < 32)
    const u32* cur2;                // [nslots]
    uint4* out_rec;                 // [nparts][part_cap] the records that travel, by owner of their slot
    u32* out_mask;                  // [nparts][part_cap] their masks of local groups
    u32* part_cursor;               // [nparts] zeroed
    u32* slot_count;                // [nslots] records that travel
    u32* slot_off;                  // [nslots] where they start in their part's array
    u32* ctl;                       // [0] error bits
    unsigned long long* dup;        // [32] or null: per tag, k-mer instances of records that repeat one of the SAME tag
    u64 part_cap;
    u32 cap2, nslots, spp;          // spp: slots per part (slot s belongs to part s / spp)
    u32 nsub;                       // cursors per part (slot s uses cursor s % nsub over 1 / nsub of the part's array): 1 = the
                                    // part is filled without gaps (it travels); more = no single hot atomic (one GPU)
};
struct KhSkmCompactJob {            // a part packed through several cursors -> the same records without gaps (they travel)
    const uint4* tmp_rec;           // [nparts][part_cap], sub-range q of a part at q * (part_cap / nsub)
    const u32* tmp_mask;
    uint4* out_rec;                 // [nparts][part_cap], filled from 0
    u32* out_mask;
    const u32* cursors;             // [nparts][nsub] records per sub-range
    u32* slot_off;                  // [nslots] rewritten to the compacted positions
    u32* part_n;                    // [nparts] out: records of the part
    u64 part_cap;
    u32 nslots, spp, nsub, nparts;
};
void kh_launch_skm_pack_compact(const KhSkmCompactJob& job, hipStream_t st);
struct KhSkmPiece {                 // what one source rank sent for this rank's slots
    const uint4* rec;
    const u32* mask;
    const u32* count;               // [nslots of this rank]
    const u32* off;                 // [nslots of this rank] first record of the slot in `rec`
    u32 dup_row;                    // row of KhSkmPhasedJob::dup its tags count into
    u32 join_next;                  // 1: the next piece continues this phase (the same tags: no fold in between)
};
struct KhSkmPhasedJob {
    const KhSkmPiece* pieces;       // device array [npieces]
    unsigned long long* hist;       // [hist_len], zeroed
    u32* ctl;                       // [0] error bits
    unsigned long long* dup;        // [npieces][32] or null: per (piece, tag), instances that repeat a k-mer of the same tag
    u32 npieces, nslots, hist_len, cs;
    u32 share_q8;                   // distinct k-mers expected per 256 instances of a slot (256: the pieces share nothing;
                                    // sub-batches of one group share most): the rounds of a slot are sized from it
    int k;
};
constexpr u32 KH_SKM_PHASED_MAX_DUP_PIECES = 32;   // pieces whose repeats can be counted (LDS counters)
size_t kh_skm_pack_lds_bytes();
size_t kh_skm_phased_lds_bytes();
void kh_launch_skm_pack(const KhSkmPackJob& job, hipStream_t st);
void kh_launch_skm_phased(const KhSkmPhasedJob& job, u32 grid, hipStream_t st);
// the same three steps for two-word keys (kh_skm2.hip): 32-byte records (two uint4 per record in reg1 / reg2)
constexpr int KH_SKM2_MAX_K = 63;         // k = 64: the all-ones low key word is a k-mer (A^32 T^32)
constexpr u32 KH_SKM2_MAX_COARSE = 512;
constexpr u32 KH_SKM2_MAX_FINE = 1024;    // slots per coarse bucket with two-word keys (10 bits in the record)
bool kh_skm2_supports_w(u32 w);
u32 kh_skm2_max_cap2();
u32 kh_skm2_table();
u32 kh_skm2_union_per_cu();   // workgroups of the two-word union that fit a CU
void kh_launch_skm2_big(const KhSkmJob& job, u32 cs, u32 nbig, hipStream_t st);
size_t kh_skm2_scatter_lds_bytes(u32 nb1);
size_t kh_skm2_regroup_lds_bytes(u32 S);
size_t kh_skm2_union_lds_bytes(u32 nbins);
void kh_launch_skm2_scatter(const KhSkmJob& job, u32 ntiles, hipStream_t st);
void kh_launch_skm2_regroup(const KhSkmJob& job, hipStream_t st);
void kh_launch_skm2_union(const KhSkmJob& job, u32 cs, u32 grid, hipStream_t st);   // persistent, as the one-word union

struct KhLookback {      // workspace of one ordered single-pass launch
    u64* desc;           // [nparts] tile descriptors, zeroed before launch
    u32* ticket;         // zeroed before launch
    u32* err;            // sticky error bits
    // Part order.  0: part = blockIdx.x, relying on workgroups being started in index order
    // (what the hardware does; saves a device-wide atomic and its round trip at every
    // workgroup start).  1: parts are handed out by an atomic ticket, which needs no such
    // assumption.  The host starts with 0 and switches a context to 1 for good if a look-back
    // ever times out (KH_ERR_SPIN_TIMEOUT), re-running the launch.
    u32 dynamic;
};
enum : u32 { KH_ERR_SPIN_TIMEOUT = 1u, KH_ERR_CAPACITY = 2u, KH_ERR_ORDER = 4u };

size_t kh_extract_lds_bytes(u32 nb_alloc);
size_t kh_sort_lds_bytes(int W, u32 cap, bool pay);

void kh_launch_extract(int W, bool scatter, const u8* seq, const KhSeg* segs, const KhTile* tiles,
                       u32 ntiles, u32 nb_alloc, int k, u32* thist, const u64* bstart, void* part,
                       u32 tile_pos, hipStream_t st);   // tile_pos: k-mer positions per workgroup, a multiple of 16384
void kh_launch_col_totals(const KhSeg* segs, u32 nseg, u32 max_nb, const u32* thist, u64* tot,
                          hipStream_t st);
// over / over_cap: grid mode only (else nullptr / 0): buckets of more than over_cap keys are listed in over[1..], over[0] = count
void kh_launch_col_offsets(const KhSeg* segs, u32 nseg, u32 max_nb, u32* thist, const u64* bstart,
                           const u32* rank, const u64* seg_out_base, KhBucketWork* work, u32* over, u32 over_cap,
                           hipStream_t st);
size_t kh_exscan_tmp_words(u64 n);
void kh_launch_exscan(const u64* in, u64* out, u64 n, u64* tmp, hipStream_t st);   // out has n+1 entries
void kh_launch_bucket_sort(int W, const void* part, const KhBucketWork* work,
                           u32 nbuckets, int k,
                           void* out_keys, u32* out_counts, KhLookback lb, u32 ci, u32 cx, u32 cs,
                           hipStream_t st);
// pass C in grid mode: k_grid_bucket over all buckets (three workgroups per CU; buckets above the LDS
// capacity skipped) + k_grid_oversize over the list k_col_offsets made of those
u32 kh_grid_bucket_capacity(int W);
void kh_launch_grid_bucket(int W, const void* part, const KhBucketWork* work, u32 nbuckets, int k, void* out_keys,
                           const u32* over, u32* err, const KhGrid& grid, hipStream_t st);
size_t kh_tag_lds_bytes(int W, u32 cap, u32 nbins, bool emit);
// one-word keys, nothing emitted: the hash-set form (k_union_hash), one workgroup per slot
u32 kh_union_hash_capacity();
void kh_launch_union_hash(const KhTagJob& job, u32 grid, int k, u32 cs, hipStream_t st);
void kh_launch_union_tagged(int W, const KhTagJob& job, u32 grid, int k, u32 cs, hipStream_t st);
void kh_launch_range_bounds(int W, const KhSetView* sets, u32 nsets, u32 nranges, int k,
                            u64* bounds, u64* zero, u64 zero_words, hipStream_t st);
void kh_launch_range_bounds_batch(int W, const KhBoundsJob* jobs, u32 njobs, u64 max_threads, int k,
                                  hipStream_t st);
void kh_launch_setop(int W, bool pay, u32 cap, const KhSetopBatch& batch, u32 njobs, int k, int op, int mode,
                     u32 cs, u32 hist_len, bool dynamic_order, hipStream_t st);
void kh_launch_histogram(const u32* counts, u64 n, unsigned long long* hist, u32 hist_len,
                         hipStream_t st);
void kh_launch_unmix(int W, const void* in, void* out, u64 n, int k, hipStream_t st);
void kh_launch_fill_u32(u32* p, u64 n, u32 v, hipStream_t st);
void kh_launch_membership(int W, const void* pivot, u64 n, const KhSetView* sets, u32 nsets, int k,
                          u32 nwords, u64* masks, hipStream_t st);
void kh_launch_table_add(const void* keys, u64 n, int k, void* table, u32 cell_bytes, hipStream_t st);
void kh_launch_table_hist(const void* table, u32 cell_bytes, u64 lo, u64 hi, u32 cs,
                          unsigned long long* hist, u32 hist_len, hipStream_t st);
