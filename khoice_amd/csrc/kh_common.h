// khoice_amd — shared host/device definitions for the k-mer engine.
//
// Key model
// ---------
// A k-mer of length k (1..64) is the integer whose base i (0 = leftmost) sits in bits
// [2(k-1-i)+1 : 2(k-1-i)], A=0 C=1 G=2 T=3, so integer order == lexicographic order
// (the order `kmc_tools transform dump -s` prints, src/merge_lists.py:19-22 consumer).
// W = ceil(2k/64) 64-bit words per key, little-endian word order (w[0] = low word).
//
// Every set the engine keeps in HBM stores MIXED keys: key' = mix_k(key), a bijection
// on 2k-bit integers, and is sorted by key'.  Mixing makes the top bits uniform whatever
// the genome's base composition, so fixed-width bucket / range splits are balanced and
// a bucket always fits LDS; because the map is a bijection the sorted-by-key' order is a
// consistent total order across sets, which is all union / intersect / subtract need.
// Keys are un-mixed on download and re-sorted only for the sorted text dump (K7).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KH_HD __host__ __device__ __forceinline__
#else
#define KH_HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;

template <int W> struct KmerKey;
template <> struct KmerKey<1> { u64 lo; };
template <> struct alignas(16) KmerKey<2> { u64 lo, hi; };

KH_HD bool key_lt(const KmerKey<1>& a, const KmerKey<1>& b) { return a.lo < b.lo; }
KH_HD bool key_eq(const KmerKey<1>& a, const KmerKey<1>& b) { return a.lo == b.lo; }
KH_HD bool key_lt(const KmerKey<2>& a, const KmerKey<2>& b) {
    return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo);
}
KH_HD bool key_eq(const KmerKey<2>& a, const KmerKey<2>& b) { return a.hi == b.hi && a.lo == b.lo; }

KH_HD u64 kh_mask(int nbits) { return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull); }

// ---------------------------------------------------------------- bijective mixing
constexpr u64 KH_C1 = 0xff51afd7ed558ccdull;   // odd multipliers (murmur3 / splitmix family)
constexpr u64 KH_C3 = 0x9e3779b97f4a7c15ull;

constexpr u64 kh_modinv(u64 c) {   // inverse of odd c modulo 2^64 (Newton)
    u64 x = c;
    for (int i = 0; i < 6; ++i) x *= 2 - c * x;
    return x;
}
constexpr u64 KH_C1_INV = kh_modinv(KH_C1);
static_assert(KH_C1 * KH_C1_INV == 1ull, "modinv");

// bijection on [0, 2^n), 2 <= n <= 64.  s >= n/2 so one xor-shift step is its own inverse.
// One odd multiply between two folds: the fold brings the high half into the low bits, the
// multiply spreads every input bit into the top bits (which choose bucket, slot and fine bin).
// A second multiply round bought nothing measurable for balance and cost 16 quarter-rate
// integer multiplies per k-mer in both extraction passes.
KH_HD u64 kh_mix64(u64 x, int n) {
    const u64 M = kh_mask(n);
    const int s = (n + 1) >> 1;
    x ^= x >> s;
    x = (x * KH_C1) & M;
    x ^= x >> s;
    return x;
}
KH_HD u64 kh_unmix64(u64 x, int n) {
    const u64 M = kh_mask(n);
    const int s = (n + 1) >> 1;
    x ^= x >> s;
    x = (x * KH_C1_INV) & M;
    x ^= x >> s;
    return x;
}
// 64-bit scrambler (need not be invertible: it is the round function of the W = 2 mix)
KH_HD u64 kh_round(u64 v, u64 c) {
    v ^= v >> 32;
    v *= c;
    v ^= v >> 29;
    return v;
}

KH_HD KmerKey<1> kh_mix(KmerKey<1> a, int k) { a.lo = kh_mix64(a.lo, 2 * k); return a; }
KH_HD KmerKey<1> kh_unmix(KmerKey<1> a, int k) { a.lo = kh_unmix64(a.lo, 2 * k); return a; }
// 64 < 2k <= 128: hi holds nh = 2k-64 bits.  The low word is mixed in place (a 64-bit
// bijection) and then scrambles the high word, whose bits lead the sort order: bijective
// because each step is undone with the other word unchanged.
KH_HD KmerKey<2> kh_mix(KmerKey<2> a, int k) {
    const u64 MH = kh_mask(2 * k - 64);
    a.lo = kh_mix64(a.lo, 64);
    a.hi ^= kh_round(a.lo, KH_C3) & MH;
    return a;
}
KH_HD KmerKey<2> kh_unmix(KmerKey<2> a, int k) {
    const u64 MH = kh_mask(2 * k - 64);
    a.hi ^= kh_round(a.lo, KH_C3) & MH;
    a.lo = kh_unmix64(a.lo, 64);
    return a;
}

// top 32 bits of the 2k-bit mixed key (left-aligned when 2k < 32): monotone in key'.
KH_HD u32 kh_top32(const KmerKey<1>& a, int k) {
    const int n = 2 * k;
    return n >= 32 ? (u32)(a.lo >> (n - 32)) : (u32)(a.lo << (32 - n));
}
KH_HD u32 kh_top32(const KmerKey<2>& a, int k) {
    const int nh = 2 * k - 64;   // 1..64
    if (nh >= 32) return (u32)(a.hi >> (nh - 32));
    return (u32)((a.hi << (32 - nh)) | (a.lo >> (32 + nh)));
}
// slot of a mixed key among `nslots` equal-width, order-preserving slots
template <int W> KH_HD u32 kh_slot(const KmerKey<W>& a, int k, u32 nslots) {
    return (u32)(((u64)kh_top32(a, k) * (u64)nslots) >> 32);
}

// ---------------------------------------------------------------- engine constants
constexpr int KH_SUBTILE = 8192;            // k-mer start positions per sub-tile (256 thr x 32)
constexpr int KH_SUBTILES_PER_TILE = 8;     // one workgroup walks 8 sub-tiles = 65536 positions
constexpr int KH_TILE = KH_SUBTILE * KH_SUBTILES_PER_TILE;
constexpr int KH_MAX_BUCKETS_PER_SEG = 16384;   // LDS cursor array limit (64 KiB)
#ifndef KH_TUNE_MEAN_W1
#define KH_TUNE_MEAN_W1 3700
#endif
#ifndef KH_TUNE_FINE_BITS
#define KH_TUNE_FINE_BITS 13
#endif
constexpr int KH_BUCKET_MEAN_W1 = KH_TUNE_MEAN_W1;     // target keys per bucket (typical P = 4096)
constexpr int KH_BUCKET_MEAN_W2 = 1750;   // 7 sigma below the 2048 capacity (1850 = 4.6 sigma re-planned one 250 Mbp batch in four)
// LDS sort capacity in keys.  Mixed keys make slot sizes Poisson-tight (mean 3400 -> sigma 58),
// so 4096 leaves 6.5 sigma over the 3700 mean (beyond: the oversize path / a re-plan); 8 keys per thread keep the unrolled per-thread arrays in <128 VGPRs.
constexpr int KH_SORT_CAP_W1 = 4096;
constexpr int KH_SORT_CAP_W2 = 2048;
constexpr int KH_SORT_CAP_PAY_W1 = 4096;    // capacity with a 32-bit payload per key
constexpr int KH_SORT_CAP_PAY_W2 = 2048;
constexpr int KH_SORT_THREADS = 512;
constexpr int KH_SORT_NW = KH_SORT_THREADS / 64;   // waves per sort workgroup
constexpr int KH_SORT_WAVES_PER_SIMD = 4;   // 2 workgroups of 8 waves per CU (3 per CU measured slower: spills)
constexpr int KH_FINE_BITS = KH_TUNE_FINE_BITS;            // fine bins of the in-LDS distribution sort
constexpr int KH_FINE_BINS = 1 << KH_FINE_BITS;
constexpr int KH_FINE_LIMIT = 64;           // fullest fine bin the in-bin repair accepts
constexpr int KH_WORKLIST = 1024;           // keys of out-of-order bins repaired per slot
constexpr int KH_TABLE_MAX_K = 16;          // direct-addressed occurrence table: 4^16 cells at most
constexpr int KH_MAX_INPUT_SETS = 128;      // fan-in of one set-operation launch
constexpr int KH_LHIST_BINS = 512;          // LDS histogram bins fused into set-ops
#ifndef KH_TUNE_HASH_ROUNDS
#define KH_TUNE_HASH_ROUNDS 3
#endif
constexpr int KH_HASH_ROUNDS = KH_TUNE_HASH_ROUNDS;   // probes in the main LDS hash table before a key moves to the second one

enum KhSetOp : int {
    KH_OP_UNION = 0,             // n-ary or binary union, counters combined by `mode`
    KH_OP_INTERSECT = 1,
    KH_OP_KMERS_SUBTRACT = 2,
    KH_OP_COUNTERS_SUBTRACT = 3,
};
enum KhCounterMode : int {       // kmc_tools -oc<mode>
    KH_OC_MIN = 0, KH_OC_MAX = 1, KH_OC_SUM = 2, KH_OC_DIFF = 3, KH_OC_LEFT = 4, KH_OC_RIGHT = 5,
};
