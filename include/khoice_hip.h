/* khoice_hip.h — C ABI of libkhoice_hip.so, the MI355X (gfx950) k-mer engine.
 *
 * The reference (vshiv18/khoice) has no FFI: its boundary to the k-mer engine is
 * process + argv + files, i.e. Snakemake `shell:` calls to KMC 3.2.1's `kmc` and
 * `kmc_tools`.  Each entry point below is what a binding for one of those call forms
 * would bind; the reference call site it replaces is cited as file:line into
 * /root/reference.  The `kmc` / `kmc_tools` executables shipped in bin/ are thin argv
 * parsers over exactly these functions.
 *
 * Conventions
 *   - plain C types only; handles are opaque pointers owned by the library
 *   - every function returns 0 on success or a negative KH_E_* code; the message of the
 *     last failure on the calling thread is kh_last_error()
 *   - a kh_ctx owns one HIP stream on one device; calls on one ctx must not overlap,
 *     different ctxs may be used from different threads / processes concurrently
 *   - k-mers are 2k-bit integers, first base most significant, A=0 C=1 G=2 T=3;
 *     host-side key arrays use W = ceil(2k/64) little-endian 64-bit words per key
 *   - there is NO CPU fallback: without a usable HIP device every call fails
 */
#ifndef KHOICE_HIP_H
#define KHOICE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kh_ctx kh_ctx;
typedef struct kh_set kh_set;

enum {
    KH_OK = 0,
    KH_E_ARG = -1,      /* bad argument */
    KH_E_HIP = -2,      /* HIP runtime error */
    KH_E_IO = -3,       /* file error */
    KH_E_FORMAT = -4,   /* not a khoice_amd database */
    KH_E_KMISMATCH = -5,/* operands built with different k */
    KH_E_CAPACITY = -6, /* a bucket could not be fitted after retries */
    KH_E_INTERNAL = -7,
    KH_E_NOMEM = -8
};

/* set operations of `kmc_tools simple` / `complex` */
enum { KH_UNION = 0, KH_INTERSECT = 1, KH_KMERS_SUBTRACT = 2, KH_COUNTERS_SUBTRACT = 3 };
/* counter calculation modes, `-oc<mode>` */
enum { KH_MODE_MIN = 0, KH_MODE_MAX = 1, KH_MODE_SUM = 2, KH_MODE_DIFF = 3,
       KH_MODE_LEFT = 4, KH_MODE_RIGHT = 5 };

#define KH_NO_MAX 0xffffffffu          /* cx: no upper cut-off */
#define KH_KMC_DEFAULT_CS 255u         /* kmc / kmc_tools default counter saturation */

/* ---------------------------------------------------------------- context */
int kh_ctx_create(int device, kh_ctx **out);
void kh_ctx_destroy(kh_ctx *ctx);
const char *kh_last_error(void);
int kh_device_count(void);
/* JSON: device, op counts, bytes/keys processed, per-kernel-class time when profiling */
int kh_stats(kh_ctx *ctx, char *buf, size_t buflen);
/* record HIP events around every kernel class (adds a sync when stats are read) */
int kh_profile_enable(kh_ctx *ctx, int on);
int kh_stats_reset(kh_ctx *ctx);
int kh_sync(kh_ctx *ctx);
/* release cached device allocations */
int kh_trim(kh_ctx *ctx);

/* ---------------------------------------------------------------- K1: build
 * `kmc -fm -m64 -k{k} -ci1 IN.fna.gz OUT tmp/`   workflow/rules/exp_type_1.smk:163
 * (also exp_type_2.smk:297,306; exp_type_3.smk:183,192; exp_type_4.smk:143,152;
 * exp_type_6.smk:171,181).  Canonical counting; symbols other than ACGTacgt break a run;
 * keep ci <= count <= cx; counters saturate at cs.
 *
 * kh_build_batch: nseq sequences in ONE launch sequence.  seqs[i] points at len[i] bytes
 * of cleaned sequence text (records separated by any non-ACGT byte, e.g. '\n'); host
 * pointers when on_device == 0, device pointers (any alignment) when on_device != 0.
 * with_counts == 0 builds plain sets (every counter 1): the fused form of
 * `kmc ...` followed by `kmc_tools transform ... set_counts 1` (exp_type_1.smk:163+173).
 */
int kh_build_batch(kh_ctx *ctx, int nseq, const uint8_t *const *seqs, const uint64_t *lens,
                   int on_device, int k, uint32_t ci, uint32_t cx, uint32_t cs, int with_counts,
                   kh_set **out_sets);
/* one (gz) multi-FASTA file -> one set; host ingest (inflate + record parsing) included */
int kh_build_fasta(kh_ctx *ctx, const char *path, int k, uint32_t ci, uint32_t cx, uint32_t cs,
                   kh_set **out);
/* ingest only: cleaned sequence text of a (gz) multi-FASTA; caller frees with kh_free_host */
int kh_read_fasta(const char *path, uint8_t **seq, uint64_t *len);
void kh_free_host(void *p);

/* Batched ingest (the inputs the dataset_N/GENOME.fna.gz files of exp_type_1.smk:44-47,158):
 * nfiles (gz) multi-FASTA files -> cleaned sequence text RESIDENT IN DEVICE MEMORY, exactly the
 * bytes kh_read_fasta returns.  nthreads host threads inflate the files (0 = one per
 * core, at most 32); files are shipped and cleaned on the device as they complete, overlapping
 * the inflation of the others.  The texts are read in place by kh_build_batch / kh_exp1_run
 * (on_device = 1): kh_seqs_get gives pointer and length of text i; kh_seqs_free releases them. */
typedef struct kh_seqs kh_seqs;
int kh_ingest_fasta(kh_ctx *ctx, int nfiles, const char *const *paths, int nthreads, kh_seqs **out);
int kh_seqs_count(const kh_seqs *seqs);
int kh_seqs_get(const kh_seqs *seqs, int i, const uint8_t **dev_ptr, uint64_t *len);
void kh_seqs_free(kh_seqs *seqs);

/* ---------------------------------------------------------------- K2: set_counts
 * `kmc_tools transform IN set_counts {v} OUT`     exp_type_1.smk:173,241
 * O(1): the result shares IN's key storage and carries a uniform counter. */
int kh_set_counts(kh_ctx *ctx, const kh_set *in, uint32_t value, kh_set **out);

/* ---------------------------------------------------------------- K3: complex union
 * `kmc_tools complex OPS.txt` with  out = (set1 + set2 + ... ), OUTPUT_PARAMS -cs{cs}
 * ops file written at exp_type_1.smk:52-61,75-84; call sites :182,:250.
 * Counters add and saturate at cs.  hist (optional, may be NULL) receives the counter
 * histogram of the result fused into the same pass (K4): hist[c] for c in [0,hist_len),
 * counters >= hist_len fall into the last bin. */
int kh_union_sum(kh_ctx *ctx, const kh_set *const *sets, int nsets, uint32_t cs, kh_set **out,
                 uint64_t *hist, uint32_t hist_len);

/* The fused histogram alone (exp_type_1.smk:243-259 when the caller keeps no step_7 database;
 * the merged slices of the multi-GPU exchange): same counters as kh_union_sum, no set handle. */
int kh_union_histogram(kh_ctx *ctx, const kh_set *const *sets, int nsets, uint32_t cs,
                       uint64_t *hist, uint32_t hist_len);

/* ---------------------------------------------------------------- K5/K6: simple
 * `kmc_tools simple A B intersect OUT -ocsum`     exp_type_2.smk:363-365,479-481
 * `kmc_tools simple A B kmers_subtract OUT`       exp_type_2.smk:377-379,493-495
 * op: KH_UNION | KH_INTERSECT | KH_KMERS_SUBTRACT | KH_COUNTERS_SUBTRACT */
int kh_simple(kh_ctx *ctx, const kh_set *a, const kh_set *b, int op, int mode, uint32_t cs,
              kh_set **out);

/* ---------------------------------------------------------------- K4: histogram
 * `kmc_tools transform IN histogram OUT.txt`      exp_type_1.smk:191,259
 * hist[c], c in [0, hist_len); counters >= hist_len are added to the last bin. */
int kh_histogram(kh_ctx *ctx, const kh_set *set, uint64_t *hist, uint32_t hist_len);
/* Small-k form of the across-group occurrence count (exp_type_1.smk:243-259 for k <= 16): a
 * direct-addressed table of 4^k cells in DEVICE memory (cell_bytes 1 or 4, caller-owned, zeroed by
 * the caller); kh_table_add_set adds 1 (saturating) to the cell of every k-mer of `set`, so after
 * adding each group set once, cell v = number of groups holding canonical k-mer v.  Tables of
 * several GPUs are summed by the caller (RCCL all-reduce).  kh_table_histogram:
 * hist[min(cell, cs, hist_len-1)] += 1 over the non-zero cells of [lo, hi) (cs 0 = no cap);
 * lo*cell_bytes must be a multiple of 16.  Work is queued on the context's stream. */
int kh_table_add_set(kh_ctx *ctx, const kh_set *set, void *d_table, uint32_t cell_bytes);
int kh_table_histogram(kh_ctx *ctx, const void *d_table, uint32_t cell_bytes, uint64_t lo, uint64_t hi,
                       uint32_t cs, uint64_t *hist, uint32_t hist_len);
/* ---------------------------------------------------------------- experiment type 4
 * Replaces the text dumps + Python dict of src/merge_lists.py:14-33 (build_dictionary /
 * update_dictionary; rules exp_type_4.smk:247-294): for every k-mer of `pivot` (with its count),
 * which of `sets` hold it.  Outputs are HOST arrays in `dump -s` order (ascending canonical
 * key): keys_out[n*W] (may be NULL), counts_out[n] (may be NULL), masks_out[n*nwords] with
 * nwords = max(1, (nsets+63)/64), bit d%64 of word d/64 = "sets[d] holds the k-mer". */
int kh_membership(kh_ctx *ctx, const kh_set *pivot, const kh_set *const *sets, int nsets,
                  uint64_t *keys_out, uint32_t *counts_out, uint64_t *masks_out);
/* One feature-level confusion-matrix row, summed in the reference's order and arithmetic
 * (src/merge_lists.py:122-141): row[d] (d < nsets) = sum over pivot k-mers held by the sets M
 * of 1/len(M)*count for d in M; *unique_pivot_count = sum of counts of k-mers no set holds. */
int kh_confusion_row(kh_ctx *ctx, const kh_set *pivot, const kh_set *const *sets, int nsets,
                     double *row, uint64_t *unique_pivot_count);

/* text form consumed at exp_type_1.smk:210-212: lines "c<TAB>n", c = 1..cmax */
int kh_histogram_file(kh_ctx *ctx, const kh_set *set, uint32_t cmax, const char *path);

/* the same text from a histogram array in host memory (what kh_exp1_run returns): counters past
 * hist_len read 0 */
int kh_write_histogram_text(const char *path, const uint64_t *hist, uint32_t hist_len, uint32_t cmax);

/* ---------------------------------------------------------------- K7: sorted dump
 * `kmc_tools transform IN dump -s OUT.txt`        exp_type_4.smk:255-257,268-270
 * lines "KMER<TAB>count", sorted A<C<G<T (consumer src/merge_lists.py:19-22). */
int kh_dump_sorted(kh_ctx *ctx, const kh_set *set, const char *path);

/* ---------------------------------------------------------------- set handles */
void kh_set_free(kh_set *set);
int kh_set_info(const kh_set *set, uint64_t *n, int *k, int *words_per_key, int *has_counts,
                uint32_t *uniform_count);
/* saturation value the set's counters were produced with (kmc -cs, OUTPUT_PARAMS -cs, or the
 * kmc_tools default 255); `kmc_tools transform histogram` prints 2^(8*bytes(counter_max))-1 lines */
int kh_set_counter_max(const kh_set *set, uint32_t *counter_max);
/* copy to host in storage order (NOT sorted by key): keys un-mixed, n*W words; counts n */
int kh_set_download(kh_ctx *ctx, const kh_set *set, uint64_t *keys, uint32_t *counts);
/* build a set from host arrays of DISTINCT keys in any order (counts may be NULL => 1) */
int kh_set_upload(kh_ctx *ctx, int k, uint64_t n, const uint64_t *keys, const uint32_t *counts,
                  kh_set **out);
/* raw device views for zero-copy exchange (multi-GPU): mixed keys sorted ascending,
 * counts NULL when uniform.  Valid until the set is freed. */
int kh_set_device_ptrs(const kh_set *set, const void **keys_mixed, const uint32_t **counts);
/* copy the raw storage (mixed keys in storage order; counters, materialised when uniform) into
 * caller-owned DEVICE buffers of n*W*8 and n*4 bytes: the send side of the exchange */
int kh_set_export_device(kh_ctx *ctx, const kh_set *set, void *keys_out, uint32_t *counts_out);
/* the same for elements [lo, hi); stream-ordered on the ctx stream (kh_sync before use elsewhere) */
int kh_set_export_range(kh_ctx *ctx, const kh_set *set, uint64_t lo, uint64_t hi, void *keys_out,
                        uint32_t *counts_out);
/* zero-copy handle over caller-owned device arrays (mixed keys ascending and distinct; counts may
 * be NULL => every counter == uniform): the receive side of the exchange.  The arrays must stay
 * alive and unchanged until kh_set_free. */
int kh_set_wrap_device(kh_ctx *ctx, int k, uint64_t n, const void *keys_mixed, const uint32_t *counts,
                       uint32_t uniform, kh_set **out);
/* wrap device arrays of already mixed, sorted, distinct keys (copied into the library) */
int kh_set_from_device(kh_ctx *ctx, int k, uint64_t n, const void *keys_mixed,
                       const uint32_t *counts, kh_set **out);
/* index of the first key of every one of `nparts` equal-width slots of the mixed key
 * space: bounds[nparts+1] (host).  Used to slice a set for the all-to-all exchange. */
int kh_set_partition_bounds(kh_ctx *ctx, const kh_set *set, uint32_t nparts, uint64_t *bounds);
/* the same for nsets sets in one launch: bounds[nsets][nparts+1] */
int kh_sets_partition_bounds(kh_ctx *ctx, const kh_set *const *sets, int nsets, uint32_t nparts,
                             uint64_t *bounds);

/* ---------------------------------------------------------------- database files
 * `<prefix>.kmc_pre` + `<prefix>.kmc_suf` (names required by the Snakemake rules,
 * exp_type_1.smk:160-161); khoice_amd container format, written atomically. */
int kh_save(kh_ctx *ctx, const kh_set *set, const char *prefix);
int kh_load(kh_ctx *ctx, const char *prefix, kh_set **out);

/* ---------------------------------------------------------------- fused experiment type 1
 * The whole device side of exp_type_1.smk:156-259 for one k on resident sequences:
 * per genome build+set (steps 1-2), per group union-sum + histogram (steps 3-4),
 * group sets (step 6), across-group union-sum + histogram (steps 7-8).
 *   group_of[i]    group index (0-based) of sequence i; ngroups groups
 *   within_hist    [ngroups * hist_len]  step_4 histograms
 *   across_hist    [hist_len]            step_8 histogram
 *   distinct_per_seq [nseq]              distinct canonical k-mers of each genome
 *   group_sets     optional [ngroups]: the step_3 group unions (caller frees)
 *   across_set     optional: the step_7 union (caller frees)
 * When across_hist and across_set are both NULL steps 7-8 are skipped (multi-GPU callers run
 * them after the exchange of the group sets).
 */
int kh_exp1_run(kh_ctx *ctx, int nseq, const uint8_t *const *seqs, const uint64_t *lens,
                int on_device, const int *group_of, int ngroups, int k, uint32_t cs,
                uint64_t *within_hist, uint64_t *across_hist, uint32_t hist_len,
                uint64_t *distinct_per_seq, kh_set **group_sets, kh_set **across_set);

/* ---------------------------------------------------------------- multi-GPU exchange (RCCL over xGMI)
 * Steps 7-8 of exp_type_1.smk:243-259 when the groups are sharded over several GPUs, one process
 * per GPU: every rank runs kh_exp1_run on its own groups asking for `across_set` (counter = number
 * of its groups holding the k-mer) and calls kh_across_exchange_histogram; the mixed key space is
 * cut into nranks slots, slices travel by grouped ncclSend / ncclRecv (full mesh), the owner sums
 * the counters (saturating at cs) and the histograms are all-reduced: every rank gets the global
 * step_8 histogram.  kh_comm_unique_id is called on one rank and its 128 bytes handed to the others
 * by whatever launched them (file, socket, MPI); RCCL is loaded on first use. */
#define KH_COMM_ID_BYTES 128
typedef struct kh_comm kh_comm;
int kh_comm_unique_id(char id[KH_COMM_ID_BYTES]);
int kh_comm_init(kh_ctx *ctx, int rank, int nranks, const char id[KH_COMM_ID_BYTES], kh_comm **out);
void kh_comm_destroy(kh_comm *comm);
int kh_across_exchange_histogram(kh_ctx *ctx, kh_comm *comm, const kh_set *local_across_set, uint32_t cs,
                                 uint64_t *hist, uint32_t hist_len);

/* host-side helpers exposed for tests (no device work) */
void kh_mix_host(int k, const uint64_t *key_words, uint64_t *out_words);
void kh_unmix_host(int k, const uint64_t *key_words, uint64_t *out_words);

/* ---------------------------------------------------------------- steps 7-8 across ranks, exchange of records
 * (workflow/rules/exp_type_1.smk:243-259 when the groups live on several GPUs; 17 <= k <= 32).
 * kh_skm_exchange_plan: the slot geometry all ranks must share, from numbers they agreed on (the largest rank's number
 *   of k-mer positions, the largest group of any rank: its genomes bring their copies of a locus into a slot together,
 *   which sizes the slot's region).
 * kh_skm_pack: this rank's genomes -> minimizer records tagged with the LOCAL group number tag_of[i] (0..31), identical
 *   records merged, packed by owner of their slot into CALLER-ALLOCATED device buffers: rec_out [nparts][part_cap] x 16
 *   bytes, mask_out [nparts][part_cap], count_out / off_out [nparts][slots_per_part] (records of a slot and where they
 *   start in their part); part_n[p] (host) = records for part p.
 * kh_skm_phased_histogram: on the owner, the pieces received (one per source rank, device pointers) -> hist[c] = number of
 *   distinct k-mers of this rank's slots that occur in c groups of all ranks (c saturating at cs). */
int kh_skm_exchange_plan(kh_ctx *ctx, int k, uint64_t positions_max, uint32_t fan_max, int nparts, uint32_t *nslots,
                         uint32_t *slots_per_part, uint64_t *part_cap);
int kh_skm_pack(kh_ctx *ctx, int nseq, const uint8_t *const *seqs, const uint64_t *lens, int on_device, const int *tag_of,
                int k, uint32_t nslots, int nparts, uint64_t part_cap, void *rec_out, uint32_t *mask_out,
                uint32_t *count_out, uint32_t *off_out, uint64_t *part_n);
int kh_skm_phased_histogram(kh_ctx *ctx, int k, int npieces, const void *const *recs, const uint32_t *const *masks,
                            const uint32_t *const *counts, const uint32_t *const *offs, uint32_t nslots, uint32_t cs,
                            uint64_t *hist, uint32_t hist_len);

#ifdef __cplusplus
}
#endif
#endif /* KHOICE_HIP_H */
