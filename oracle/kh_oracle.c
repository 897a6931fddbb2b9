/* kh_oracle.c — TEST INFRASTRUCTURE ONLY: plain-C CPU restatement of the k-mer hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (khoice_amd, libkhoice_hip.so) never links or calls it.
 *
 * What it restates: the arithmetic khoice delegates to KMC 3.2.1 (third-party, pinned at
 * workflow/envs/khoice_exps.yaml:97, NOT in /root/reference, NOT in this image), at the
 * call forms khoice uses:
 *   kmc -fm -k{k} -ci1            workflow/rules/exp_type_1.smk:163
 *   transform set_counts 1        exp_type_1.smk:173,241
 *   complex (set1 + ...) -cs5000  exp_type_1.smk:52-61,182,250
 *   transform histogram           exp_type_1.smk:191,259
 *   simple intersect -ocsum       exp_type_2.smk:363-365
 *   simple kmers_subtract         exp_type_2.smk:377-379
 * k-mer windowing and canonical form follow src/merge_lists.py:53-58 and :60-73.
 *
 * PARITY STATUS: canonical form is pinned by tests/golden/canonical_kmers.json (captured
 * from the reference's src/merge_lists.py) through the Python oracle this file is checked
 * against; counting / set semantics are PARITY UNPINNED (no reference fixture exists and
 * KMC cannot be run here) and follow KMC's documented CLI behaviour.
 *
 * Databases are arrays sorted by k-mer: keys[n][w] little-endian 64-bit words, counts[n].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint64_t n;       /* distinct k-mers */
    uint64_t kmers;   /* k-mer instances counted (build only) */
    int k, w;
    uint64_t* keys;
    uint32_t* counts;
} kho_db;

static int g_inner_threads = 1;   /* threads inside one sort (kho_exp1 sets it; 1 = the serial passes) */

#define KEY_T uint64_t
#define SFX 64
#include "kh_oracle_impl.inc"
#undef KEY_T
#undef SFX
#define KEY_T unsigned __int128
#define SFX 128
#include "kh_oracle_impl.inc"
#undef KEY_T
#undef SFX

void kho_free(kho_db* d) { free(d->keys); free(d->counts); d->keys = NULL; d->counts = NULL; d->n = 0; }

int kho_count(const uint8_t* seq, uint64_t len, int k, uint32_t ci, uint32_t cx, uint32_t cs, kho_db* out) {
    if (k < 1 || k > 64) return -2;
    return k <= 32 ? count_64(seq, len, k, ci, cx, cs, out) : count_128(seq, len, k, ci, cx, cs, out);
}
void kho_set_counts(kho_db* d, uint32_t v) { for (uint64_t i = 0; i < d->n; ++i) d->counts[i] = v; }
int kho_union_sum(const kho_db* const* in, int nin, uint32_t cs, kho_db* out) {
    return in[0]->w == 1 ? union_sum_64(in, nin, cs, out) : union_sum_128(in, nin, cs, out);
}
int kho_simple(const kho_db* a, const kho_db* b, int op, int mode, uint32_t cs, kho_db* out) {
    if (a->k != b->k) return -5;
    return a->w == 1 ? simple_64(a, b, op, mode, cs, out) : simple_128(a, b, op, mode, cs, out);
}
void kho_histogram(const kho_db* d, uint64_t* hist, uint32_t len) {
    memset(hist, 0, 8 * (size_t)len);
    for (uint64_t i = 0; i < d->n; ++i) hist[d->counts[i] < len ? d->counts[i] : len - 1]++;
}

/* The device side of experiment type 1 (exp_type_1.smk:156-259) for one k, on the CPU:
 * per genome count + set_counts 1, per group union-sum + histogram, group sets,
 * across-group union-sum + histogram.  Genomes and groups are spread over OpenMP threads.
 * Returns the number of threads used (>= 1) or a negative error. */
int kho_exp1(int nseq, const uint8_t* const* seqs, const uint64_t* lens, const int* group_of, int ngroups,
             int k, uint32_t cs, uint64_t* within_hist, uint64_t* across_hist, uint32_t hist_len,
             uint64_t* distinct_per_seq, int nthreads) {
    kho_db* g = (kho_db*)calloc(nseq, sizeof *g);
    kho_db* u = (kho_db*)calloc(ngroups, sizeof *u);
    int used = 1, err = 0;
    int outer = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
    /* more threads than genomes: the surplus works INSIDE the sorts (parallel radix passes) */
    outer = used > nseq ? nseq : used;
    g_inner_threads = used / outer > 1 ? used / outer : 1;
    omp_set_max_active_levels(2);
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(outer)
    for (int i = 0; i < nseq; ++i) {
        if (kho_count(seqs[i], lens[i], k, 1, 0xffffffffu, 255, &g[i]) != 0) err = 1;
        else { kho_set_counts(&g[i], 1); if (distinct_per_seq) distinct_per_seq[i] = g[i].n; }
    }
#ifdef _OPENMP
    {   /* group unions: fewer, larger sorts */
        const int og = used > ngroups ? ngroups : used;
        g_inner_threads = used / og > 1 ? used / og : 1;
        outer = og;
    }
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(outer)
    for (int grp = 0; grp < ngroups; ++grp) {
        const kho_db** in = (const kho_db**)malloc(nseq * sizeof *in);
        int m = 0;
        for (int i = 0; i < nseq; ++i) if (group_of[i] == grp) in[m++] = &g[i];
        if (m == 0 || kho_union_sum(in, m, cs, &u[grp]) != 0) err = 1;
        else {
            if (within_hist) kho_histogram(&u[grp], within_hist + (size_t)grp * hist_len, hist_len);
            kho_set_counts(&u[grp], 1);
        }
        free(in);
    }
    g_inner_threads = used;   /* the across-group union is one sort: all threads inside it */
    if (!err) {
        const kho_db** in = (const kho_db**)malloc(ngroups * sizeof *in);
        for (int grp = 0; grp < ngroups; ++grp) in[grp] = &u[grp];
        kho_db all;
        if (kho_union_sum(in, ngroups, cs, &all) != 0) err = 1;
        else { if (across_hist) kho_histogram(&all, across_hist, hist_len); kho_free(&all); }
        free(in);
    }
    for (int i = 0; i < nseq; ++i) kho_free(&g[i]);
    for (int grp = 0; grp < ngroups; ++grp) kho_free(&u[grp]);
    free(g);
    free(u);
    g_inner_threads = 1;
    return err ? -1 : used;
}
