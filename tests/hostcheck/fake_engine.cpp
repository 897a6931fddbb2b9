// TEST INFRASTRUCTURE ONLY — host-only stand-in for the device side of libkhoice_hip.so, so that the product's
// host code (khoice_amd/csrc/kh_io.cpp: FASTA reader, database files, text outputs; kh_cli.cpp: the `kmc` /
// `kmc_tools` argv forms and the `complex` operations-file parser) can be built with
// -fsanitize=address,undefined and run on a machine without a GPU (sanitizers on the CPU side).
//
// "Device memory" is host memory here: DevBuf::p comes from malloc, the few HIP calls kh_io.cpp makes are
// memcpy / no-ops.  The set arithmetic is the C restatement's (oracle/kh_oracle.c, compiled into this
// executable with the same sanitizers).  Nothing of this is part of the product: the executable lives under
// tests/hostcheck/_build and is only started by tests/test_hostcheck.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "khoice_hip.h"
#include "kh_engine.h"
#include "kh_cli.h"

extern "C" {
typedef struct {
    uint64_t n, kmers;
    int k, w;
    uint64_t* keys;
    uint32_t* counts;
} kho_db;
void kho_free(kho_db* d);
int kho_count(const uint8_t* seq, uint64_t len, int k, uint32_t ci, uint32_t cx, uint32_t cs, kho_db* out);
int kho_union_sum(const kho_db* const* in, int nin, uint32_t cs, kho_db* out);
int kho_simple(const kho_db* a, const kho_db* b, int op, int mode, uint32_t cs, kho_db* out);
}

// ---------------------------------------------------------------- errors
static thread_local std::string g_last_error;
int kh_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
extern "C" const char* kh_last_error(void) { return g_last_error.c_str(); }

// ---------------------------------------------------------------- "HIP"
extern "C" hipError_t hipSetDevice(int) { return hipSuccess; }
extern "C" hipError_t hipGetLastError(void) { return hipSuccess; }
extern "C" const char* hipGetErrorString(hipError_t) { return "host stand-in"; }
extern "C" hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
extern "C" hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
    if (n) memcpy(dst, src, n);
    return hipSuccess;
}

// ---------------------------------------------------------------- buffers, context
void buf_ref(DevBuf* b) { if (b) b->refs.fetch_add(1); }
void buf_unref(DevBuf* b) {
    if (b && b->refs.fetch_sub(1) == 1) { free(b->p); delete b; }
}
DevBuf* kh_ctx::buf_alloc(size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p) return nullptr;
    return new DevBuf{p, bytes, {1}, this};
}
void kh_ctx::prof_begin(int) {}
void kh_ctx::prof_end() {}

// the device ingest is not part of this build
size_t kh_fasta_clean_workspace(u64) { return 0; }
u32 kh_fasta_tile_bytes() { return 4096; }
void kh_launch_fasta_clean(const u8*, u64, const u8*, u8*, void*, unsigned long long*, hipStream_t) { abort(); }

// ---------------------------------------------------------------- sets
static kh_set* new_set(int k, u64 n, DevBuf* kb, DevBuf* cb, u32 uniform, u32 counter_max) {
    kh_set* s = new kh_set;
    s->k = k; s->W = k <= 32 ? 1 : 2; s->n = n; s->kb = kb; s->koff = 0; s->cb = cb; s->coff = 0;
    s->uniform = uniform; s->counter_max = counter_max;
    return s;
}
int kh_set_from_mixed_host(kh_ctx* c, int k, u64 n, const void* keys, const u32* counts, u32 uniform, u32 counter_max,
                           kh_set** out) {
    const size_t kb = 8 * (size_t)(k <= 32 ? 1 : 2);
    if (!n) { *out = new_set(k, 0, nullptr, nullptr, uniform, counter_max); return KH_OK; }
    DevBuf* kbuf = c->buf_alloc(kb * n);
    DevBuf* cbuf = counts ? c->buf_alloc(4 * n) : nullptr;
    if (!kbuf || (counts && !cbuf)) { buf_unref(kbuf); buf_unref(cbuf); return kh_fail(KH_E_NOMEM, "allocation failed"); }
    memcpy(kbuf->p, keys, kb * n);
    if (counts) memcpy(cbuf->p, counts, 4 * n);
    *out = new_set(k, n, kbuf, cbuf, uniform, counter_max);
    return KH_OK;
}
extern "C" void kh_set_free(kh_set* s) {
    if (!s) return;
    buf_unref(s->kb);
    buf_unref(s->cb);
    delete s;
}
extern "C" int kh_set_info(const kh_set* s, uint64_t* n, int* k, int* w, int* has_counts, uint32_t* uniform) {
    if (!s) return kh_fail(KH_E_ARG, "set is NULL");
    if (n) *n = s->n;
    if (k) *k = s->k;
    if (w) *w = s->W;
    if (has_counts) *has_counts = s->cb != nullptr;
    if (uniform) *uniform = s->uniform;
    return KH_OK;
}
extern "C" int kh_set_counter_max(const kh_set* s, uint32_t* cm) {
    if (!s || !cm) return kh_fail(KH_E_ARG, "kh_set_counter_max: NULL argument");
    *cm = s->counter_max;
    return KH_OK;
}
extern "C" void kh_mix_host(int k, const uint64_t* in, uint64_t* out) {
    if (k <= 32) { KmerKey<1> a{in[0]}; a = kh_mix(a, k); out[0] = a.lo; }
    else { KmerKey<2> a{in[0], in[1]}; a = kh_mix(a, k); out[0] = a.lo; out[1] = a.hi; }
}
static void unmix_host(int k, const uint64_t* in, uint64_t* out) {
    if (k <= 32) { KmerKey<1> a{in[0]}; a = kh_unmix(a, k); out[0] = a.lo; }
    else { KmerKey<2> a{in[0], in[1]}; a = kh_unmix(a, k); out[0] = a.lo; out[1] = a.hi; }
}
extern "C" int kh_set_download(kh_ctx* c, const kh_set* s, uint64_t* keys, uint32_t* counts) {
    if (!c || !s) return kh_fail(KH_E_ARG, "kh_set_download: NULL argument");
    const uint64_t* mk = static_cast<const uint64_t*>(s->keys_ptr());
    for (u64 i = 0; keys && i < s->n; ++i) unmix_host(s->k, mk + i * s->W, keys + i * s->W);
    for (u64 i = 0; counts && i < s->n; ++i) counts[i] = s->cb ? s->counts_ptr()[i] : s->uniform;
    return KH_OK;
}

// set -> database sorted by k-mer (what the C restatement works on) and back
struct Db {
    kho_db d{};
    Db() = default;
    Db(const Db&) = delete;
    ~Db() { kho_free(&d); }
};
static void to_db(kh_ctx* c, const kh_set* s, Db& out) {
    const int W = s->W;
    std::vector<uint64_t> keys((size_t)s->n * W);
    std::vector<uint32_t> counts(s->n);
    kh_set_download(c, s, keys.data(), counts.data());
    std::vector<u64> idx(s->n);
    for (u64 i = 0; i < s->n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](u64 a, u64 b) {
        return W == 1 ? keys[a] < keys[b]
                      : (keys[2 * a + 1] < keys[2 * b + 1] || (keys[2 * a + 1] == keys[2 * b + 1] && keys[2 * a] < keys[2 * b]));
    });
    out.d.n = s->n; out.d.kmers = 0; out.d.k = s->k; out.d.w = W;
    out.d.keys = static_cast<uint64_t*>(malloc(8 * (size_t)W * (s->n + 1)));
    out.d.counts = static_cast<uint32_t*>(malloc(4 * (s->n + 1)));
    for (u64 i = 0; i < s->n; ++i) {
        for (int w = 0; w < W; ++w) out.d.keys[i * W + w] = keys[idx[i] * W + w];
        out.d.counts[i] = counts[idx[i]];
    }
}
static int from_db(kh_ctx* c, const kho_db& d, u32 counter_max, bool with_counts, kh_set** out) {
    const int W = d.w;
    std::vector<uint64_t> mixed((size_t)d.n * W);
    for (u64 i = 0; i < d.n; ++i) kh_mix_host(d.k, d.keys + i * W, mixed.data() + i * W);
    std::vector<u64> idx(d.n);
    for (u64 i = 0; i < d.n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](u64 a, u64 b) {
        return W == 1 ? mixed[a] < mixed[b]
                      : (mixed[2 * a + 1] < mixed[2 * b + 1] || (mixed[2 * a + 1] == mixed[2 * b + 1] && mixed[2 * a] < mixed[2 * b]));
    });
    std::vector<uint64_t> sk((size_t)d.n * W);
    std::vector<uint32_t> sc(d.n);
    for (u64 i = 0; i < d.n; ++i) {
        for (int w = 0; w < W; ++w) sk[i * W + w] = mixed[idx[i] * W + w];
        sc[i] = d.counts[idx[i]];
    }
    return kh_set_from_mixed_host(c, d.k, d.n, sk.data(), with_counts ? sc.data() : nullptr, 1, counter_max, out);
}

extern "C" int kh_build_batch(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens, int, int k,
                              uint32_t ci, uint32_t cx, uint32_t cs, int with_counts, kh_set** out) {
    if (!c || !seqs || !lens || !out || nseq <= 0) return kh_fail(KH_E_ARG, "kh_build_batch: bad argument");
    if (k < 1 || k > 64) return kh_fail(KH_E_ARG, "k=%d outside the supported range 1..64", k);
    for (int i = 0; i < nseq; ++i) {
        Db db;
        if (kho_count(seqs[i], lens[i], k, ci, cx, cs, &db.d) != 0) return kh_fail(KH_E_INTERNAL, "count failed");
        const int r = from_db(c, db.d, cs, with_counts != 0, &out[i]);
        if (r != KH_OK) return r;
    }
    return KH_OK;
}
extern "C" int kh_set_counts(kh_ctx* c, const kh_set* in, uint32_t value, kh_set** out) {
    if (!c || !in || !out) return kh_fail(KH_E_ARG, "kh_set_counts: NULL argument");
    if (value == 0) return kh_fail(KH_E_ARG, "set_counts 0 would empty the database");
    buf_ref(in->kb);
    *out = new_set(in->k, in->n, in->kb, nullptr, value, in->counter_max);
    return KH_OK;
}
extern "C" int kh_union_sum(kh_ctx* c, const kh_set* const* sets, int nsets, uint32_t cs, kh_set** out, uint64_t* hist,
                            uint32_t hist_len) {
    if (!c || !sets || nsets <= 0 || !out) return kh_fail(KH_E_ARG, "kh_union_sum: bad argument");
    std::vector<Db> dbs(nsets);
    std::vector<const kho_db*> in(nsets);
    for (int i = 0; i < nsets; ++i) {
        if (sets[i]->k != sets[0]->k) return kh_fail(KH_E_KMISMATCH, "operands built with different k");
        to_db(c, sets[i], dbs[i]);
        in[i] = &dbs[i].d;
    }
    Db u;
    if (kho_union_sum(in.data(), nsets, cs, &u.d) != 0) return kh_fail(KH_E_INTERNAL, "union failed");
    if (hist) {
        memset(hist, 0, 8 * (size_t)hist_len);
        for (u64 i = 0; i < u.d.n; ++i) hist[std::min<u32>(u.d.counts[i], hist_len - 1)]++;
    }
    return from_db(c, u.d, cs, true, out);
}
extern "C" int kh_simple(kh_ctx* c, const kh_set* a, const kh_set* b, int op, int mode, uint32_t cs, kh_set** out) {
    if (!c || !a || !b || !out) return kh_fail(KH_E_ARG, "kh_simple: NULL argument");
    if (a->k != b->k) return kh_fail(KH_E_KMISMATCH, "operands built with different k (%d and %d)", a->k, b->k);
    Db da, db2, r;
    to_db(c, a, da);
    to_db(c, b, db2);
    if (kho_simple(&da.d, &db2.d, op, mode, cs, &r.d) != 0) return kh_fail(KH_E_INTERNAL, "simple failed");
    return from_db(c, r.d, cs, true, out);
}
extern "C" int kh_histogram(kh_ctx* c, const kh_set* s, uint64_t* hist, uint32_t hist_len) {
    if (!c || !s || !hist || hist_len < 2) return kh_fail(KH_E_ARG, "kh_histogram: bad argument");
    memset(hist, 0, 8 * (size_t)hist_len);
    for (u64 i = 0; i < s->n; ++i) hist[std::min<u32>(s->cb ? s->counts_ptr()[i] : s->uniform, hist_len - 1)]++;
    return KH_OK;
}

// ---------------------------------------------------------------- driver
//   hostcheck kmc <argv of kmc>             hostcheck kmc_tools <argv of kmc_tools>
//   hostcheck read_fasta <file>             (cleaned text to stdout)
//   hostcheck hist_text <out> <cmax> <count>...
int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: hostcheck kmc|kmc_tools|read_fasta|hist_text ...\n"); return 2; }
    const std::string tool = argv[1];
    std::vector<std::string> args(argv + 2, argv + argc);
    kh_ctx ctx;
    std::string out, err;
    int status = 2;
    if (tool == "kmc") status = kh_cli_kmc(&ctx, args, out, err);
    else if (tool == "kmc_tools") status = kh_cli_kmc_tools(&ctx, args, out, err);
    else if (tool == "read_fasta" && args.size() == 1) {
        uint8_t* seq = nullptr;
        uint64_t len = 0;
        status = kh_read_fasta(args[0].c_str(), &seq, &len) == KH_OK ? 0 : 1;
        if (!status) { fwrite(seq, 1, len, stdout); kh_free_host(seq); }
        else err = std::string(kh_last_error()) + "\n";
    } else if (tool == "hist_text" && args.size() >= 2) {
        std::vector<uint64_t> h;
        for (size_t i = 2; i < args.size(); ++i) h.push_back(strtoull(args[i].c_str(), nullptr, 10));
        status = kh_write_histogram_text(args[0].c_str(), h.data(), (uint32_t)h.size(), (uint32_t)strtoul(args[1].c_str(), nullptr, 10)) == KH_OK ? 0 : 1;
        if (status) err = std::string(kh_last_error()) + "\n";
    }
    fputs(out.c_str(), stdout);
    fputs(err.c_str(), stderr);
    return status;
}
