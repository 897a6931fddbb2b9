// khoice_amd — host side of libkhoice_hip.so: context, device memory pool, set handles,
// orchestration of the gfx950 kernels and the C ABI declared in include/khoice_hip.h.
// There is deliberately no CPU implementation of any operation in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <memory>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/khoice_hip.h"
#include "kh_engine.h"
#include "kh_launch.h"

// ------------------------------------------------------------------------------ errors
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

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return kh_fail(KH_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                           \
    } while (0)
#define KHCHK(expr)                  \
    do {                             \
        int r__ = (expr);            \
        if (r__ != KH_OK) return r__; \
    } while (0)

// ------------------------------------------------------------------------------ pool
// Stream-ordered caching allocator: every user of a ctx runs on its single stream, so a
// block released by one operation can be handed to the next without synchronising.
void* Pool::alloc(size_t bytes, size_t* got) {
    if (bytes == 0) bytes = 256;
    size_t want = (bytes + 255) & ~(size_t)255;
    if (want > (1u << 20)) want = (want + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    auto it = free_.lower_bound(want);
    if (it != free_.end() && it->first <= want + want / 4 + (1u << 20)) {
        void* p = it->second;
        *got = it->first;
        cached_bytes -= it->first;
        free_.erase(it);
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        trim();
        e = hipMalloc(&p, want);
        if (e != hipSuccess) return nullptr;
    }
    total_bytes += want;
    *got = want;
    return p;
}
void Pool::release(void* p, size_t bytes) {
    free_.emplace(bytes, p);
    cached_bytes += bytes;
}
void Pool::trim() {
    for (auto& kv : free_) {
        (void)hipFree(kv.second);
        total_bytes -= kv.first;
    }
    free_.clear();
    cached_bytes = 0;
}

DevBuf* kh_ctx::buf_alloc(size_t bytes) {
    size_t got = 0;
    void* p = pool.alloc(bytes, &got);
    if (!p) return nullptr;
    DevBuf* b = new DevBuf;
    b->p = p;
    b->bytes = got;
    b->refs = 1;
    b->ctx = this;
    live_bufs.fetch_add(1);
    return b;
}
void buf_ref(DevBuf* b) { if (b) b->refs.fetch_add(1); }
void buf_unref(DevBuf* b) {
    if (!b) return;
    if (b->refs.fetch_sub(1) == 1) {
        if (kh_ctx* c = b->ctx) {                           // borrowed buffers have no ctx
            if (!c->closed) {
                c->pool.release(b->p, b->bytes);
                c->live_bufs.fetch_sub(1);
            } else {                                        // set freed after kh_ctx_destroy
                (void)hipSetDevice(c->dev);
                (void)hipFree(b->p);
                if (c->live_bufs.fetch_sub(1) == 1) delete c;
            }
        }
        delete b;
    }
}
// scoped temporary
struct Tmp {
    DevBuf* b = nullptr;
    ~Tmp() { buf_unref(b); }
    template <class T> T* as() const { return reinterpret_cast<T*>(b->p); }
};
#define TMP_ALLOC(tmp, ctx, bytes)                                                        \
    do {                                                                                  \
        (tmp).b = (ctx)->buf_alloc(bytes);                                                \
        if (!(tmp).b) return kh_fail(KH_E_NOMEM, "device allocation of %zu bytes failed", \
                                     (size_t)(bytes));                                    \
    } while (0)

void* kh_ctx::pin_alloc(size_t bytes, size_t* got) {
    bytes = (bytes + 4095) & ~(size_t)4095;
    auto it = pinned_free.lower_bound(bytes);
    if (it != pinned_free.end() && it->first <= 4 * bytes) {
        void* p = it->second;
        *got = it->first;
        pinned_free.erase(it);
        return p;
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    *got = bytes;
    return p;
}
void kh_ctx::pin_release(void* p, size_t bytes) {
    if (p) pinned_free.emplace(bytes, p);
}

// KHOICE_TRACE=1: host-side timeline of kh_exp1_run on stderr (diagnostics only)
#include <chrono>
static bool g_trace = getenv("KHOICE_TRACE") != nullptr;
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double g_t_build_submitted = 0, g_t_build_synced = 0;

// ------------------------------------------------------------------------------ profiling
void kh_ctx::prof_begin(int cls) {
    if (!profile) return;
    ProfEvt ev;
    ev.cls = cls;
    auto take = [&](hipEvent_t* e) {   // events are recycled: creating one costs microseconds
        if (!free_events.empty()) { *e = free_events.back(); free_events.pop_back(); }
        else (void)hipEventCreate(e);
    };
    take(&ev.a);
    take(&ev.b);
    (void)hipEventRecord(ev.a, st);
    evts.push_back(ev);
}
void kh_ctx::prof_end() {
    if (!profile || evts.empty()) return;
    (void)hipEventRecord(evts.back().b, st);
}
void kh_ctx::prof_collect() {
    if (evts.empty()) return;
    (void)hipStreamSynchronize(st);
    for (auto& ev : evts) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
            cls_ms[ev.cls] += ms;
            cls_n[ev.cls] += 1;
        }
        free_events.push_back(ev.a);
        free_events.push_back(ev.b);
    }
    evts.clear();
}
static const char* kClsNames[KC_COUNT] = {"extract_hist", "bucket_plan", "extract_scatter",
                                          "bucket_sort_rle", "range_bounds", "setop", "histogram",
                                          "remix", "copy_in", "union_tagged", "skm_scatter",
                                          "skm_regroup", "skm_union", "skm_big", "skm_pack", "skm_phased"};

// ------------------------------------------------------------------------------ ctx API
extern "C" int kh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int kh_ctx_create(int device, kh_ctx** out) {
    if (!out) return kh_fail(KH_E_ARG, "kh_ctx_create: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return kh_fail(KH_E_HIP, "no HIP device available (%s); khoice_amd has no CPU fallback",
                       e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (device < 0 || device >= n) return kh_fail(KH_E_ARG, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    kh_ctx* c = new kh_ctx;
    c->dev = device;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->arch = prop.gcnArchName;
    c->cus = prop.multiProcessorCount;
    if (c->arch.find("gfx950") == std::string::npos && !getenv("KHOICE_ALLOW_ANY_ARCH")) {
        std::string a = c->arch;
        delete c;
        return kh_fail(KH_E_HIP, "device %d is %s; this library carries gfx950 code only", device,
                       a.c_str());
    }
    HIPCHK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    c->dynamic_order = getenv("KHOICE_TICKETS") != nullptr;   // see KhLookback::dynamic
    *out = c;
    return KH_OK;
}
extern "C" void kh_ctx_destroy(kh_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->dev);
    (void)hipStreamSynchronize(c->st);
    c->prof_collect();
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    for (auto& kv : c->pinned_free) (void)hipHostFree(kv.second);
    c->pinned_free.clear();
    c->pool.trim();
    (void)hipStreamDestroy(c->st);
    c->st = nullptr;
    c->closed = true;
    if (c->live_bufs.load() == 0) delete c;
}
extern "C" int kh_sync(kh_ctx* c) {
    if (!c) return kh_fail(KH_E_ARG, "ctx is NULL");
    HIPCHK(hipStreamSynchronize(c->st));
    return KH_OK;
}
extern "C" int kh_trim(kh_ctx* c) {
    if (!c) return kh_fail(KH_E_ARG, "ctx is NULL");
    HIPCHK(hipStreamSynchronize(c->st));
    c->pool.trim();
    return KH_OK;
}
extern "C" int kh_profile_enable(kh_ctx* c, int on) {
    if (!c) return kh_fail(KH_E_ARG, "ctx is NULL");
    c->prof_collect();
    c->profile = on != 0;
    return KH_OK;
}
extern "C" int kh_stats_reset(kh_ctx* c) {
    if (!c) return kh_fail(KH_E_ARG, "ctx is NULL");
    c->prof_collect();
    for (int i = 0; i < KC_COUNT; ++i) { c->cls_ms[i] = 0; c->cls_n[i] = 0; }
    c->stat = Stats();
    return KH_OK;
}
extern "C" int kh_stats(kh_ctx* c, char* buf, size_t buflen) {
    if (!c || !buf) return kh_fail(KH_E_ARG, "kh_stats: NULL argument");
    c->prof_collect();
    std::string s = "{";
    char t[384];
    snprintf(t, sizeof t, "\"device\":%d,\"arch\":\"%s\",\"cus\":%d,", c->dev, c->arch.c_str(), c->cus);
    s += t;
    snprintf(t, sizeof t,
             "\"builds\":%llu,\"bases\":%llu,\"kmers\":%llu,\"distinct\":%llu,\"setops\":%llu,"
             "\"setop_in\":%llu,\"setop_out\":%llu,\"retries\":%llu,\"order_fallbacks\":%llu,\"skm_records\":%llu,\"big_slots\":%llu,\"pool_bytes\":%zu,",
             (unsigned long long)c->stat.builds, (unsigned long long)c->stat.bases,
             (unsigned long long)c->stat.kmers, (unsigned long long)c->stat.distinct,
             (unsigned long long)c->stat.setops, (unsigned long long)c->stat.setop_in,
             (unsigned long long)c->stat.setop_out, (unsigned long long)c->stat.retries,
             (unsigned long long)c->stat.order_fallbacks, (unsigned long long)c->stat.skm_records,
             (unsigned long long)c->stat.big_slots, c->pool.total_bytes);
    s += t;
    s += "\"kernels\":{";
    for (int i = 0; i < KC_COUNT; ++i) {
        snprintf(t, sizeof t, "%s\"%s\":{\"ms\":%.6f,\"launches\":%llu}", i ? "," : "", kClsNames[i],
                 c->cls_ms[i], (unsigned long long)c->cls_n[i]);
        s += t;
    }
    s += "}}";
    if (s.size() + 1 > buflen) return kh_fail(KH_E_ARG, "kh_stats: buffer too small (%zu needed)", s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return KH_OK;
}

// ------------------------------------------------------------------------------ sets
extern "C" void kh_set_free(kh_set* s) {
    if (!s) return;
    buf_unref(s->kb);
    buf_unref(s->cb);
    delete s;
}
extern "C" int kh_set_info(const kh_set* s, uint64_t* n, int* k, int* w, int* has_counts,
                           uint32_t* uniform) {
    if (!s) return kh_fail(KH_E_ARG, "set is NULL");
    if (n) *n = s->n;
    if (k) *k = s->k;
    if (w) *w = s->W;
    if (has_counts) *has_counts = s->cb != nullptr;
    if (uniform) *uniform = s->uniform;
    return KH_OK;
}
static kh_set* make_set(int k, u64 n, DevBuf* kb, size_t koff, DevBuf* cb, size_t coff, u32 uniform,
                        u32 counter_max = KH_KMC_DEFAULT_CS) {
    kh_set* s = new kh_set;
    s->k = k;
    s->W = k <= 32 ? 1 : 2;
    s->n = n;
    s->kb = kb;
    s->koff = koff;
    s->cb = cb;
    s->coff = coff;
    s->uniform = uniform;
    s->counter_max = counter_max;
    return s;
}
static int check_k(int k) {
    if (k < 1 || k > 64) return kh_fail(KH_E_ARG, "k=%d outside the supported range 1..64", k);
    return KH_OK;
}

extern "C" int kh_set_counter_max(const kh_set* s, uint32_t* counter_max) {
    if (!s || !counter_max) return kh_fail(KH_E_ARG, "kh_set_counter_max: NULL argument");
    *counter_max = s->counter_max;
    return KH_OK;
}

extern "C" int kh_set_device_ptrs(const kh_set* s, const void** keys, const uint32_t** counts) {
    if (!s) return kh_fail(KH_E_ARG, "set is NULL");
    if (keys) *keys = s->n ? s->keys_ptr() : nullptr;
    if (counts) *counts = s->counts_ptr();
    return KH_OK;
}

extern "C" int kh_set_counts(kh_ctx* c, const kh_set* in, uint32_t value, kh_set** out) {
    if (!c || !in || !out) return kh_fail(KH_E_ARG, "kh_set_counts: NULL argument");
    if (value == 0) return kh_fail(KH_E_ARG, "set_counts 0 would empty the database");
    buf_ref(in->kb);
    *out = make_set(in->k, in->n, in->kb, in->koff, nullptr, 0, value, std::max<u32>(value, KH_KMC_DEFAULT_CS));
    return KH_OK;
}

#ifdef KH_STAMPS
void kh_debug_set_stamps(u64* p);
void kh_debug_set_stamps_skm(u64* p);
// diagnostic build: average shader-clock deltas between phase stamps over all workgroups
static void report_stamps(kh_ctx* c, const char* what, DevBuf* sb, u64 nparts) {
    std::vector<u64> h(nparts * 16);
    (void)hipMemcpyAsync(h.data(), sb->p, nparts * 128, hipMemcpyDeviceToHost, c->st);
    (void)hipStreamSynchronize(c->st);
    double sum[16] = {0};
    u64 cnt = 0, tmin = ~0ull, tmax = 0;
    for (u64 q = 0; q < nparts; ++q) {
        const u64* t = &h[q * 16];
        if (!t[0] || !t[8]) continue;
        ++cnt;
        for (int i = 1; i <= 8; ++i) if (t[i] && t[i - 1]) sum[i] += (double)(t[i] - t[i - 1]);
        tmin = std::min(tmin, t[0]);
        tmax = std::max(tmax, t[8]);
    }
    {   // optional finer stamps 9..11 inside the phase between stamps 4 and 5
        double s9 = 0, s10 = 0, s11 = 0; u64 c9 = 0;
        for (u64 q = 0; q < nparts; ++q) {
            const u64* t = &h[q * 16];
            if (!t[4] || !t[9] || !t[10] || !t[11] || !t[5]) continue;
            ++c9; s9 += (double)(t[9] - t[4]); s10 += (double)(t[10] - t[9]); s11 += (double)(t[11] - t[10]);
        }
        if (c9) fprintf(stderr, "[stamps] %s inside 4-5: first-probe %.0f rounds %.0f stores %.0f (then barrier)\n", what, s9 / c9, s10 / c9, s11 / c9);
    }
    fprintf(stderr, "[stamps] %s parts=%llu span=%.0f cyc | load %.0f count %.0f scan %.0f scatter %.0f insert %.0f walk1 %.0f lookback %.0f walk2 %.0f\n",
            what, (unsigned long long)cnt, (double)(tmax - tmin), sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt,
            sum[5] / cnt, sum[6] / cnt, sum[7] / cnt, sum[8] / cnt);
    {   // every stamp as an offset from stamp 0 (order-free: stamps 9..15 may sit anywhere)
        std::string line;
        for (int i = 1; i < 16; ++i) {
            double s2 = 0; u64 c2 = 0;
            for (u64 q = 0; q < nparts; ++q) {
                const u64* t = &h[q * 16];
                if (!t[0] || !t[i]) continue;
                ++c2; s2 += (double)((long long)(t[i] - t[0]));
            }
            if (c2) { char b[64]; snprintf(b, sizeof b, " t%d=%.0f(n=%llu)", i, s2 / c2, (unsigned long long)c2); line += b; }
        }
        fprintf(stderr, "[stamps] %s offsets from t0:%s\n", what, line.c_str());
    }
}
#endif

// ------------------------------------------------------------------------------ K1 build
// What a grid-mode build (KhGrid, kh_launch.h) leaves behind: nothing has been synchronised when
// build_once returns, the caller queues the tagged union behind it and waits once.
struct GridBuild {
    kh_ctx* c = nullptr;
    u32 fan = 1;            // in: largest group (sets the slot fill of the tagged union)
    u32 cap = 0;            // in: slot capacity of the tagged union
    double s_scale = 1.0;       // in: more sub-ranges per bucket than the estimate (retry after a slot overflow)
    u32 wave = 0, nwaves = 1;   // in: key-range wave (KhSeg::nb_virtual / b_first): this build keeps slice `wave` of `nwaves`
    u32 nb = 0, S = 0, nb_total = 0;
    u64 total_pos = 0, bases = 0, key_cap = 0;   // key_cap: records the key arrays of this build can hold
    DevBuf *okeys = nullptr, *bstart = nullptr, *off = nullptr, *distinct = nullptr, *lb = nullptr;
    void* pin = nullptr;    // plan staging: must outlive the asynchronous upload
    size_t pin_bytes = 0;
    ~GridBuild() {
        buf_unref(okeys); buf_unref(bstart); buf_unref(off); buf_unref(distinct); buf_unref(lb);
        if (pin && c) c->pin_release(pin, pin_bytes);
    }
};

static int build_once(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens,
                      int on_device, int k, u32 ci, u32 cx, u32 cs, int with_counts, u32 mean,
                      kh_set** out_sets, bool* capacity_hit, bool* order_retry, GridBuild* grid = nullptr) {
    const int W = k <= 32 ? 1 : 2;
    const size_t kb = 8 * (size_t)W;
    hipStream_t st = c->st;
    *capacity_hit = false;
    *order_retry = false;

    // ---- segment layout
    std::vector<KhSeg> segs(nseq);
    std::vector<u64> pack_off(nseq);
    std::vector<KhTile> tiles;
    u64 seq_bytes = 0, total_pos = 0, thist_n = 0, bases = 0;
    u32 nb_total = 0, max_nb = 1;
    // Positions per extraction workgroup (tile): 65536 amortises the cursor rows best, but a small
    // batch (the per-call path: one 5 Mbp genome is 77 such tiles on 256 CUs) is cut finer so that
    // passes A and B fill the chip — down to one staging round of pass B.
    u32 tile_pos = KH_TILE;
    {
        u64 all_pos = 0;
        for (int i = 0; i < nseq; ++i) all_pos += lens[i] >= (u64)k ? lens[i] - k + 1 : 0;
        const u64 want_tiles = 4ull * (u64)std::max(1, c->cus);
        while (tile_pos > 2 * KH_SUBTILE && (all_pos + tile_pos - 1) / tile_pos < want_tiles) tile_pos >>= 1;
        if (const char* ev = getenv("KHOICE_TILE_POS")) tile_pos = std::max<u32>(2 * KH_SUBTILE, (u32)strtoul(ev, nullptr, 10) / (2 * KH_SUBTILE) * (2 * KH_SUBTILE));
    }
    u64 grid_nb = 1;   // grid mode: one bucket grid for every sequence, sized by the longest
    const u32 nwaves = grid ? std::max<u32>(1, grid->nwaves) : 1u;
    if (grid) {
        const u64 per = (u64)mean * nwaves;   // a wave keeps 1/nwaves of every genome's keys
        for (int i = 0; i < nseq; ++i)
            if (lens[i] >= (u64)k) grid_nb = std::max<u64>(grid_nb, (lens[i] - k + 1 + per - 1) / per);
    }
    for (int i = 0; i < nseq; ++i) {
        KhSeg& s = segs[i];
        s.seq = nullptr;
        pack_off[i] = seq_bytes;
        s.len = lens[i];
        s.npos = lens[i] >= (u64)k ? lens[i] - k + 1 : 0;
        const u64 want_b = grid ? grid_nb : std::max<u64>(1, (s.npos + mean - 1) / mean);
        if (want_b > KH_MAX_BUCKETS_PER_SEG)
            return kh_fail(KH_E_ARG,
                           "sequence %d has %llu k-mer positions; at most %llu per sequence are "
                           "supported by this build",
                           i, (unsigned long long)s.npos,
                           (unsigned long long)KH_MAX_BUCKETS_PER_SEG * mean);
        s.nbuckets = (u32)want_b;
        s.nb_virtual = (u32)want_b * nwaves;
        s.b_first = grid ? grid->wave * (u32)want_b : 0u;
        s.bucket_base = nb_total;
        s.ntiles = (u32)((s.npos + tile_pos - 1) / tile_pos);
        s.tile_base = (u32)tiles.size();
        s.thist_base = thist_n;
        for (u32 t = 0; t < s.ntiles; ++t) tiles.push_back(KhTile{(u32)i, t});
        thist_n += (u64)s.ntiles * s.nbuckets;
        nb_total += s.nbuckets;
        max_nb = std::max(max_nb, s.nbuckets);
        total_pos += s.npos;
        seq_bytes += (lens[i] + 15) & ~15ull;
        bases += lens[i];
    }
    seq_bytes += 256;   // tail padding: 16-byte loads may run past the last base
    const u32 ntiles = (u32)tiles.size();
    const u32 nb_alloc = (max_nb + 3) & ~3u;

    // ---- device buffers
    Tmp d_seq, d_thist, d_tot, d_bstart, d_part, d_lb;
    TMP_ALLOC(d_seq, c, seq_bytes);
    TMP_ALLOC(d_thist, c, 4 * std::max<u64>(1, thist_n));
    TMP_ALLOC(d_tot, c, 8 * (u64)nb_total);
    TMP_ALLOC(d_bstart, c, 8 * ((u64)nb_total + 1));
    Tmp d_scan;
    TMP_ALLOC(d_scan, c, 8 * kh_exscan_tmp_words(nb_total));
    // a key-range wave keeps about total_pos / nwaves keys — how many exactly is known after the
    // bucket plan, so its partition and output arrays are allocated there (one extra wait per wave)
    const bool late_alloc = nwaves > 1;
    u64 key_cap = std::max<u64>(1, total_pos);
    if (!late_alloc) TMP_ALLOC(d_part, c, kb * key_cap);
    TMP_ALLOC(d_lb, c, 8 * (u64)nb_total + 64);
    KhGrid kgrid{nullptr, nullptr, 0, 0};
    if (grid) {
        // sub-ranges per bucket: the tagged union's slot = one sub-range of one bucket in every
        // genome; fill T with T + 5 sigma <= capacity, sigma <= sqrt(T * fan) (copies of one key
        // inside a group arrive together; setop_prepare has the same rule)
        const double zg = 5.0 * std::sqrt((double)std::max<u32>(1, grid->fan));
        const double x = 0.5 * (-zg + std::sqrt(zg * zg + 4.0 * (double)grid->cap));
        const u64 target = std::max<u64>(16, std::min<u64>((u64)grid->cap * 92 / 100, (u64)(x * x)));
        const u64 per_bucket = (total_pos + grid_nb * nwaves - 1) / (grid_nb * nwaves);   // keys of one bucket over all genomes
        grid->S = (u32)std::max<u64>(1, (u64)std::ceil((double)((per_bucket + target - 1) / target) * grid->s_scale));
        if (grid->S > (u32)KH_FINE_BINS / 2 || grid_nb * grid->S > 0x7fffffffull)
            return kh_fail(KH_E_ARG, "grid build: %llu sub-ranges per bucket", (unsigned long long)grid->S);
        grid->c = c;
        grid->nb = (u32)grid_nb;
        grid->nb_total = nb_total;
        grid->total_pos = total_pos;
        grid->bases = bases;
        grid->off = c->buf_alloc(2 * (size_t)nb_total * (grid->S + 1));
        grid->distinct = c->buf_alloc(8 * (size_t)nseq);
        if (!grid->off || !grid->distinct) return kh_fail(KH_E_NOMEM, "device allocation failed (bucket index)");
        HIPCHK(hipMemsetAsync(grid->distinct->p, 0, 8 * (size_t)nseq, st));
        kgrid = KhGrid{reinterpret_cast<u16*>(grid->off->p), reinterpret_cast<unsigned long long*>(grid->distinct->p),
                       grid->S, grid->nb};
    }
    DevBuf* okeys = late_alloc ? nullptr : c->buf_alloc(kb * key_cap);
    if (!late_alloc && !okeys) return kh_fail(KH_E_NOMEM, "device allocation failed (output keys)");
    DevBuf* ocnt = nullptr;
    if (with_counts) {
        ocnt = c->buf_alloc(4 * std::max<u64>(1, total_pos));
        if (!ocnt) { buf_unref(okeys); return kh_fail(KH_E_NOMEM, "device allocation failed (counts)"); }
    }
    struct Guard { DevBuf *&a, *&b; ~Guard() { buf_unref(a); buf_unref(b); } } guard{okeys, ocnt};

    // device-resident, 16-byte aligned inputs are read in place; anything else is packed into
    // the aligned batch buffer first
    c->prof_begin(KC_COPY_IN);
    for (int i = 0; i < nseq; ++i) {
        if (on_device && (reinterpret_cast<uintptr_t>(seqs[i]) & 15) == 0) {
            segs[i].seq = seqs[i];
            continue;
        }
        segs[i].seq = d_seq.as<u8>() + pack_off[i];
        if (!lens[i]) continue;
        HIPCHK(hipMemcpyAsync(d_seq.as<u8>() + pack_off[i], seqs[i], lens[i],
                              on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    // Start order of pass C (KhBucketWork): buckets of up to 64 consecutive segments are
    // interleaved (bucket 0 of each, bucket 1 of each, ...), every segment writes into its own
    // output region [out_base, out_base + npos) and runs its own look-back chain.
    struct Pin {
        kh_ctx* c; void* p = nullptr; size_t bytes = 0;
        ~Pin() { if (p) c->pin_release(p, bytes); }
    } plan_pin{c};
    // the launch sequence's small tables — output bases, start ranks, segments, tiles — travel in ONE upload
    // (three separate copies, two of them from pageable vectors, were a sixth of a single-genome build)
    const size_t up_rank = 8 * (size_t)nseq, up_segs = (up_rank + 4 * (size_t)nb_total + 7) & ~(size_t)7,
                 up_tiles = up_segs + sizeof(KhSeg) * (size_t)nseq,
                 up_bytes = up_tiles + sizeof(KhTile) * (size_t)std::max<u32>(1, ntiles);
    plan_pin.p = c->pin_alloc(up_bytes, &plan_pin.bytes);
    if (!plan_pin.p) return kh_fail(KH_E_NOMEM, "pinned host allocation failed");
    u64* h_out_base = static_cast<u64*>(plan_pin.p);
    u32* h_rank = reinterpret_cast<u32*>(h_out_base + nseq);
    memcpy(static_cast<u8*>(plan_pin.p) + up_segs, segs.data(), sizeof(KhSeg) * (size_t)nseq);
    if (ntiles) memcpy(static_cast<u8*>(plan_pin.p) + up_tiles, tiles.data(), sizeof(KhTile) * (size_t)ntiles);
    {
        u64 ob = 0;
        u32 next = 0;
        for (int i = 0; i < nseq; ++i) { h_out_base[i] = ob; ob += segs[i].npos; }
        for (int s0 = 0; s0 < nseq; s0 += 64) {
            const int s1 = std::min(nseq, s0 + 64);
            u32 deepest = 0;
            for (int i = s0; i < s1; ++i) deepest = std::max(deepest, segs[i].nbuckets);
            for (u32 b = 0; b < deepest; ++b)
                for (int i = s0; i < s1; ++i)
                    if (b < segs[i].nbuckets) h_rank[segs[i].bucket_base + b] = next++;
        }
    }
    Tmp d_plan, d_work;
    TMP_ALLOC(d_plan, c, up_bytes);
    TMP_ALLOC(d_work, c, sizeof(KhBucketWork) * (size_t)nb_total);
    HIPCHK(hipMemcpyAsync(d_plan.b->p, plan_pin.p, up_bytes, hipMemcpyHostToDevice, st));
    const u64* d_out_base = d_plan.as<u64>();
    const u32* d_rank = reinterpret_cast<const u32*>(d_out_base + nseq);
    const KhSeg* d_segs_p = reinterpret_cast<const KhSeg*>(d_plan.as<u8>() + up_segs);
    const KhTile* d_tiles_p = reinterpret_cast<const KhTile*>(d_plan.as<u8>() + up_tiles);
    c->prof_end();

    // ---- pass A, bucket plan, pass B
    c->prof_begin(KC_EXTRACT_HIST);
    kh_launch_extract(W, false, d_seq.as<u8>(), d_segs_p, d_tiles_p, ntiles,
                      nb_alloc, k, d_thist.as<u32>(), nullptr, nullptr, tile_pos, st);
    c->prof_end();
    c->prof_begin(KC_BUCKET_PLAN);
    kh_launch_col_totals(d_segs_p, nseq, max_nb, d_thist.as<u32>(), d_tot.as<u64>(), st);
    kh_launch_exscan(d_tot.as<u64>(), d_bstart.as<u64>(), nb_total, d_scan.as<u64>(), st);
    Tmp d_over;   // grid mode: list of the buckets above the LDS capacity
    if (grid) {
        TMP_ALLOC(d_over, c, 4 * ((size_t)nb_total + 1));
        HIPCHK(hipMemsetAsync(d_over.b->p, 0, 4, st));
    }
    kh_launch_col_offsets(d_segs_p, nseq, max_nb, d_thist.as<u32>(), d_bstart.as<u64>(), d_rank,
                          d_out_base, d_work.as<KhBucketWork>(), grid ? d_over.as<u32>() : nullptr,
                          grid ? kh_grid_bucket_capacity(W) : 0u, st);
    c->prof_end();
    if (late_alloc) {
        u64 nkeys = 0;
        HIPCHK(hipMemcpyAsync(&nkeys, d_bstart.as<u64>() + nb_total, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        key_cap = std::max<u64>(1, nkeys);
        TMP_ALLOC(d_part, c, kb * key_cap);
        okeys = c->buf_alloc(kb * key_cap);
        if (!okeys) return kh_fail(KH_E_NOMEM, "device allocation failed (output keys)");
    }
#ifdef KH_STAMPS
    Tmp d_stamps_b;
    TMP_ALLOC(d_stamps_b, c, 128 * (u64)std::max<u32>(1, ntiles));
    HIPCHK(hipMemsetAsync(d_stamps_b.b->p, 0, 128 * (u64)std::max<u32>(1, ntiles), st));
    kh_debug_set_stamps(d_stamps_b.as<u64>());
#endif
    c->prof_begin(KC_EXTRACT_SCATTER);
    kh_launch_extract(W, true, d_seq.as<u8>(), d_segs_p, d_tiles_p, ntiles,
                      nb_alloc, k, d_thist.as<u32>(), d_bstart.as<u64>(), d_part.b->p, tile_pos, st);
    c->prof_end();
#ifdef KH_STAMPS
    report_stamps(c, "extract_scatter (last round of a tile: decode / extract / scan / place / flush / - / - / cursors)", d_stamps_b.b, ntiles);
    kh_debug_set_stamps(nullptr);
#endif

    // ---- pass C
    KhLookback lb;
    lb.desc = d_lb.as<u64>();
    lb.ticket = reinterpret_cast<u32*>(d_lb.as<u64>() + nb_total);
    lb.err = lb.ticket + 1;
    lb.dynamic = c->dynamic_order ? 1u : 0u;
    HIPCHK(hipMemsetAsync(d_lb.b->p, 0, 8 * (u64)nb_total + 64, st));
#ifdef KH_STAMPS
    Tmp d_stamps;
    TMP_ALLOC(d_stamps, c, 128 * (u64)nb_total);
    HIPCHK(hipMemsetAsync(d_stamps.b->p, 0, 128 * (u64)nb_total, st));
    kh_debug_set_stamps(d_stamps.as<u64>());
#endif
    c->prof_begin(KC_BUCKET_SORT);
    if (grid)
        kh_launch_grid_bucket(W, d_part.b->p, d_work.as<KhBucketWork>(), nb_total, k, okeys->p, d_over.as<u32>(), lb.err,
                              kgrid, st);
    else
        kh_launch_bucket_sort(W, d_part.b->p, d_work.as<KhBucketWork>(), nb_total, k, okeys->p,
                              ocnt ? reinterpret_cast<u32*>(ocnt->p) : nullptr, lb, ci, cx, cs, st);
    c->prof_end();
    HIPCHK(hipGetLastError());
#ifdef KH_STAMPS
    report_stamps(c, "bucket_sort", d_stamps.b, nb_total);
    kh_debug_set_stamps(nullptr);
#endif
    if (grid) {   // hand the device state over; the caller synchronises
        grid->key_cap = key_cap;
        buf_ref(okeys); grid->okeys = okeys;
        buf_ref(d_bstart.b); grid->bstart = d_bstart.b;
        buf_ref(d_lb.b); grid->lb = d_lb.b;
        grid->pin = plan_pin.p; grid->pin_bytes = plan_pin.bytes;
        plan_pin.p = nullptr;
        return KH_OK;
    }

    // ---- read back set boundaries
    if (g_trace) g_t_build_submitted = now_ms();
    Pin pin{c};
    pin.p = c->pin_alloc(8 * (size_t)nb_total + 64 + 8, &pin.bytes);
    if (!pin.p) return kh_fail(KH_E_NOMEM, "pinned host allocation failed");
    u64* desc = static_cast<u64*>(pin.p);
    u64& nvalid = desc[(size_t)nb_total + 8];
    HIPCHK(hipMemcpyAsync(desc, d_lb.b->p, 8 * (u64)nb_total + 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&nvalid, d_bstart.as<u64>() + nb_total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (g_trace) g_t_build_synced = now_ms();
    const u32 err = reinterpret_cast<const u32*>(&desc[nb_total])[1];
    if (err & KH_ERR_SPIN_TIMEOUT) {
        if (c->dynamic_order) return kh_fail(KH_E_INTERNAL, "look-back spin timed out in bucket sort");
        c->dynamic_order = true;          // index order did not hold: tickets from now on
        c->stat.order_fallbacks++;
        *order_retry = true;              // same batch again with the same plan
        return KH_OK;
    }
    if (err & KH_ERR_CAPACITY) { *capacity_hit = true; return KH_OK; }
    c->stat.kmers += nvalid;
    c->stat.bases += bases;      // counted once per successful build, not per attempt

    for (int i = 0; i < nseq; ++i) {   // a segment's chain ends with its inclusive total
        const u64 n = desc[segs[i].bucket_base + segs[i].nbuckets - 1] & ((1ull << 62) - 1);
        buf_ref(okeys);
        if (ocnt) buf_ref(ocnt);
        out_sets[i] = make_set(k, n, okeys, h_out_base[i] * kb, ocnt, h_out_base[i] * 4, 1, cs);
        c->stat.distinct += n;
    }
    c->stat.builds += nseq;
    return KH_OK;
}

static int build_batch_plain(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens,
                             int on_device, int k, u32 ci, u32 cx, u32 cs, int with_counts,
                             kh_set** out_sets) {
    u32 mean = k <= 32 ? KH_BUCKET_MEAN_W1 : KH_BUCKET_MEAN_W2;
    u64 max_pos = 0;
    for (int i = 0; i < nseq; ++i)
        if (lens[i] >= (u64)k) max_pos = std::max<u64>(max_pos, lens[i] - k + 1);
    for (int attempt = 0; attempt < 6; ++attempt) {
        bool cap = false, again = false;
        for (int i = 0; i < nseq; ++i) out_sets[i] = nullptr;
        int r = build_once(c, nseq, seqs, lens, on_device, k, ci, cx, cs, with_counts, mean, out_sets, &cap, &again);
        if (r != KH_OK) return r;
        if (again) continue;              // look-back order fallback: nothing overflowed, same plan
        if (!cap) return KH_OK;
        c->stat.retries++;
        // more buckets, but never more than one segment's cursor table holds
        const u32 floor_mean = (u32)std::max<u64>(64, (max_pos + KH_MAX_BUCKETS_PER_SEG - 1) / KH_MAX_BUCKETS_PER_SEG);
        const u32 next = std::max<u32>(floor_mean, mean / 4);
        if (next >= mean) break;
        mean = next;
    }
    return kh_fail(KH_E_CAPACITY, "a bucket still holds more distinct k-mers than fit in LDS after re-planning");
}

extern "C" int kh_build_batch(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens,
                              int on_device, int k, uint32_t ci, uint32_t cx, uint32_t cs,
                              int with_counts, kh_set** out_sets) {
    if (!c || !seqs || !lens || !out_sets || nseq <= 0) return kh_fail(KH_E_ARG, "kh_build_batch: bad argument");
    KHCHK(check_k(k));
    if (ci < 1) ci = 1;
    if (cs < 1) return kh_fail(KH_E_ARG, "cs must be >= 1");
    HIPCHK(hipSetDevice(c->dev));
    // A sequence whose k-mer positions exceed what one segment's bucket table covers (the cursor
    // table of pass A/B lives in LDS) is cut into chunks that overlap by k-1 bases: every k-mer
    // start position belongs to exactly one chunk, so the chunk databases add up exactly.
    const u32 mean = k <= 32 ? KH_BUCKET_MEAN_W1 : KH_BUCKET_MEAN_W2;
    u64 chunk_pos = (u64)(KH_MAX_BUCKETS_PER_SEG / 4) * mean;
    if (const char* e = getenv("KHOICE_MAX_SEG_POS")) chunk_pos = std::max<u64>(64, strtoull(e, nullptr, 10));
    bool any_long = false;
    for (int i = 0; i < nseq; ++i)
        if (lens[i] >= (u64)k && lens[i] - k + 1 > chunk_pos) any_long = true;
    if (!any_long) return build_batch_plain(c, nseq, seqs, lens, on_device, k, ci, cx, cs, with_counts, out_sets);
    if (ci > 1 || cx != KH_NO_MAX)
        return kh_fail(KH_E_ARG, "-ci/-cx cut-offs are not supported for sequences of more than %llu k-mer positions",
                       (unsigned long long)chunk_pos);
    std::vector<const uint8_t*> cseq;
    std::vector<uint64_t> clen;
    std::vector<int> first(nseq + 1, 0);
    for (int i = 0; i < nseq; ++i) {
        first[i] = (int)cseq.size();
        const u64 npos = lens[i] >= (u64)k ? lens[i] - k + 1 : 0;
        if (npos <= chunk_pos) { cseq.push_back(seqs[i]); clen.push_back(lens[i]); continue; }
        for (u64 p = 0; p < npos; p += chunk_pos) {
            const u64 np = std::min(chunk_pos, npos - p);
            cseq.push_back(seqs[i] + p);
            clen.push_back(np + k - 1);
        }
    }
    first[nseq] = (int)cseq.size();
    std::vector<kh_set*> csets(cseq.size(), nullptr);
    auto cleanup = [&]() { for (auto* s : csets) kh_set_free(s); };
    // chunk databases keep exact counters (no saturation) until they are added up
    int r = build_batch_plain(c, (int)cseq.size(), cseq.data(), clen.data(), on_device, k, 1, KH_NO_MAX,
                              with_counts ? 0x7fffffffu : cs, with_counts, csets.data());
    if (r != KH_OK) { cleanup(); return r; }
    for (int i = 0; i < nseq; ++i) out_sets[i] = nullptr;
    for (int i = 0; i < nseq && r == KH_OK; ++i) {
        const int m = first[i + 1] - first[i];
        if (m == 1) { out_sets[i] = csets[first[i]]; csets[first[i]] = nullptr; continue; }
        kh_set* u = nullptr;
        r = kh_union_sum(c, csets.data() + first[i], m, with_counts ? cs : 0x7fffffffu, &u, nullptr, 0);
        if (r != KH_OK) break;
        if (with_counts) { out_sets[i] = u; continue; }
        r = kh_set_counts(c, u, 1, &out_sets[i]);     // plain set: every counter 1
        kh_set_free(u);
    }
    if (r != KH_OK) for (int i = 0; i < nseq; ++i) { kh_set_free(out_sets[i]); out_sets[i] = nullptr; }
    cleanup();
    return r;
}

// ------------------------------------------------------------------------------ set ops
// One set operation as an asynchronous job: setop_launch() only enqueues work on the ctx stream
// (kernels + the small device-to-host copies of the result size / histogram); setop_finish()
// is called after a stream synchronisation, turns the result into a handle and re-plans with
// smaller slots in the (rare) case a slot overflowed LDS.  Independent operations can thus be
// enqueued back to back and share one synchronisation (kh_exp1_run does that for the groups).
struct SetopJob {
    kh_ctx* c = nullptr;
    std::vector<const kh_set*> in;
    int op = 0, mode = 0, k = 0, W = 1;
    u32 cs = 0, hist_len = 0, cap = 0, nranges = 0;
    uint64_t* hist = nullptr;
    bool pay = false, empty = false;
    bool hist_only = false;   // nobody reads the output as one array: may run as several chains
    u64 total = 0, target = 0;
    DevBuf *okeys = nullptr, *ocnt = nullptr, *d_views = nullptr, *d_bounds = nullptr, *d_lb = nullptr;
    // pinned staging, mirroring the device workspace from its last descriptor on:
    // [last descriptor: 1 x u64][control: 8 x u64 = ticket, err | fullest slot | ...][hist][views]
    void* pin = nullptr;
    size_t pin_bytes = 0;
    u64* tail = nullptr;
    u64* pin_hist = nullptr;
    KhSetView* views = nullptr;
    u64 lb_words() const { return (u64)nranges + 8 + (hist ? hist_len : 0); }
    ~SetopJob() {
        buf_unref(okeys); buf_unref(ocnt); buf_unref(d_views); buf_unref(d_bounds); buf_unref(d_lb);
        if (pin && c) c->pin_release(pin, pin_bytes);
    }
};
#define JOB_ALLOC(field, bytes)                                                              \
    do {                                                                                     \
        buf_unref(j.field);                                                                  \
        j.field = c->buf_alloc(bytes);                                                       \
        if (!j.field) return kh_fail(KH_E_NOMEM, "device allocation of %zu bytes failed", (size_t)(bytes)); \
    } while (0)

static int setop_prepare(SetopJob& j) {
    kh_ctx* c = j.c;
    const int nsets = (int)j.in.size();
    j.k = j.in[0]->k;
    j.W = j.in[0]->W;
    for (auto* s : j.in)
        if (s->k != j.k) return kh_fail(KH_E_KMISMATCH, "operands built with different k (%d vs %d)", j.k, s->k);
    j.pay = !(j.op == KH_OP_UNION && j.mode == KH_OC_SUM);
    j.total = 0;
    if (!j.pin) {
        j.pin = c->pin_alloc(72 + 8 * (size_t)j.hist_len + sizeof(KhSetView) * nsets, &j.pin_bytes);
        if (!j.pin) return kh_fail(KH_E_NOMEM, "pinned host allocation failed");
        j.tail = static_cast<u64*>(j.pin);
        j.pin_hist = j.tail + 9;
        j.views = reinterpret_cast<KhSetView*>(j.pin_hist + j.hist_len);
    }
    j.tail[0] = j.tail[1] = j.tail[2] = j.tail[3] = 0;
    KhSetView* views = j.views;
    for (int g = 0; g < nsets; ++g) {
        views[g].keys = j.in[g]->n ? j.in[g]->keys_ptr() : nullptr;
        views[g].counts = j.in[g]->counts_ptr();
        views[g].n = j.in[g]->n;
        views[g].uniform = j.in[g]->uniform;
        views[g].pad = 0;
        if (j.in[g]->cb || j.in[g]->uniform != 1) j.pay = true;
        j.total += j.in[g]->n;
    }
    c->stat.setops++;
    c->stat.setop_in += j.total;
    j.empty = j.total == 0;
    if (j.empty) return KH_OK;
    j.cap = j.W == 1 ? (j.pay ? KH_SORT_CAP_PAY_W1 : KH_SORT_CAP_W1) : (j.pay ? KH_SORT_CAP_PAY_W2 : KH_SORT_CAP_W2);
    // Mean slot fill T: a slot holds a Poisson number of distinct keys, each up to G = nsets
    // times, so its size has sigma <= sqrt(T * G); T is the largest fill with T + 5 sigma <= cap
    // (G = 5: 84 % of cap, G = 10: 78 %, G = 128: 42 %; 90 % at G = 5 was measured to re-plan).
    {
        const double zg = 5.0 * std::sqrt((double)std::max(1, nsets));
        const double x = 0.5 * (-zg + std::sqrt(zg * zg + 4.0 * (double)j.cap));
        j.target = std::max<u64>(16, std::min<u64>((u64)j.cap * 92 / 100, (u64)(x * x)));
    }
    JOB_ALLOC(okeys, 8 * (size_t)j.W * j.total);
    JOB_ALLOC(ocnt, 4 * j.total);
    JOB_ALLOC(d_views, sizeof(KhSetView) * nsets);
    HIPCHK(hipMemcpyAsync(j.d_views->p, views, sizeof(KhSetView) * nsets, hipMemcpyHostToDevice, c->st));
    return KH_OK;
}

// plan: number of slots and the per-launch workspaces
static int setop_plan(SetopJob& j) {
    kh_ctx* c = j.c;
    if (j.empty) return KH_OK;
    const int nsets = (int)j.in.size();
    const u64 nr64 = std::max<u64>(1, (j.total + j.target - 1) / j.target);
    if (nr64 > 0x7fffffffull) return kh_fail(KH_E_ARG, "set operation too large for one launch");
    j.nranges = (u32)nr64;
    JOB_ALLOC(d_bounds, 8 * ((u64)j.nranges + 1) * nsets);
    // one workspace: [descriptors: nranges][control: 64 B][histogram: hist_len], zeroed by the
    // range-bounds kernel and read back (from the last descriptor on) with one copy
    JOB_ALLOC(d_lb, 8 * j.lb_words());
    return KH_OK;
}

static KhBoundsJob setop_bounds_job(const SetopJob& j) {
    return KhBoundsJob{reinterpret_cast<const KhSetView*>(j.d_views->p), reinterpret_cast<u64*>(j.d_bounds->p),
                       reinterpret_cast<u64*>(j.d_lb->p), j.lb_words(), (u32)j.in.size(), j.nranges};
}

// slot bounds of one operation (also clears its workspace)
static int setop_bounds(SetopJob& j) {
    kh_ctx* c = j.c;
    if (j.empty) return KH_OK;
    c->prof_begin(KC_RANGE_BOUNDS);
    kh_launch_range_bounds(j.W, reinterpret_cast<KhSetView*>(j.d_views->p), (u32)j.in.size(), j.nranges, j.k,
                           reinterpret_cast<u64*>(j.d_bounds->p), reinterpret_cast<u64*>(j.d_lb->p),
                           j.lb_words(), c->st);
    c->prof_end();
    // Test hook (tests/test_gpu_parity.py::test_corrupt_slot_bounds_fail_closed): plant a slot bound
    // that lies past the end of operand 0, i.e. the state a broken bounds pass would leave behind.
    // The set operation must report KH_ERR_ORDER and read nothing through it.
    if (getenv("KHOICE_DEBUG_CORRUPT_BOUNDS") && j.nranges >= 2) {
        static const u64 poison = 0x0000ffffffffff00ull;
        HIPCHK(hipMemcpyAsync(reinterpret_cast<u64*>(j.d_bounds->p) + 1, &poison, 8, hipMemcpyHostToDevice, c->st));
    }
    return KH_OK;
}

// slot bounds of several planned operations (same k) in ONE launch
static int setop_bounds_batch(kh_ctx* c, std::vector<SetopJob>& jobs, Tmp& d_jobs, void** pin, size_t* pin_bytes) {
    std::vector<KhBoundsJob> hb;
    u64 max_threads = 0;
    int W = 1, k = 0;
    for (auto& j : jobs) {
        if (j.empty || j.in.empty()) continue;
        hb.push_back(setop_bounds_job(j));
        max_threads = std::max<u64>(max_threads, ((u64)j.nranges + 1) * j.in.size());
        W = j.W;
        k = j.k;
    }
    if (hb.empty()) return KH_OK;
    *pin = c->pin_alloc(sizeof(KhBoundsJob) * hb.size(), pin_bytes);
    if (!*pin) return kh_fail(KH_E_NOMEM, "pinned host allocation failed");
    memcpy(*pin, hb.data(), sizeof(KhBoundsJob) * hb.size());
    TMP_ALLOC(d_jobs, c, sizeof(KhBoundsJob) * hb.size());
    HIPCHK(hipMemcpyAsync(d_jobs.b->p, *pin, sizeof(KhBoundsJob) * hb.size(), hipMemcpyHostToDevice, c->st));
    c->prof_begin(KC_RANGE_BOUNDS);
    kh_launch_range_bounds_batch(W, d_jobs.as<KhBoundsJob>(), (u32)hb.size(), max_threads, k, c->st);
    c->prof_end();
    HIPCHK(hipGetLastError());
    return KH_OK;
}

// kernel-side description of one planned operation: one entry, or — for a histogram-only
// operation in index order — up to KH_SETOP_BATCH chains over consecutive slot ranges
static void setop_kernel_jobs(const SetopJob& j, std::vector<KhSetopJob>& out) {
    u64* desc = reinterpret_cast<u64*>(j.d_lb->p);
    u32* ctl = reinterpret_cast<u32*>(desc + j.nranges);
    u32 chains = 1;
    if (j.hist_only && !j.c->dynamic_order) chains = std::min<u32>(KH_SETOP_BATCH, std::max<u32>(1, j.nranges / 64));
    for (u32 ch = 0; ch < chains; ++ch) {
        const u32 s0 = (u32)((u64)j.nranges * ch / chains), s1 = (u32)((u64)j.nranges * (ch + 1) / chains);
        out.push_back(KhSetopJob{reinterpret_cast<const KhSetView*>(j.d_views->p),
                                 reinterpret_cast<const u64*>(j.d_bounds->p), j.okeys->p,
                                 reinterpret_cast<u32*>(j.ocnt->p), desc + s0, ctl,
                                 j.hist ? reinterpret_cast<unsigned long long*>(desc + j.nranges + 8) : nullptr,
                                 (u32)j.in.size(), j.nranges, s0, s1 - s0});
    }
}

// Planned operations of the same kind (same k, payload mode, operation, cs, hist_len) as launches
// of up to KH_SETOP_BATCH each, then the read-back of every tail.
static int setop_run_batch(kh_ctx* c, SetopJob* const* jobs, size_t njobs) {
    hipStream_t st = c->st;
    std::vector<SetopJob*> live;
    for (size_t i = 0; i < njobs; ++i)
        if (!jobs[i]->empty && !jobs[i]->in.empty()) live.push_back(jobs[i]);
    if (live.empty()) return KH_OK;
    std::vector<KhSetopJob> entries;
    const SetopJob& f = *live[0];
    for (SetopJob* jp : live) {
        const SetopJob& j = *jp;
        if (j.W != f.W || j.pay != f.pay || j.cap != f.cap || j.k != f.k || j.op != f.op || j.mode != f.mode ||
            j.cs != f.cs || j.hist_len != f.hist_len)
            return kh_fail(KH_E_INTERNAL, "set operations of different kinds in one batch");
        setop_kernel_jobs(j, entries);
    }
    for (size_t i0 = 0; i0 < entries.size(); i0 += KH_SETOP_BATCH) {
        const size_t m = std::min<size_t>(KH_SETOP_BATCH, entries.size() - i0);
        KhSetopBatch batch;
        memset(&batch, 0, sizeof batch);
        for (size_t i = 0; i < m; ++i) batch.job[i] = entries[i0 + i];
#ifdef KH_STAMPS
        Tmp d_stamps;
        u64 stamp_parts = 0;
        for (size_t i = 0; i < m; ++i) stamp_parts = std::max<u64>(stamp_parts, batch.job[i].nranges);
        stamp_parts *= m;
        TMP_ALLOC(d_stamps, c, 128 * stamp_parts);
        HIPCHK(hipMemsetAsync(d_stamps.b->p, 0, 128 * stamp_parts, st));
        kh_debug_set_stamps(d_stamps.as<u64>());
#endif
        c->prof_begin(KC_SETOP);
        kh_launch_setop(f.W, f.pay, f.cap, batch, (u32)m, f.k, f.op, f.mode, f.cs, f.hist_len, c->dynamic_order, st);
        c->prof_end();
        HIPCHK(hipGetLastError());
#ifdef KH_STAMPS
        report_stamps(c, "setop", d_stamps.b, stamp_parts);
        kh_debug_set_stamps(nullptr);
#endif
    }
    for (SetopJob* j : live)
        HIPCHK(hipMemcpyAsync(j->tail, reinterpret_cast<u64*>(j->d_lb->p) + (j->nranges - 1),
                              8 * (9 + (j->hist ? j->hist_len : 0)), hipMemcpyDeviceToHost, st));
    return KH_OK;
}

static int setop_run(SetopJob& j) {
    SetopJob* one = &j;
    return setop_run_batch(j.c, &one, 1);
}

static int setop_launch(SetopJob& j) {
    KHCHK(setop_plan(j));
    KHCHK(setop_bounds(j));
    return setop_run(j);
}

// precondition: the stream was synchronised after setop_launch(j)
static int setop_finish(SetopJob& j, kh_set** out) {
    kh_ctx* c = j.c;
    if (j.empty) {
        if (out) *out = make_set(j.k, 0, nullptr, 0, nullptr, 0, 1, j.cs);
        if (j.hist) memset(j.hist, 0, 8 * (size_t)j.hist_len);
        return KH_OK;
    }
    for (int attempt = 0; attempt < 8; ++attempt) {
        const u32 err = reinterpret_cast<const u32*>(&j.tail[1])[1];
        if (err & KH_ERR_ORDER)
            return kh_fail(KH_E_ARG, "set operation: an operand is not sorted by mixed key (kh_set_wrap_device and "
                                     "kh_set_from_device trust their caller on that); nothing was read outside it");
        if (err & KH_ERR_SPIN_TIMEOUT) {
            if (c->dynamic_order) return kh_fail(KH_E_INTERNAL, "look-back spin timed out in set operation");
            c->dynamic_order = true;      // index order did not hold: tickets from now on
            c->stat.order_fallbacks++;
            KHCHK(setop_launch(j));
            HIPCHK(hipStreamSynchronize(c->st));
            continue;
        }
        if (!(err & KH_ERR_CAPACITY)) {
            const u64 n = j.tail[3];   // outputs of all chains (control words, u64 at byte 16)
            if (out) {
                buf_ref(j.okeys);
                buf_ref(j.ocnt);
                *out = make_set(j.k, n, j.okeys, 0, j.ocnt, 0, 1, j.cs);
            }
            c->stat.setop_out += n;
            if (j.hist) memcpy(j.hist, j.pin_hist, 8 * (size_t)j.hist_len);
            return KH_OK;
        }
        c->stat.retries++;
        // the kernel recorded its fullest slot: shrink the mean fill so that one fits with 10 %
        // to spare (never by less than 1/8, by 4 when the record is missing)
        const u64 fullest = (u32)j.tail[2];
        u64 next = fullest > j.cap ? j.target * j.cap * 9 / (fullest * 10) : j.target / 4;
        j.target = std::max<u64>(16, std::min<u64>(next, j.target * 7 / 8));
        KHCHK(setop_launch(j));
        HIPCHK(hipStreamSynchronize(c->st));
    }
    return kh_fail(KH_E_CAPACITY, "set operation: a key range still overflows LDS after 8 re-plans");
}

static int run_setop(kh_ctx* c, const std::vector<const kh_set*>& in, int op, int mode, u32 cs,
                     kh_set** out, uint64_t* hist, u32 hist_len) {
    SetopJob j;
    j.c = c; j.in = in; j.op = op; j.mode = mode; j.cs = cs; j.hist = hist; j.hist_len = hist_len;
    j.hist_only = out == nullptr;   // no set wanted: the output need not be one compact array
    KHCHK(setop_prepare(j));
    KHCHK(setop_launch(j));
    HIPCHK(hipStreamSynchronize(c->st));
    return setop_finish(j, out);
}

extern "C" int kh_union_sum(kh_ctx* c, const kh_set* const* sets, int nsets, uint32_t cs, kh_set** out,
                            uint64_t* hist, uint32_t hist_len) {
    if (!c || !sets || !out || nsets <= 0) return kh_fail(KH_E_ARG, "kh_union_sum: bad argument");
    if (hist && hist_len < 2) return kh_fail(KH_E_ARG, "hist_len must be >= 2");
    if (cs < 1) return kh_fail(KH_E_ARG, "cs must be >= 1");
    HIPCHK(hipSetDevice(c->dev));
    for (int i = 0; i < nsets; ++i)
        if (!sets[i]) return kh_fail(KH_E_ARG, "kh_union_sum: operand %d is NULL", i);
    if (nsets <= KH_MAX_INPUT_SETS) {
        std::vector<const kh_set*> in(sets, sets + nsets);
        return run_setop(c, in, KH_OP_UNION, KH_OC_SUM, cs, out, hist, hist_len);
    }
    // fan-in above one launch's limit: sum groups of 64 without saturation, then sum the sums
    std::vector<kh_set*> partial;
    auto cleanup = [&]() { for (auto* p : partial) kh_set_free(p); };
    for (int i = 0; i < nsets; i += KH_MAX_INPUT_SETS) {
        const int m = std::min(KH_MAX_INPUT_SETS, nsets - i);
        std::vector<const kh_set*> in(sets + i, sets + i + m);
        kh_set* p = nullptr;
        int r = run_setop(c, in, KH_OP_UNION, KH_OC_SUM, 0x7fffffffu, &p, nullptr, 0);
        if (r != KH_OK) { cleanup(); return r; }
        partial.push_back(p);
    }
    int r = kh_union_sum(c, partial.data(), (int)partial.size(), cs, out, hist, hist_len);
    cleanup();
    return r;
}

// The histogram of a union-sum without handing the union back (steps 7+8 of experiment type 1
// when only step_8's histogram is wanted, and the merged slices of the multi-GPU exchange):
// every output record is still written, but not as one compact array, so the slots run as
// independent chains.
extern "C" int kh_union_histogram(kh_ctx* c, const kh_set* const* sets, int nsets, uint32_t cs,
                                  uint64_t* hist, uint32_t hist_len) {
    if (!c || !sets || !hist || nsets <= 0) return kh_fail(KH_E_ARG, "kh_union_histogram: bad argument");
    if (hist_len < 2) return kh_fail(KH_E_ARG, "hist_len must be >= 2");
    if (cs < 1) return kh_fail(KH_E_ARG, "cs must be >= 1");
    HIPCHK(hipSetDevice(c->dev));
    for (int i = 0; i < nsets; ++i)
        if (!sets[i]) return kh_fail(KH_E_ARG, "kh_union_histogram: operand %d is NULL", i);
    if (nsets > KH_MAX_INPUT_SETS) {   // beyond one launch's fan-in: through the general path
        kh_set* u = nullptr;
        KHCHK(kh_union_sum(c, sets, nsets, cs, &u, hist, hist_len));
        kh_set_free(u);
        return KH_OK;
    }
    std::vector<const kh_set*> in(sets, sets + nsets);
    return run_setop(c, in, KH_OP_UNION, KH_OC_SUM, cs, nullptr, hist, hist_len);
}

extern "C" int kh_simple(kh_ctx* c, const kh_set* a, const kh_set* b, int op, int mode, uint32_t cs,
                         kh_set** out) {
    if (!c || !a || !b || !out) return kh_fail(KH_E_ARG, "kh_simple: NULL argument");
    if (op < KH_UNION || op > KH_COUNTERS_SUBTRACT) return kh_fail(KH_E_ARG, "unknown set operation %d", op);
    if (mode < KH_MODE_MIN || mode > KH_MODE_RIGHT) return kh_fail(KH_E_ARG, "unknown counter mode %d", mode);
    if (cs < 1) return kh_fail(KH_E_ARG, "cs must be >= 1");
    HIPCHK(hipSetDevice(c->dev));
    std::vector<const kh_set*> in{a, b};
    return run_setop(c, in, op, mode, cs, out, nullptr, 0);
}

// ------------------------------------------------------------------------------ histogram
extern "C" int kh_histogram(kh_ctx* c, const kh_set* s, uint64_t* hist, uint32_t hist_len) {
    if (!c || !s || !hist || hist_len < 2) return kh_fail(KH_E_ARG, "kh_histogram: bad argument");
    HIPCHK(hipSetDevice(c->dev));
    memset(hist, 0, 8 * (size_t)hist_len);
    if (!s->n) return KH_OK;
    if (!s->cb) {
        hist[std::min<u32>(s->uniform, hist_len - 1)] = s->n;
        return KH_OK;
    }
    Tmp d_hist;
    TMP_ALLOC(d_hist, c, 8 * (u64)hist_len);
    HIPCHK(hipMemsetAsync(d_hist.b->p, 0, 8 * (u64)hist_len, c->st));
    c->prof_begin(KC_HISTOGRAM);
    kh_launch_histogram(s->counts_ptr(), s->n, d_hist.as<unsigned long long>(), hist_len, c->st);
    c->prof_end();
    HIPCHK(hipMemcpyAsync(hist, d_hist.b->p, 8 * (size_t)hist_len, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    return KH_OK;
}

// ------------------------------------------------------------------- occurrence table
// Small-k variant of the across-group step: cell v of a 4^k-cell table in
// device memory counts the sets that hold canonical k-mer v.
extern "C" int kh_table_add_set(kh_ctx* c, const kh_set* s, void* d_table, uint32_t cell_bytes) {
    if (!c || !s || !d_table) return kh_fail(KH_E_ARG, "kh_table_add_set: NULL argument");
    if (cell_bytes != 1 && cell_bytes != 4) return kh_fail(KH_E_ARG, "kh_table_add_set: cell_bytes must be 1 or 4");
    if (s->k > KH_TABLE_MAX_K) return kh_fail(KH_E_ARG, "kh_table_add_set: k > 16 has no direct-addressed table");
    HIPCHK(hipSetDevice(c->dev));
    if (!s->n) return KH_OK;
    c->prof_begin(KC_HISTOGRAM);
    kh_launch_table_add(s->keys_ptr(), s->n, s->k, d_table, cell_bytes, c->st);
    c->prof_end();
    HIPCHK(hipGetLastError());
    return KH_OK;
}

extern "C" int kh_table_histogram(kh_ctx* c, const void* d_table, uint32_t cell_bytes, uint64_t lo,
                                  uint64_t hi, uint32_t cs, uint64_t* hist, uint32_t hist_len) {
    if (!c || !d_table || !hist || hist_len < 2) return kh_fail(KH_E_ARG, "kh_table_histogram: bad argument");
    if (cell_bytes != 1 && cell_bytes != 4) return kh_fail(KH_E_ARG, "kh_table_histogram: cell_bytes must be 1 or 4");
    if (hi < lo || (lo * cell_bytes) % 16 || ((uintptr_t)d_table % 16))
        return kh_fail(KH_E_ARG, "kh_table_histogram: the range must start on a 16-byte boundary");
    HIPCHK(hipSetDevice(c->dev));
    memset(hist, 0, 8 * (size_t)hist_len);
    if (hi == lo) return KH_OK;
    Tmp d_hist;
    TMP_ALLOC(d_hist, c, 8 * (u64)hist_len);
    HIPCHK(hipMemsetAsync(d_hist.b->p, 0, 8 * (u64)hist_len, c->st));
    c->prof_begin(KC_HISTOGRAM);
    kh_launch_table_hist(d_table, cell_bytes, lo, hi, cs ? cs : 0xFFFFFFFFu, d_hist.as<unsigned long long>(),
                         hist_len, c->st);
    c->prof_end();
    HIPCHK(hipMemcpyAsync(hist, d_hist.b->p, 8 * (size_t)hist_len, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    return KH_OK;
}

// ------------------------------------------------------------------- membership matrix
// Experiment type 4 (src/merge_lists.py:14-33,101-141): which of `sets` hold each pivot k-mer.
// Device: one search per (pivot key, set).  Host: the records are put in the order the
// reference walks them — its dict is filled from the `dump -s` text, i.e. ascending canonical
// key — because the floating-point row sums below depend on the order of the additions.
namespace {
struct KeyIdx { u64 key; u32 idx; };

// ascending order of n canonical keys; W == 1: LSD radix over the 2k significant bits
void canonical_order(int W, int k, const u64* keys, u64 n, std::vector<u32>& order) {
    order.resize(n);
    if (W == 2) {
        for (u64 i = 0; i < n; ++i) order[i] = (u32)i;
        std::sort(order.begin(), order.end(), [&](u32 a, u32 b) {
            return keys[2 * (u64)a + 1] != keys[2 * (u64)b + 1] ? keys[2 * (u64)a + 1] < keys[2 * (u64)b + 1]
                                                                : keys[2 * (u64)a] < keys[2 * (u64)b];
        });
        return;
    }
    std::vector<KeyIdx> a(n), b(n);
    for (u64 i = 0; i < n; ++i) a[i] = KeyIdx{keys[i], (u32)i};
    const int bits = 2 * k;
    for (int sh = 0; sh < bits; sh += 11) {
        size_t cnt[2049] = {0};
        for (u64 i = 0; i < n; ++i) ++cnt[((a[i].key >> sh) & 2047) + 1];
        for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
        for (u64 i = 0; i < n; ++i) b[cnt[(a[i].key >> sh) & 2047]++] = a[i];
        a.swap(b);
    }
    for (u64 i = 0; i < n; ++i) order[i] = a[i].idx;
}

struct Membership {
    std::vector<u64> keys;     // canonical, as stored (mixed order)
    std::vector<u32> counts;
    std::vector<u64> masks;
    std::vector<u32> order;    // ascending canonical key
    u32 nwords = 0;
};

int membership_compute(kh_ctx* c, const kh_set* pivot, const kh_set* const* sets, int nsets, Membership& m) {
    if (!c || !pivot || (nsets > 0 && !sets) || nsets < 0) return kh_fail(KH_E_ARG, "membership: bad argument");
    if (pivot->n >> 32) return kh_fail(KH_E_ARG, "membership: pivot sets of 2^32 k-mers or more are not supported");
    for (int i = 0; i < nsets; ++i) {
        if (!sets[i]) return kh_fail(KH_E_ARG, "membership: NULL set");
        if (sets[i]->k != pivot->k) return kh_fail(KH_E_KMISMATCH, "membership: k differs (%d vs %d)", sets[i]->k, pivot->k);
    }
    HIPCHK(hipSetDevice(c->dev));
    const u64 n = pivot->n;
    const int W = pivot->W;
    m.nwords = (u32)std::max(1, (nsets + 63) / 64);
    m.keys.assign(n * W, 0);
    m.counts.assign(n, pivot->uniform);
    m.masks.assign(n * m.nwords, 0);
    if (!n) { m.order.clear(); return KH_OK; }
    std::vector<KhSetView> v((size_t)std::max(nsets, 1));
    for (int i = 0; i < nsets; ++i)
        v[i] = KhSetView{sets[i]->n ? sets[i]->keys_ptr() : nullptr, nullptr, sets[i]->n, 1, 0};
    Tmp d_view, d_masks, d_keys;
    TMP_ALLOC(d_view, c, sizeof(KhSetView) * v.size());
    TMP_ALLOC(d_masks, c, 8 * n * m.nwords);
    TMP_ALLOC(d_keys, c, 8 * (u64)W * n);
    HIPCHK(hipMemcpyAsync(d_view.b->p, v.data(), sizeof(KhSetView) * v.size(), hipMemcpyHostToDevice, c->st));
    c->prof_begin(KC_SETOP);
    kh_launch_membership(W, pivot->keys_ptr(), n, d_view.as<KhSetView>(), (u32)nsets, pivot->k, m.nwords,
                         d_masks.as<u64>(), c->st);
    c->prof_end();
    kh_launch_unmix(W, pivot->keys_ptr(), d_keys.b->p, n, pivot->k, c->st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(m.masks.data(), d_masks.b->p, 8 * n * m.nwords, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipMemcpyAsync(m.keys.data(), d_keys.b->p, 8 * (u64)W * n, hipMemcpyDeviceToHost, c->st));
    if (pivot->cb)
        HIPCHK(hipMemcpyAsync(m.counts.data(), pivot->counts_ptr(), 4 * n, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    canonical_order(W, pivot->k, m.keys.data(), n, m.order);
    return KH_OK;
}
}  // namespace

extern "C" int kh_membership(kh_ctx* c, const kh_set* pivot, const kh_set* const* sets, int nsets,
                             uint64_t* keys_out, uint32_t* counts_out, uint64_t* masks_out) {
    Membership m;
    KHCHK(membership_compute(c, pivot, sets, nsets, m));
    const int W = pivot->W;
    for (u64 r = 0; r < pivot->n; ++r) {
        const u64 i = m.order[r];
        if (keys_out) for (int w = 0; w < W; ++w) keys_out[r * W + w] = m.keys[i * W + w];
        if (counts_out) counts_out[r] = m.counts[i];
        if (masks_out) for (u32 w = 0; w < m.nwords; ++w) masks_out[r * m.nwords + w] = m.masks[i * m.nwords + w];
    }
    return KH_OK;
}

// One row of the feature-level confusion matrix, exactly as src/merge_lists.py:122-141 adds it
// up: k-mers in dump order; for a k-mer with count c held by the sets M (ascending index, the
// order update_dictionary appends them, :26-33), row[m] += 1 / len(M) * c for m in M;
// unique_pivot_count = sum of c over the k-mers no set holds (:122-126).
extern "C" int kh_confusion_row(kh_ctx* c, const kh_set* pivot, const kh_set* const* sets, int nsets,
                                double* row, uint64_t* unique_pivot_count) {
#pragma clang fp contract(off)
    if (!row || !unique_pivot_count) return kh_fail(KH_E_ARG, "kh_confusion_row: NULL output");
    Membership m;
    KHCHK(membership_compute(c, pivot, sets, nsets, m));
    for (int d = 0; d < nsets; ++d) row[d] = 0.0;
    u64 uniq = 0;
    for (u64 r = 0; r < pivot->n; ++r) {
        const u64 i = m.order[r];
        int len = 0;
        for (u32 w = 0; w < m.nwords; ++w) len += __builtin_popcountll(m.masks[i * m.nwords + w]);
        if (!len) { uniq += m.counts[i]; continue; }
        const double share = 1.0 / (double)len;              // Python: 1 / len(matches)
        const double add = share * (double)m.counts[i];      //         ... * count
        for (u32 w = 0; w < m.nwords; ++w) {
            u64 bits = m.masks[i * m.nwords + w];
            while (bits) {
                const int d = __builtin_ctzll(bits);
                bits &= bits - 1;
                volatile double sum = row[w * 64 + d] + add;
                row[w * 64 + d] = sum;
            }
        }
    }
    *unique_pivot_count = uniq;
    return KH_OK;
}

// ------------------------------------------------------------------------------ transfer
extern "C" int kh_set_download(kh_ctx* c, const kh_set* s, uint64_t* keys, uint32_t* counts) {
    if (!c || !s) return kh_fail(KH_E_ARG, "kh_set_download: NULL argument");
    HIPCHK(hipSetDevice(c->dev));
    if (!s->n) return KH_OK;
    const size_t kb = 8 * (size_t)s->W;
    if (keys) {
        Tmp d_tmp;
        TMP_ALLOC(d_tmp, c, kb * s->n);
        c->prof_begin(KC_REMIX);
        kh_launch_unmix(s->W, s->keys_ptr(), d_tmp.b->p, s->n, s->k, c->st);
        c->prof_end();
        HIPCHK(hipMemcpyAsync(keys, d_tmp.b->p, kb * s->n, hipMemcpyDeviceToHost, c->st));
        HIPCHK(hipStreamSynchronize(c->st));
    }
    if (counts) {
        if (s->cb) {
            HIPCHK(hipMemcpyAsync(counts, s->counts_ptr(), 4 * s->n, hipMemcpyDeviceToHost, c->st));
            HIPCHK(hipStreamSynchronize(c->st));
        } else {
            std::fill(counts, counts + s->n, s->uniform);
        }
    }
    return KH_OK;
}

int kh_set_from_mixed_host(kh_ctx* c, int k, u64 n, const void* keys_mixed_sorted, const u32* counts,
                           u32 uniform, u32 counter_max, kh_set** out) {
    const int W = k <= 32 ? 1 : 2;
    const size_t kb = 8 * (size_t)W;
    if (!n) { *out = make_set(k, 0, nullptr, 0, nullptr, 0, uniform, counter_max); return KH_OK; }
    DevBuf* kbuf = c->buf_alloc(kb * n);
    DevBuf* cbuf = counts ? c->buf_alloc(4 * n) : nullptr;
    struct Guard { DevBuf *a, *b; ~Guard() { buf_unref(a); buf_unref(b); } } guard{kbuf, cbuf};
    if (!kbuf || (counts && !cbuf)) return kh_fail(KH_E_NOMEM, "device allocation failed (upload)");
    HIPCHK(hipMemcpyAsync(kbuf->p, keys_mixed_sorted, kb * n, hipMemcpyHostToDevice, c->st));
    if (counts) HIPCHK(hipMemcpyAsync(cbuf->p, counts, 4 * n, hipMemcpyHostToDevice, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    buf_ref(kbuf);
    if (cbuf) buf_ref(cbuf);
    *out = make_set(k, n, kbuf, 0, cbuf, 0, uniform, counter_max);
    return KH_OK;
}

extern "C" int kh_set_upload(kh_ctx* c, int k, uint64_t n, const uint64_t* keys, const uint32_t* counts,
                             kh_set** out) {
    if (!c || !out || (n && !keys)) return kh_fail(KH_E_ARG, "kh_set_upload: bad argument");
    KHCHK(check_k(k));
    HIPCHK(hipSetDevice(c->dev));
    const int W = k <= 32 ? 1 : 2;
    // order by mixed key on the host (upload is a test / interchange path, not the hot path)
    std::vector<u64> mixed((size_t)n * W);
    for (u64 i = 0; i < n; ++i) kh_mix_host(k, keys + i * W, mixed.data() + i * W);
    std::vector<u64> idx(n);
    for (u64 i = 0; i < n; ++i) idx[i] = i;
    if (W == 1)
        std::sort(idx.begin(), idx.end(), [&](u64 a, u64 b) { return mixed[a] < mixed[b]; });
    else
        std::sort(idx.begin(), idx.end(), [&](u64 a, u64 b) {
            return mixed[2 * a + 1] < mixed[2 * b + 1] ||
                   (mixed[2 * a + 1] == mixed[2 * b + 1] && mixed[2 * a] < mixed[2 * b]);
        });
    std::vector<u64> sk((size_t)n * W);
    std::vector<u32> sc(counts ? n : 0);
    for (u64 i = 0; i < n; ++i) {
        for (int w = 0; w < W; ++w) sk[i * W + w] = mixed[idx[i] * W + w];
        if (counts) sc[i] = counts[idx[i]];
        if (i && memcmp(&sk[i * W], &sk[(i - 1) * W], 8 * W) == 0)
            return kh_fail(KH_E_ARG, "kh_set_upload: keys are not distinct");
    }
    return kh_set_from_mixed_host(c, k, n, sk.data(), counts ? sc.data() : nullptr, 1, KH_KMC_DEFAULT_CS, out);
}

extern "C" int kh_set_from_device(kh_ctx* c, int k, uint64_t n, const void* keys_mixed, const uint32_t* counts,
                                  kh_set** out) {
    if (!c || !out || (n && !keys_mixed)) return kh_fail(KH_E_ARG, "kh_set_from_device: bad argument");
    KHCHK(check_k(k));
    HIPCHK(hipSetDevice(c->dev));
    const int W = k <= 32 ? 1 : 2;
    const size_t kb = 8 * (size_t)W;
    if (!n) { *out = make_set(k, 0, nullptr, 0, nullptr, 0, 1); return KH_OK; }
    DevBuf* kbuf = c->buf_alloc(kb * n);
    DevBuf* cbuf = counts ? c->buf_alloc(4 * n) : nullptr;
    struct Guard { DevBuf *a, *b; ~Guard() { buf_unref(a); buf_unref(b); } } guard{kbuf, cbuf};
    if (!kbuf || (counts && !cbuf)) return kh_fail(KH_E_NOMEM, "device allocation failed");
    HIPCHK(hipMemcpyAsync(kbuf->p, keys_mixed, kb * n, hipMemcpyDeviceToDevice, c->st));
    if (counts) HIPCHK(hipMemcpyAsync(cbuf->p, counts, 4 * n, hipMemcpyDeviceToDevice, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    buf_ref(kbuf);
    if (cbuf) buf_ref(cbuf);
    *out = make_set(k, n, kbuf, 0, cbuf, 0, 1);
    return KH_OK;
}

extern "C" int kh_set_export_range(kh_ctx* c, const kh_set* s, uint64_t lo, uint64_t hi, void* keys_out,
                                   uint32_t* counts_out) {
    if (!c || !s) return kh_fail(KH_E_ARG, "kh_set_export_range: NULL argument");
    if (lo > hi || hi > s->n) return kh_fail(KH_E_ARG, "kh_set_export_range: [%llu,%llu) outside [0,%llu)",
                                             (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)s->n);
    HIPCHK(hipSetDevice(c->dev));
    const u64 n = hi - lo;
    if (!n) return KH_OK;
    const size_t kb = 8 * (size_t)s->W;
    if (keys_out)
        HIPCHK(hipMemcpyAsync(keys_out, static_cast<const u8*>(s->keys_ptr()) + lo * kb, kb * n,
                              hipMemcpyDeviceToDevice, c->st));
    if (counts_out) {
        if (s->cb) HIPCHK(hipMemcpyAsync(counts_out, s->counts_ptr() + lo, 4 * n, hipMemcpyDeviceToDevice, c->st));
        else kh_launch_fill_u32(counts_out, n, s->uniform, c->st);
    }
    return KH_OK;   // stream-ordered: call kh_sync before another stream reads the buffers
}
extern "C" int kh_set_export_device(kh_ctx* c, const kh_set* s, void* keys_out, uint32_t* counts_out) {
    if (!c || !s) return kh_fail(KH_E_ARG, "kh_set_export_device: NULL argument");
    int r = kh_set_export_range(c, s, 0, s->n, keys_out, counts_out);
    if (r != KH_OK) return r;
    HIPCHK(hipStreamSynchronize(c->st));
    return KH_OK;
}
// zero-copy view of caller-owned device arrays (mixed keys ascending, distinct); the caller
// keeps them alive and unchanged for the life of the handle
extern "C" int kh_set_wrap_device(kh_ctx* c, int k, uint64_t n, const void* keys_mixed, const uint32_t* counts,
                                  uint32_t uniform, kh_set** out) {
    if (!c || !out || (n && !keys_mixed)) return kh_fail(KH_E_ARG, "kh_set_wrap_device: bad argument");
    KHCHK(check_k(k));
    if (!n) { *out = make_set(k, 0, nullptr, 0, nullptr, 0, uniform ? uniform : 1); return KH_OK; }
    auto borrow = [&](const void* p, size_t bytes) {
        DevBuf* b = new DevBuf;
        b->p = const_cast<void*>(p);
        b->bytes = bytes;
        b->refs = 1;
        b->ctx = nullptr;   // not owned: never returned to the pool
        return b;
    };
    const int W = k <= 32 ? 1 : 2;
    DevBuf* kbuf = borrow(keys_mixed, 8 * (size_t)W * n);
    DevBuf* cbuf = counts ? borrow(counts, 4 * n) : nullptr;
    *out = make_set(k, n, kbuf, 0, cbuf, 0, uniform ? uniform : 1);
    return KH_OK;
}

extern "C" int kh_set_partition_bounds(kh_ctx* c, const kh_set* s, uint32_t nparts, uint64_t* bounds) {
    if (!c || !s || !bounds || !nparts) return kh_fail(KH_E_ARG, "kh_set_partition_bounds: bad argument");
    HIPCHK(hipSetDevice(c->dev));
    if (!s->n) { for (u32 i = 0; i <= nparts; ++i) bounds[i] = 0; return KH_OK; }
    KhSetView v{s->keys_ptr(), nullptr, s->n, 1, 0};
    Tmp d_view, d_bounds;
    TMP_ALLOC(d_view, c, sizeof v);
    TMP_ALLOC(d_bounds, c, 8 * ((u64)nparts + 1));
    HIPCHK(hipMemcpyAsync(d_view.b->p, &v, sizeof v, hipMemcpyHostToDevice, c->st));
    kh_launch_range_bounds(s->W, d_view.as<KhSetView>(), 1, nparts, s->k, d_bounds.as<u64>(), nullptr, 0, c->st);
    HIPCHK(hipMemcpyAsync(bounds, d_bounds.b->p, 8 * ((u64)nparts + 1), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    return KH_OK;
}

extern "C" int kh_sets_partition_bounds(kh_ctx* c, const kh_set* const* sets, int nsets, uint32_t nparts,
                                        uint64_t* bounds) {
    if (!c || !sets || !bounds || !nparts || nsets <= 0) return kh_fail(KH_E_ARG, "kh_sets_partition_bounds: bad argument");
    HIPCHK(hipSetDevice(c->dev));
    const int k = sets[0]->k, W = sets[0]->W;
    std::vector<KhSetView> v(nsets);
    for (int i = 0; i < nsets; ++i) {
        if (!sets[i] || sets[i]->k != k) return kh_fail(KH_E_KMISMATCH, "sets built with different k");
        v[i] = KhSetView{sets[i]->n ? sets[i]->keys_ptr() : nullptr, nullptr, sets[i]->n, 1, 0};
    }
    Tmp d_view, d_bounds;
    const u64 nb = ((u64)nparts + 1) * nsets;
    TMP_ALLOC(d_view, c, sizeof(KhSetView) * nsets);
    TMP_ALLOC(d_bounds, c, 8 * nb);
    HIPCHK(hipMemcpyAsync(d_view.b->p, v.data(), sizeof(KhSetView) * nsets, hipMemcpyHostToDevice, c->st));
    kh_launch_range_bounds(W, d_view.as<KhSetView>(), nsets, nparts, k, d_bounds.as<u64>(), nullptr, 0, c->st);
    HIPCHK(hipMemcpyAsync(bounds, d_bounds.b->p, 8 * nb, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    return KH_OK;
}

// ------------------------------------------------------------------------------ fused exp 1
// A group whose genomes do not fit the wave budget together ("HBM spill"):
// its genomes are built in sub-waves, each summed into a running union that carries counters
// (no saturation until the end), so the memory in flight is one sub-wave + the union.  The
// final pass applies `cs` and takes the histogram.  Same result as the one-wave path.
static int group_union_incremental(kh_ctx* c, const std::vector<int>& members, const uint8_t* const* seqs,
                                   const uint64_t* lens, int on_device, int k, u32 cs, u64 budget,
                                   uint64_t* hist, u32 hist_len, uint64_t* distinct_per_seq,
                                   kh_set** out_union) {
    kh_set* running = nullptr;
    std::vector<kh_set*> wsets;
    auto cleanup = [&]() {
        for (auto* s : wsets) kh_set_free(s);
        wsets.clear();
        kh_set_free(running);
        running = nullptr;
    };
    size_t i0 = 0;
    while (i0 < members.size()) {
        size_t i1 = i0;
        u64 acc = 0;
        while (i1 < members.size() && (i1 == i0 || acc + lens[members[i1]] <= budget) &&
               i1 - i0 < (size_t)KH_MAX_INPUT_SETS - 1)
            acc += lens[members[i1++]];
        std::vector<const uint8_t*> wseqs(i1 - i0);
        std::vector<uint64_t> wlens(i1 - i0);
        wsets.assign(i1 - i0, nullptr);
        for (size_t j = i0; j < i1; ++j) { wseqs[j - i0] = seqs[members[j]]; wlens[j - i0] = lens[members[j]]; }
        int r = kh_build_batch(c, (int)(i1 - i0), wseqs.data(), wlens.data(), on_device, k, 1, KH_NO_MAX,
                               KH_KMC_DEFAULT_CS, 0, wsets.data());
        if (r != KH_OK) { cleanup(); return r; }
        std::vector<const kh_set*> in;
        if (running) in.push_back(running);
        for (size_t j = i0; j < i1; ++j) {
            if (distinct_per_seq) distinct_per_seq[members[j]] = wsets[j - i0]->n;
            in.push_back(wsets[j - i0]);
        }
        kh_set* next = nullptr;
        r = run_setop(c, in, KH_OP_UNION, KH_OC_SUM, 0x7fffffffu, &next, nullptr, 0);
        if (r != KH_OK) { cleanup(); return r; }
        for (auto* s : wsets) kh_set_free(s);
        wsets.clear();
        kh_set_free(running);
        running = next;
        i0 = i1;
    }
    std::vector<const kh_set*> in{running};
    int r = run_setop(c, in, KH_OP_UNION, KH_OC_SUM, cs, out_union, hist, hist_len);
    cleanup();
    return r;
}

// The super-k-mer form of the fused path (kh_skm.hip): bases -> 16-byte records of consecutive k-mers that
// share their minimizer slot -> two counting-sort levels (coarse bucket, slot) -> one LDS hash set per
// slot.  Takes what the key-array form below takes when k is in [KH_SKM_MIN_K, KH_SKM_MAX_K], nothing is
// emitted and the batch fits one slot grid; *done == false: not applicable, or a region overflowed
// (low-complexity input, far more records than estimated) — the caller goes on to the key-array form.
// Minimizer length of the one-word super-k-mer form.  A longer window (shorter minimizer) makes longer runs: fewer,
// longer records for all three kernels — as long as the 4^m minimizers still spread over the slots (m = 11 overfills
// too many).  Measured on the headline shape (ms per step, m = 15 / 13 / 12): k = 21 3.40 / 2.95 / 2.87, k = 24
// 2.84 / 2.60 / 2.57, k = 27 2.53 / 2.47 / 2.51; from k = 28 the record's 24 .. 27 k-mers are the limit, not the window
// (k = 30 .. 32: no difference): m = 16 where that makes the window a power of two, else 15.
static int skm_minimizer_len(int k) {
    if (k <= 24) return k >= 17 ? 12 : 11;   // k = 17 .. 24: 12 (k = 17: windows of 6, 3.75 ms against 4.6 with 11 bases and 4.4 with key
                                             // arrays); the kernels alone at k = 15, 16: 11
    if (k <= 27) return 13;
    const int m15w = k - 15 + 1;                                   // m-mers per k-mer with m = 15
    return (m15w > 1 && ((m15w - 1) & (m15w - 2)) == 0) ? 16 : 15;
}
// by_group: the operands of the union are the GROUPS (every record carries its genome's group number): only the
// across-group histogram comes out — the second pass of a run over more than 64 genomes, whose batches of
// whole groups have answered the within-group questions.
// records_only (the exchange form of the multi-GPU step): stop behind the regroup and hand the records by slot out —
// with force_slots slots, the number all ranks agreed on (slots are a global function of the minimizer).
struct SkmRecords {
    u32 force_slots = 0;
    DevBuf* reg2 = nullptr;     // out: [nslots][cap2] records
    DevBuf* ws = nullptr;       // out: workspace that holds cur2
    const u32* cur2 = nullptr;  // out: [nslots] records per slot
    u32 nslots = 0, cap2 = 0;
    u64 records = 0;            // out: records written
    u32 fan_hint = 0;           // in: genomes whose copies of a locus arrive together (tags of ONE group: the sub-batch size)
    std::vector<u64>* inst = nullptr;   // out (if set): k-mer instances per sequence, in the caller's order
    bool want_spill = false;    // in: records that did not fit their slot's region are handed out too (else: a failure)
    DevBuf* spill = nullptr;    // out: [spill_cap] records, then [spill_cap] u32 slots
    u32 spill_n = 0, spill_cap = 0;
    ~SkmRecords() { buf_unref(reg2); buf_unref(ws); buf_unref(spill); }
};
static int exp1_skm(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens, int on_device,
                    const int* group_of, int ngroups, int k, u32 cs, uint64_t* within_hist,
                    uint64_t* across_hist, u32 hist_len, uint64_t* distinct_per_seq, bool* done, bool by_group = false,
                    SkmRecords* rec_out = nullptr) {
    *done = false;
    {
        // k = 17 .. 19 on the headline shape: 3.75 / 3.6 / 3.3 ms against 4.4 with key arrays; k = 16 and 15 (windows of 5
        // or 6 over 11 bases: too many overfull slots; over 12: twice the coarse buckets) 5.2 / 6.7 ms: below 17 the key
        // arrays stay (KHOICE_SKM_MIN_K: experiments and tests, the kernels take k >= 15)
        int min_k = KH_SKM_MIN_K;
        if (const char* e = getenv("KHOICE_SKM_MIN_K")) min_k = std::max(15, atoi(e));
        if (k < min_k || k > KH_SKM2_MAX_K || getenv("KHOICE_NO_SKM")) return KH_OK;
    }
    const bool two = k > KH_SKM_MAX_K;   // two-word keys: 32-byte records, kh_skm2.hip
    if (two && getenv("KHOICE_NO_SKM2")) return KH_OK;
    if ((!by_group && nseq > KH_TAG_MAX_OPS) || ngroups > KH_TAG_MAX_OPS) return KH_OK;
    std::vector<int> gsize(ngroups, 0), gstart(ngroups + 1, 0), perm(nseq);
    for (int i = 0; i < nseq; ++i) gsize[group_of[i]]++;
    u32 nbins = 0;
    std::vector<u32> bin0(ngroups);
    for (int g = 0; g < ngroups; ++g) {
        if (!gsize[g]) return kh_fail(KH_E_ARG, "group %d has no sequences", g);
        gstart[g + 1] = gstart[g] + gsize[g];
        bin0[g] = nbins;
        nbins += (by_group ? 1u : (u32)gsize[g]) + 1;   // by_group: every operand is a group of its own
    }
    const u32 abase = nbins;
    nbins += (u32)ngroups + 1;
    if (nbins > (u32)KH_TAG_MAX_BINS) return KH_OK;
    {
        std::vector<int> at(gstart.begin(), gstart.end() - 1);
        for (int i = 0; i < nseq; ++i) perm[at[group_of[i]]++] = i;
    }
    // ---- geometry: minimizer length, slots, regions
    int m;
    u32 w, nmax;
    if (!two) {
        m = skm_minimizer_len(k);
        if (const char* e = getenv("KHOICE_SKM_M")) m = std::min(16, std::max(2, atoi(e)));   // experiments
        if (m >= k) return KH_OK;
        w = (u32)(k - m + 1);
        nmax = (u32)std::min(31, 55 - k);
        if (!kh_skm_supports_w(w)) return KH_OK;
    } else {   // the scatter exists for every third window width: one of m = 16, 15, 14 fits
        m = 16;
        if (const char* e = getenv("KHOICE_SKM2_M")) m = std::min(16, std::max(10, atoi(e)));   // experiments: the first length tried
        const int m_lo = m - 2;
        while (m >= m_lo && !kh_skm2_supports_w((u32)(k - m + 1))) --m;
        if (m < m_lo) return KH_OK;
        w = (u32)(k - m + 1);
        nmax = (u32)std::min(63, 118 - k);
    }
    u64 total_pos = 0, bases = 0, seq_bytes = 0;
    for (int i = 0; i < nseq; ++i) {
        total_pos += lens[i] >= (u64)k ? lens[i] - k + 1 : 0;
        bases += lens[i];
    }
    if (!total_pos) return KH_OK;
    // k-mer instances per slot: the hash set takes T = 4096 (2048 for two-word keys) per round.  The genomes of a
    // group put their copies of a locus in the same slot, a minimizer run at a time: clumps of c = (w + 1) / 2 x
    // (largest group) instances, so sigma = sqrt(mean x c); mean + 2.5 sigma = T  (measured: 2500 / 2900 / 3200 /
    // 3500 / 3800 at k = 31 with groups of five: union 1.96 / 1.81 / 1.75 / 1.80 / 1.94 ms; the rule gives 3180)
    // entries of the union's hash set (one-word keys): 4096 with 1024 threads (two workgroups per CU) or 2048 with 512
    // (four per CU, half-size slots)
    u32 table = 4096;
    if (const char* e = getenv("KHOICE_SKM_TABLE")) table = atoi(e) == 2048 ? 2048u : (atoi(e) == 2560 ? 2560u : 4096u);
    u32 mean;
    double clump = 1.0;   // instances that land in a slot together
    {
        u32 fan = 1;
        for (int g = 0; g < ngroups; ++g) fan = std::max<u32>(fan, (u32)gsize[g]);
        if (rec_out && rec_out->fan_hint) fan = std::max(fan, rec_out->fan_hint);
        const double T = two ? (double)kh_skm2_table() : (double)table, cl = 0.5 * (double)(w + 1) * (double)fan;
        const double r = 0.5 * (-2.5 * std::sqrt(cl) + std::sqrt(6.25 * cl + 4.0 * T));
        mean = (u32)std::max(256.0, r * r);
        clump = cl;
    }
    if (const char* e = getenv("KHOICE_SKM_MEAN")) mean = std::max<u32>(64, (u32)strtoul(e, nullptr, 10));
    // records: a run of k-mers with one minimizer is (w + 1) / 2 long on average; cuts at the waves' 2048
    // positions, at boundaries a run may not cross (a second thread boundary, nmax) and at invalid bases add
    // a little (measured: 0.120 records per k-mer at w = 16, 0.27 at w = 7)
    const double per_kmer = 2.0 / (double)(w + 1) + 1.0 / 48.0;
    // a slot's records vary like its instances: sigma / mean = sqrt(clump / mean) (12 % at k = 31 with groups of five,
    // 36 % with ten genomes per group and two-word keys): five sigma, at least 1.7
    double slack1 = 1.25, slack2 = 1.7;
    const u32 max_cap2 = two ? kh_skm2_max_cap2() : kh_skm_union_max_cap2(table);
    for (int it = 0; it < 8; ++it) {   // short windows (k = 20 .. 22: 3.5 k-mers per record): fewer instances per slot so that its records fit
        slack2 = std::max(1.7, 1.0 + 5.0 * std::sqrt(clump / (double)mean));
        const double c2 = (double)mean * per_kmer * slack2 + 96 + 16;
        if (c2 <= (double)max_cap2 || mean <= 256) break;
        mean = std::max<u32>(256, (u32)((double)mean * (double)max_cap2 / c2 * 0.98));
    }
    u64 nslots64 = rec_out && rec_out->force_slots ? rec_out->force_slots : std::max<u64>(1, (total_pos + mean - 1) / mean);
    // One-word keys a little above 256 x 512 slots (short windows on the headline shape: k = 20): 512 coarse buckets
    // double the scatter's cursors and LDS (k = 20: scatter 2.3 ms against 1.0 at k = 21).  Since a slot's region may
    // overflow (side list, k_skm_big) its slack can be cut instead: stay at 256 x 512 slots with slack down to 1.4.
    if (!two && !(rec_out && rec_out->force_slots) && nslots64 > (u64)KH_SKM_MAX_COARSE * KH_SKM_MAX_FINE) {
        const u64 lim = (u64)KH_SKM_MAX_COARSE * KH_SKM_MAX_FINE;
        const double per_slot = (double)total_pos * per_kmer / (double)lim;
        const double s2 = ((double)max_cap2 - 112.0) / per_slot;
        double s2_min = 1.4;
        if (const char* e = getenv("KHOICE_SKM_MIN_SLACK")) s2_min = std::max(1.0, atof(e));   // experiments
        if (s2 >= s2_min) { nslots64 = lim; slack2 = std::min(slack2, s2); }
    }
    // two-word keys a little above 512 x 1024 slots (configs[2] at k = 41, the pass by group over 500 M positions):
    // the same cut, up to what the union's table takes in one go on average
    if (two && !(rec_out && rec_out->force_slots) && nslots64 > (u64)KH_SKM2_MAX_COARSE * KH_SKM2_MAX_FINE) {
        const u64 lim = (u64)KH_SKM2_MAX_COARSE * KH_SKM2_MAX_FINE;
        const double per_slot = (double)total_pos * per_kmer / (double)lim;
        const double s2 = ((double)max_cap2 - 112.0) / per_slot;
        if (s2 >= 1.4 && (double)total_pos / (double)lim <= 0.75 * (double)kh_skm2_table()) { nslots64 = lim; slack2 = std::min(slack2, s2); }
    }
    // coarse buckets: 256 keep the scatter's runs long; inputs past 256 x 512 slots (> 400 M k-mers) take 512
    const u32 max_coarse = (two || nslots64 > (u64)KH_SKM_MAX_COARSE * KH_SKM_MAX_FINE) ? KH_SKM2_MAX_COARSE : KH_SKM_MAX_COARSE;
    if (nslots64 > (u64)max_coarse * (two ? KH_SKM2_MAX_FINE : KH_SKM_MAX_FINE)) return KH_OK;
    const u32 nslots = (u32)nslots64;
    const u32 S = std::max<u32>(1, (nslots + max_coarse - 1) / max_coarse);
    const u32 nb1 = (nslots + S - 1) / S;
    const double recs = (double)total_pos * per_kmer;
    if (const char* e = getenv("KHOICE_SKM_SLACK")) slack1 = slack2 = std::max(0.01, atof(e));   // below 1: tests of the overflow fall-back
    const u64 cap1_64 = ((u64)(recs / nb1 * slack1) + 2048 + 63) & ~63ull;
    u64 cap2_64 = ((u64)(recs / nslots * slack2) + 96 + 15) & ~15ull;
    if (cap2_64 > max_cap2 && !(rec_out && rec_out->force_slots)) {
        // Large groups: the copies of a locus (one record per genome of the group) arrive in a slot together, so the
        // five-sigma region is larger than the union takes.  The region is cut to what it takes and the slots with a
        // locus too many go the side-list way (k_skm_big); only when there are more of those than the side list
        // holds is the call given to the key arrays.  (Models of how many there will be — loci per slot Poisson, the
        // group's genomes as one clump — were off by 10 x in both directions on synthetic sets: it is tried.)
        u32 fan = 1;
        for (int g = 0; g < ngroups; ++g) fan = std::max<u32>(fan, (u32)gsize[g]);
        if ((double)fan + recs / (double)nslots <= (double)max_cap2 - 112.0) cap2_64 = max_cap2 & ~15u;
    }
    if (cap2_64 > max_cap2 || cap1_64 > 0x7fffffffull) return KH_OK;
    const u32 cap1 = (u32)cap1_64, cap2 = (u32)cap2_64;
    const size_t rec_bytes = two ? 32 : 16;
    const size_t reg1_bytes = rec_bytes * (size_t)nb1 * cap1, reg2_bytes = rec_bytes * (size_t)nslots * cap2;
    HIPCHK(hipSetDevice(c->dev));
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
            reg1_bytes + reg2_bytes + bases + (64u << 20) > free_b + c->pool.cached_bytes)
            return KH_OK;   // the key-range waves of the key-array form handle what does not fit
    }
    hipStream_t st = c->st;

    // ---- segments and tiles (operands in group-major order: the genomes of a group are consecutive mask bits)
    u32 tile_pos = KH_TILE;
    {
        const u64 want_tiles = 8ull * (u64)std::max(1, c->cus);
        while (tile_pos > (u32)KH_SUBTILE && (total_pos + tile_pos - 1) / tile_pos < want_tiles) tile_pos >>= 1;
    }
    std::vector<KhSeg> segs(nseq);
    std::vector<u64> pack_off(nseq);
    std::vector<KhTile> tiles;
    for (int i = 0; i < nseq; ++i) {
        KhSeg& sg = segs[i];
        memset(&sg, 0, sizeof sg);
        const u64 len = lens[perm[i]];
        pack_off[i] = seq_bytes;
        sg.len = len;
        sg.npos = len >= (u64)k ? len - k + 1 : 0;
        sg.ntiles = (u32)((sg.npos + tile_pos - 1) / tile_pos);
        sg.tile_base = (u32)tiles.size();
        for (u32 t = 0; t < sg.ntiles; ++t) tiles.push_back(KhTile{(u32)i, t});
        seq_bytes += (len + 15) & ~15ull;
    }
    seq_bytes += 256;
    const u32 ntiles = (u32)tiles.size();
    // the union is persistent (as many workgroups as fit the chip, each with a histogram of its own)
    const u32 ugrid = std::min<u32>(nslots, (two ? kh_skm2_union_per_cu() : kh_skm_union_per_cu(table)) * (u32)std::max(1, c->cus));
    const u32 reps = ugrid;
    const size_t hist_words = (size_t)reps * nbins;
    // workspace: [hist][ctl: 8 u32][inst: nseq u64][dup: 64 u64][cur1: nb1 u32][cur2: nslots u32] (zeroed) [ginfo: 64 u32]
    const size_t off_ctl = 8 * hist_words, off_inst = off_ctl + 32, off_dup = off_inst + 8 * (size_t)nseq,
                 off_cur1 = off_dup + 8 * 64, off_cur2 = off_cur1 + 4 * (size_t)KH_SKM_CUR1_STRIDE * nb1,
                 off_ginfo = off_cur2 + 4 * (size_t)((nslots + 3) & ~3u), off_tags = off_ginfo + 256,
                 off_segs = off_tags + (((size_t)nseq + 15) & ~(size_t)15), off_tiles = off_segs + sizeof(KhSeg) * nseq,
                 ws_bytes = off_tiles + sizeof(KhTile) * (size_t)std::max<u32>(1, ntiles);   // [ginfo .. tiles]: one upload
    Tmp d_seq, d_ws, d_reg1, d_reg2, d_spill;
    // one-word keys: what a slot's region cannot hold goes to a side list, the slot to a kernel of its own
    const u32 spill_cap = 1u << 20, big_cap = 16384u;
    bool need_pack = false;
    for (int i = 0; i < nseq; ++i)
        if (!(on_device && (reinterpret_cast<uintptr_t>(seqs[perm[i]]) & 15) == 0)) need_pack = true;
    TMP_ALLOC(d_seq, c, need_pack ? seq_bytes : 256);
    TMP_ALLOC(d_ws, c, ws_bytes);
    TMP_ALLOC(d_reg1, c, reg1_bytes);
    TMP_ALLOC(d_reg2, c, reg2_bytes);
    if (spill_cap) TMP_ALLOC(d_spill, c, (size_t)spill_cap * (rec_bytes + 4) + (size_t)big_cap * 4);
    struct PinG { kh_ctx* c; void* p = nullptr; size_t n = 0; ~PinG() { if (p) c->pin_release(p, n); } } pin{c};
    // pinned staging: [segs][tiles][ginfo] up, [hist .. dup] down
    const size_t up_bytes = ws_bytes - off_ginfo;   // the upload, laid out as on the device
    const size_t down_bytes = off_cur1;
    pin.p = c->pin_alloc(up_bytes + down_bytes + 64, &pin.n);
    if (!pin.p) return kh_fail(KH_E_NOMEM, "pinned host allocation failed");
    u8* h_up = static_cast<u8*>(pin.p);
    u8* h_down = h_up + ((up_bytes + 63) & ~(size_t)63);
    c->prof_begin(KC_COPY_IN);
    for (int i = 0; i < nseq; ++i) {
        const uint8_t* src = seqs[perm[i]];
        if (on_device && (reinterpret_cast<uintptr_t>(src) & 15) == 0) { segs[i].seq = src; continue; }
        segs[i].seq = d_seq.as<u8>() + pack_off[i];
        if (!segs[i].len) continue;
        HIPCHK(hipMemcpyAsync(d_seq.as<u8>() + pack_off[i], src, segs[i].len,
                              on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    u32* h_ginfo = reinterpret_cast<u32*>(h_up);
    u8* h_tags = h_up + (off_tags - off_ginfo);
    KhSeg* h_segs = reinterpret_cast<KhSeg*>(h_up + (off_segs - off_ginfo));
    KhTile* h_tiles = reinterpret_cast<KhTile*>(h_up + (off_tiles - off_ginfo));
    memcpy(h_segs, segs.data(), sizeof(KhSeg) * nseq);
    if (ntiles) memcpy(h_tiles, tiles.data(), sizeof(KhTile) * (size_t)ntiles);
    memset(h_ginfo, 0, 256);
    if (by_group) {
        for (int g = 0; g < ngroups; ++g) h_ginfo[g] = (u32)g | (1u << 8) | (bin0[g] << 16);
        for (int i = 0; i < nseq; ++i) h_tags[i] = (u8)group_of[perm[i]];
    } else {
        for (int g = 0; g < ngroups; ++g)
            for (int j = 0; j < gsize[g]; ++j)
                h_ginfo[gstart[g] + j] = (u32)gstart[g] | ((u32)gsize[g] << 8) | (bin0[g] << 16);
    }
    u8* wsp = d_ws.as<u8>();
    HIPCHK(hipMemsetAsync(wsp, 0, off_ginfo, st));
    HIPCHK(hipMemcpyAsync(wsp + off_ginfo, h_up, up_bytes, hipMemcpyHostToDevice, st));
    c->prof_end();

    KhSkmJob job;
    job.seg_tag = by_group ? wsp + off_tags : nullptr;
    job.segs = reinterpret_cast<const KhSeg*>(wsp + off_segs);
    job.tiles = reinterpret_cast<const KhTile*>(wsp + off_tiles);
    job.reg1 = d_reg1.as<uint4>();
    job.reg2 = d_reg2.as<uint4>();
    job.cur1 = reinterpret_cast<u32*>(wsp + off_cur1);
    job.cur2 = reinterpret_cast<u32*>(wsp + off_cur2);
    job.inst = reinterpret_cast<unsigned long long*>(wsp + off_inst);
    job.dup = reinterpret_cast<unsigned long long*>(wsp + off_dup);
    job.ginfo = reinterpret_cast<const u32*>(wsp + off_ginfo);
    job.hist = reinterpret_cast<unsigned long long*>(wsp);
    job.ctl = reinterpret_cast<u32*>(wsp + off_ctl);
    job.tile_pos = tile_pos;
    job.k = k; job.m = m; job.w = w; job.nmax = nmax;
    job.nslots = nslots; job.S = S; job.nb1 = nb1; job.cap1 = cap1; job.cap2 = cap2;
    job.nbins = nbins; job.abase = abase; job.reps = reps; job.nops = by_group ? (u32)ngroups : (u32)nseq;
    job.table = table;
    job.spill_rec = spill_cap ? d_spill.as<uint4>() : nullptr;
    job.spill_slot = spill_cap ? reinterpret_cast<u32*>(d_spill.as<u8>() + (size_t)spill_cap * rec_bytes) : nullptr;
    job.big_list = spill_cap ? reinterpret_cast<u32*>(d_spill.as<u8>() + (size_t)spill_cap * (rec_bytes + 4)) : nullptr;
    job.spill_cap = spill_cap;
    job.big_cap = big_cap;
#ifdef KH_STAMPS
    Tmp d_stamps;
    const u64 nst = std::max<u64>(ntiles, nslots);
    TMP_ALLOC(d_stamps, c, 128 * nst);
    HIPCHK(hipMemsetAsync(d_stamps.b->p, 0, 128 * nst, st));
    kh_debug_set_stamps_skm(d_stamps.as<u64>());
#endif
    c->prof_begin(KC_SKM_SCATTER);
    if (two) kh_launch_skm2_scatter(job, ntiles, st);
    else kh_launch_skm_scatter(job, ntiles, st);
    c->prof_end();
#ifdef KH_STAMPS
    report_stamps(c, "skm_scatter (second sub-tile: codes / hashes / minima / slots+count / - / append / barrier / -)", d_stamps.b, ntiles);
    HIPCHK(hipMemsetAsync(d_stamps.b->p, 0, 128 * nst, st));
    kh_debug_set_stamps_skm(nullptr);
#endif
    c->prof_begin(KC_SKM_REGROUP);
    if (two) kh_launch_skm2_regroup(job, st);
    else kh_launch_skm_regroup(job, st);
    c->prof_end();
#ifdef KH_STAMPS
    kh_debug_set_stamps_skm(d_stamps.as<u64>());
#endif
    if (rec_out) {   // the records by slot are what the caller wants
        HIPCHK(hipGetLastError());
        u32 h_ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(h_ctl, job.ctl, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (getenv("KHOICE_SKM_DEBUG"))
            fprintf(stderr, "[skm records] k=%d positions=%llu nb1=%u S=%u nslots=%u cap1 %u cap2 %u | records %u errors %u spilled %u\n", k,
                    (unsigned long long)total_pos, nb1, S, nslots, cap1, cap2, h_ctl[2], h_ctl[0], h_ctl[5]);
        // (records on the side list: handed out to a caller that asked for them, a failure otherwise)
        if ((h_ctl[0] & (KH_ERR_CAPACITY | KH_ERR_ORDER)) || (h_ctl[5] && (!rec_out->want_spill || two || h_ctl[5] > spill_cap))) {
            c->stat.retries++;
            return KH_OK;
        }
        if (h_ctl[5]) {
            rec_out->spill = d_spill.b; d_spill.b = nullptr;
            rec_out->spill_n = h_ctl[5];
            rec_out->spill_cap = spill_cap;
        }
        c->stat.skm_records += h_ctl[2];
        if (rec_out->inst) {
            std::vector<u64> hi(nseq);
            HIPCHK(hipMemcpyAsync(hi.data(), job.inst, 8 * (size_t)nseq, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            rec_out->inst->assign(nseq, 0);
            for (int i = 0; i < nseq; ++i) (*rec_out->inst)[perm[i]] = hi[i];
        }
        rec_out->records = h_ctl[2];
        rec_out->reg2 = d_reg2.b; d_reg2.b = nullptr;
        rec_out->ws = d_ws.b; d_ws.b = nullptr;
        rec_out->cur2 = job.cur2;
        rec_out->nslots = nslots;
        rec_out->cap2 = cap2;
        *done = true;
        return KH_OK;
    }
    c->prof_begin(KC_SKM_UNION);
    if (two) kh_launch_skm2_union(job, cs, ugrid, st);
    else kh_launch_skm_union(job, cs, ugrid, st);
    c->prof_end();
    HIPCHK(hipGetLastError());
#ifdef KH_STAMPS
    report_stamps(c, "skm_union (scan barrier / owner / expand / insert / barrier / read-out / barrier / flush)", d_stamps.b, nslots);
    kh_debug_set_stamps_skm(nullptr);
#endif
    if (getenv("KHOICE_SKM_DEBUG")) {   // diagnostics: how full the regions are
        std::vector<u32> h1(nb1), h2(nslots);
        HIPCHK(hipMemcpy2DAsync(h1.data(), 4, job.cur1, 4 * (size_t)KH_SKM_CUR1_STRIDE, 4, nb1, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h2.data(), job.cur2, 4 * (size_t)nslots, hipMemcpyDeviceToHost, st));
        u32 hc[8];
        HIPCHK(hipMemcpyAsync(hc, job.ctl, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        u64 t1 = 0, t2 = 0;
        u32 m1 = 0, m2 = 0;
        for (u32 v : h1) { t1 += v; m1 = std::max(m1, v); }
        for (u32 v : h2) { t2 += v; m2 = std::max(m2, v); }
        fprintf(stderr, "[skm] k=%d m=%d w=%u nmax=%u positions=%llu records=%llu (%.2f k-mers each) nb1=%u S=%u nslots=%u | "
                        "coarse: mean %.0f max %u cap %u | slot: mean %.1f max %u cap %u | tiles %u x %u | expanded: %u k-mers | errors %u spilled %u overfull slots %u\n",
                k, m, w, nmax, (unsigned long long)total_pos, (unsigned long long)t1, (double)total_pos / std::max<u64>(1, t1),
                nb1, S, nslots, (double)t1 / nb1, m1, cap1, (double)t2 / nslots, m2, cap2, ntiles, tile_pos, hc[3], hc[0], hc[5], hc[6]);
    }
    HIPCHK(hipMemcpyAsync(h_down, wsp, down_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const u64* h_hist = reinterpret_cast<const u64*>(h_down);
    const u32* h_ctl = reinterpret_cast<const u32*>(h_down + off_ctl);
    if (h_ctl[6] && !(h_ctl[0] & (KH_ERR_CAPACITY | KH_ERR_ORDER))) {
        // overfull slots (skewed input: a minimizer shared by far more k-mers than a hash predicts): the union left
        // them out; one workgroup each now, and the read-back again — only these slots are done twice, not the call
        if (h_ctl[5] > spill_cap || h_ctl[6] > big_cap) { c->stat.retries++; return KH_OK; }   // too many: the key-array form
        c->stat.big_slots += h_ctl[6];
        c->prof_begin(KC_SKM_BIG);
        if (two) kh_launch_skm2_big(job, cs, h_ctl[6], st);
        else kh_launch_skm_big(job, cs, h_ctl[6], st);
        c->prof_end();
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h_down, wsp, down_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    const u64* h_inst = reinterpret_cast<const u64*>(h_down + off_inst);
    const u64* h_dup = reinterpret_cast<const u64*>(h_down + off_dup);
    if (h_ctl[0] & (KH_ERR_CAPACITY | KH_ERR_ORDER)) {
        c->stat.retries++;
        return KH_OK;   // a region or a slot overflowed: the key-array form takes over
    }
    c->stat.skm_records += h_ctl[2];
    if (!by_group) {
        c->stat.bases += bases;
        c->stat.builds += nseq;
        u64 inst = 0, dsum = 0;
        for (int i = 0; i < nseq; ++i) {
            const u64 d = h_inst[i] - h_dup[i];
            inst += h_inst[i];
            dsum += d;
            if (distinct_per_seq) distinct_per_seq[perm[i]] = d;
        }
        c->stat.kmers += inst;
        c->stat.distinct += dsum;
        c->stat.setop_in += dsum;
    }
    c->stat.setops++;
    std::vector<u64> bins(nbins, 0);
    for (u32 r = 0; r < reps; ++r)
        for (u32 b = 0; b < nbins; ++b) bins[b] += h_hist[(size_t)r * nbins + b];
    if (within_hist && !by_group) {
        memset(within_hist, 0, 8 * (size_t)ngroups * hist_len);
        for (int g = 0; g < ngroups; ++g)
            for (int cnt = 1; cnt <= gsize[g]; ++cnt)
                within_hist[(size_t)g * hist_len + std::min<u32>((u32)cnt, hist_len - 1)] += bins[bin0[g] + cnt];
    }
    u64 across_n = 0;
    for (int cnt = 1; cnt <= ngroups; ++cnt) across_n += bins[abase + cnt];
    c->stat.setop_out += across_n;
    if (across_hist) {
        memset(across_hist, 0, 8 * (size_t)hist_len);
        for (int cnt = 1; cnt <= ngroups; ++cnt)
            across_hist[std::min<u32>((u32)cnt, hist_len - 1)] += bins[abase + cnt];
    }
    *done = true;
    return KH_OK;
}

// The fused form of steps 1-8 (no per-genome / per-group database is handed out): ONE batched build
// in grid mode, ONE tagged union over all genomes, ONE host synchronisation.  *done == false on
// return means "not applicable or a slot overflowed": the caller takes the general path.
static int exp1_fused(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens, int on_device,
                      const int* group_of, int ngroups, int k, u32 cs, uint64_t* within_hist,
                      uint64_t* across_hist, u32 hist_len, uint64_t* distinct_per_seq, kh_set** across_set,
                      bool* done, u32 nwaves = 1, double s_scale = 1.0, double* want_scale = nullptr, u32 fan_hint = 0) {
    *done = false;
    if (want_scale) *want_scale = 0.0;
    const int W = k <= 32 ? 1 : 2;
    if (nseq > KH_TAG_MAX_OPS || ngroups > KH_TAG_MAX_OPS) return KH_OK;
    // operands in group-major order: the genomes of a group are consecutive bits of the mask
    std::vector<int> gsize(ngroups, 0), gstart(ngroups + 1, 0), perm(nseq);
    for (int i = 0; i < nseq; ++i) gsize[group_of[i]]++;
    u32 fan = 1, nbins = 0;
    std::vector<u32> bin0(ngroups);
    for (int g = 0; g < ngroups; ++g) {
        if (!gsize[g]) return kh_fail(KH_E_ARG, "group %d has no sequences", g);
        gstart[g + 1] = gstart[g] + gsize[g];
        bin0[g] = nbins;
        nbins += (u32)gsize[g] + 1;
        fan = std::max<u32>(fan, (u32)gsize[g]);
    }
    const u32 abase = nbins;
    nbins += (u32)ngroups + 1;
    if (nbins > (u32)KH_TAG_MAX_BINS) return KH_OK;
    {
        std::vector<int> at(gstart.begin(), gstart.end() - 1);
        for (int i = 0; i < nseq; ++i) perm[at[group_of[i]]++] = i;
    }
    const u32 mean = k <= 32 ? KH_BUCKET_MEAN_W1 : KH_BUCKET_MEAN_W2;
    std::vector<const uint8_t*> pseq(nseq);
    std::vector<uint64_t> plen(nseq);
    for (int i = 0; i < nseq; ++i) {
        pseq[i] = seqs[perm[i]];
        plen[i] = lens[perm[i]];
        // longer than one segment's bucket table: the general path cuts such sequences into chunks
        if (lens[perm[i]] >= (u64)k && lens[perm[i]] - k + 1 > (u64)(KH_MAX_BUCKETS_PER_SEG / 4) * mean) return KH_OK;
    }
    HIPCHK(hipSetDevice(c->dev));
    hipStream_t st = c->st;

    std::vector<u64> bins(nbins, 0), dist_acc(nseq, 0);
    const bool emit = across_set != nullptr;
    std::vector<kh_set*> wave_sets;   // emitted across-group sets, one per wave (disjoint, ascending key ranges)
    auto drop_sets = [&]() { for (auto* s : wave_sets) kh_set_free(s); wave_sets.clear(); };
    struct SetsGuard { std::vector<kh_set*>& v; ~SetsGuard() { for (auto* s : v) kh_set_free(s); } } sets_guard{wave_sets};
    for (u32 wave = 0; wave < nwaves; ++wave) {
    GridBuild gb;
    gb.fan = std::max(fan, fan_hint);   // (fan_hint: genomes that are related although every one is a group of its own here)
    gb.wave = wave;
    gb.nwaves = nwaves;
    gb.s_scale = s_scale;
    // one-word keys, nothing emitted: the hash-set kernel; its slot size is a tuning choice
    const bool hash_form = W == 1 && !emit && !getenv("KHOICE_NO_UNION_HASH");
    gb.cap = hash_form ? kh_union_hash_capacity() : (W == 1 ? KH_SORT_CAP_PAY_W1 : KH_SORT_CAP_PAY_W2);
    bool cap_hit = false, again = false;
    {
        const int br = build_once(c, nseq, pseq.data(), plen.data(), on_device, k, 1, KH_NO_MAX, KH_KMC_DEFAULT_CS, 0, mean,
                                  nullptr, &cap_hit, &again, &gb);
        if (br != KH_OK) { drop_sets(); return br; }
    }

    // ---- tagged union queued behind the build
    const u32 nslots = gb.nb * gb.S;
    const u32 grid = nslots;
    const u32 reps = std::min<u32>(256, std::max<u32>(1, grid));
    // workspace: [hist: reps x nbins u64][ctl: 8 u32][out_n u64][ginfo: 64 u32][descriptors: nslots u64 when emitting]
    const size_t hist_words = (size_t)reps * nbins;
    const size_t ws_bytes = 8 * hist_words + 32 + 8 + 256 + (emit ? 8 * (size_t)nslots : 0);
    Tmp d_ws;
    TMP_ALLOC(d_ws, c, ws_bytes);
    u8* wsp = d_ws.as<u8>();
    struct PinG { kh_ctx* c; void* p = nullptr; size_t n = 0; ~PinG() { if (p) c->pin_release(p, n); } } pin{c};
    // pinned staging: [ginfo upload: 64 u32][hist read-back][ctl + out_n: 40 B][distinct: nseq u64][pass C tail: 64 B][nvalid u64]
    const size_t pin_need = 256 + 8 * hist_words + 40 + 8 * (size_t)nseq + 64 + 8;
    pin.p = c->pin_alloc(pin_need, &pin.n);
    if (!pin.p) return kh_fail(KH_E_NOMEM, "pinned host allocation failed");
    u32* h_ginfo = static_cast<u32*>(pin.p);
    u64* h_hist = reinterpret_cast<u64*>(h_ginfo + 64);
    u32* h_ctl = reinterpret_cast<u32*>(h_hist + hist_words);
    u64* h_distinct = reinterpret_cast<u64*>(h_ctl + 10);
    u64* h_ctail = h_distinct + nseq;
    u64* h_nvalid = h_ctail + 8;
    memset(h_ginfo, 0, 256);
    for (int g = 0; g < ngroups; ++g)
        for (int j = 0; j < gsize[g]; ++j)
            h_ginfo[gstart[g] + j] = (u32)gstart[g] | ((u32)gsize[g] << 8) | (bin0[g] << 16);
    HIPCHK(hipMemsetAsync(wsp, 0, 8 * hist_words + 40, st));
    HIPCHK(hipMemcpyAsync(wsp + 8 * hist_words + 40, h_ginfo, 256, hipMemcpyHostToDevice, st));
    if (emit) HIPCHK(hipMemsetAsync(wsp + 8 * hist_words + 40 + 256, 0, 8 * (size_t)nslots, st));
    DevBuf *okeys = nullptr, *ocnt = nullptr;
    struct OutGuard { DevBuf*& a; DevBuf*& b; ~OutGuard() { buf_unref(a); buf_unref(b); } } og{okeys, ocnt};
    if (emit) {
        okeys = c->buf_alloc(8 * (size_t)W * std::max<u64>(1, gb.key_cap));
        ocnt = c->buf_alloc(4 * std::max<u64>(1, gb.key_cap));
        if (!okeys || !ocnt) return kh_fail(KH_E_NOMEM, "device allocation failed (across-group set)");
    }
    KhTagJob job;
    job.keys = gb.okeys->p;
    job.bstart = reinterpret_cast<const u64*>(gb.bstart->p);
    job.off = reinterpret_cast<const u16*>(gb.off->p);
    job.ginfo = reinterpret_cast<const u32*>(wsp + 8 * hist_words + 40);
    job.hist = reinterpret_cast<unsigned long long*>(wsp);
    job.ctl = reinterpret_cast<u32*>(wsp + 8 * hist_words);
    job.nb = gb.nb; job.S = gb.S; job.nops = (u32)nseq; job.nbins = nbins; job.abase = abase;
    job.ngroups = (u32)ngroups; job.reps = reps;
    job.nbv = gb.nb * nwaves;
    job.binmul = (1u << 26) / ((KH_FINE_BINS + gb.S - 1) / gb.S);
    job.out_keys = emit ? okeys->p : nullptr;
    job.out_counts = emit ? reinterpret_cast<u32*>(ocnt->p) : nullptr;
    job.desc = emit ? reinterpret_cast<u64*>(wsp + 8 * hist_words + 40 + 256) : nullptr;
    job.out_n = reinterpret_cast<unsigned long long*>(wsp + 8 * hist_words + 32);
#ifdef KH_STAMPS
    Tmp d_stamps;
    TMP_ALLOC(d_stamps, c, 128 * (u64)grid);
    HIPCHK(hipMemsetAsync(d_stamps.b->p, 0, 128 * (u64)grid, st));
    kh_debug_set_stamps(d_stamps.as<u64>());
#endif
    c->prof_begin(KC_UNION_TAGGED);
    if (hash_form) kh_launch_union_hash(job, grid, k, cs, st);
    else kh_launch_union_tagged(W, job, grid, k, cs, st);
    c->prof_end();
    HIPCHK(hipGetLastError());
#ifdef KH_STAMPS
    report_stamps(c, "union_tagged", d_stamps.b, grid);
    kh_debug_set_stamps(nullptr);
#endif
    // ---- one read-back, one wait
    HIPCHK(hipMemcpyAsync(h_hist, wsp, 8 * hist_words + 40, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_distinct, gb.distinct->p, 8 * (size_t)nseq, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_ctail, reinterpret_cast<u64*>(gb.lb->p) + gb.nb_total, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_nvalid, reinterpret_cast<u64*>(gb.bstart->p) + gb.nb_total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const u32 cerr = reinterpret_cast<const u32*>(h_ctail)[1];   // pass C: [ticket, err]
    const u32 uerr = h_ctl[0];
    if ((cerr | uerr) & (KH_ERR_CAPACITY | KH_ERR_SPIN_TIMEOUT | KH_ERR_ORDER)) {
        c->stat.retries++;
        drop_sets();
        // a slot of the union held more records than fit (keys shared by many genomes, repeats): the
        // kernel recorded its fullest slot, the caller may try again with that many more sub-ranges
        if (want_scale && !(cerr & KH_ERR_CAPACITY) && (uerr & KH_ERR_CAPACITY) && !(uerr & KH_ERR_ORDER) && h_ctl[1] > gb.cap)
            *want_scale = s_scale * 1.15 * (double)h_ctl[1] / (double)gb.cap;
        return KH_OK;   // otherwise the general path re-plans
    }
    if (wave == 0) { c->stat.bases += gb.bases; c->stat.builds += nseq; }
    c->stat.kmers += *h_nvalid;
    c->stat.setops++;
    u64 dsum = 0;
    for (int i = 0; i < nseq; ++i) {
        dist_acc[i] += h_distinct[i];
        dsum += h_distinct[i];
    }
    c->stat.distinct += dsum;
    c->stat.setop_in += dsum;
    for (u32 r = 0; r < reps; ++r)
        for (u32 b = 0; b < nbins; ++b) bins[b] += h_hist[(size_t)r * nbins + b];
    if (emit) {
        const u64 n = *reinterpret_cast<const u64*>(h_ctl + 8);
        buf_ref(okeys);
        buf_ref(ocnt);
        wave_sets.push_back(make_set(k, n, okeys, 0, ocnt, 0, 1, cs));
    }
    }   // waves
    if (distinct_per_seq)
        for (int i = 0; i < nseq; ++i) distinct_per_seq[perm[i]] = dist_acc[i];
    if (within_hist) {
        memset(within_hist, 0, 8 * (size_t)ngroups * hist_len);
        for (int g = 0; g < ngroups; ++g)
            for (int cnt = 1; cnt <= gsize[g]; ++cnt)
                within_hist[(size_t)g * hist_len + std::min<u32>((u32)cnt, hist_len - 1)] += bins[bin0[g] + cnt];
    }
    u64 across_n = 0;
    for (int cnt = 1; cnt <= ngroups; ++cnt) across_n += bins[abase + cnt];
    c->stat.setop_out += across_n;
    if (across_hist) {
        memset(across_hist, 0, 8 * (size_t)hist_len);
        for (int cnt = 1; cnt <= ngroups; ++cnt)
            across_hist[std::min<u32>((u32)cnt, hist_len - 1)] += bins[abase + cnt];
    }
    if (across_set) {
        if (wave_sets.size() == 1) {
            *across_set = wave_sets[0];
            wave_sets.clear();
        } else {   // the waves' sets cover disjoint key ranges: their union is their concatenation
            const int r = kh_union_sum(c, wave_sets.data(), (int)wave_sets.size(), cs, across_set, nullptr, 0);
            drop_sets();
            if (r != KH_OK) return r;
        }
    }
    *done = true;
    return KH_OK;
}

static int exp1_big_group_skm(kh_ctx* c, int n, const uint8_t* const* seqs, const uint64_t* lens, int on_device, int k, u32 cs,
                              uint64_t* whist, u32 hist_len, uint64_t* dist, bool* done);
extern "C" int kh_exp1_run(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens,
                           int on_device, const int* group_of, int ngroups, int k, uint32_t cs,
                           uint64_t* within_hist, uint64_t* across_hist, uint32_t hist_len,
                           uint64_t* distinct_per_seq, kh_set** group_sets, kh_set** across_set) {
    if (!c || !seqs || !lens || !group_of || nseq <= 0 || ngroups <= 0)
        return kh_fail(KH_E_ARG, "kh_exp1_run: bad argument");
    if ((within_hist || across_hist) && hist_len < 2) return kh_fail(KH_E_ARG, "hist_len must be >= 2");
    for (int i = 0; i < nseq; ++i)
        if (group_of[i] < 0 || group_of[i] >= ngroups)
            return kh_fail(KH_E_ARG, "group_of[%d]=%d outside [0,%d)", i, group_of[i], ngroups);
    KHCHK(check_k(k));
    if (cs < 1) return kh_fail(KH_E_ARG, "cs must be >= 1");
    // No per-group database wanted: the fused form (one build in grid mode + one tagged union per
    // BATCH of whole groups; a batch holds at most 64 genomes — the width of the genome mask — and
    // fits the memory budget).  One batch: its histograms are the answer.  Several: each batch also
    // emits its across-group set (counter = groups of the batch holding the k-mer) and one
    // counter-summing union of those sets gives step 7/8.  Anything the fused form cannot take
    // (a group of more than 64 genomes, a slot overflow) falls through to the general path below.
    // A fast form that is abandoned part-way (a later batch or wave overflows) has already counted its earlier
    // batches: the statistics of an attempt are kept only when the attempt succeeds.
    const Stats stat_at_entry = c->stat;
    auto forget_attempt = [&]() {
        const u64 retries = c->stat.retries, order_fallbacks = c->stat.order_fallbacks;
        c->stat = stat_at_entry;
        c->stat.retries = retries;
        c->stat.order_fallbacks = order_fallbacks;
    };
    if (!group_sets && !getenv("KHOICE_NO_FUSED")) {
        u64 fbudget = 4ull << 30;
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipSetDevice(c->dev));
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            fbudget = std::max<u64>(1u << 20, (free_b + c->pool.cached_bytes) / (k <= 32 ? 64 : 96));
        if (const char* e = getenv("KHOICE_WAVE_BASES")) fbudget = std::max<u64>(1, strtoull(e, nullptr, 10));
        std::vector<u64> gbases(ngroups, 0);
        std::vector<int> gcount(ngroups, 0);
        for (int i = 0; i < nseq; ++i) { gbases[group_of[i]] += lens[i]; gcount[group_of[i]]++; }
        std::vector<int> batch_end;      // group index one past each batch
        std::vector<char> batch_big;     // the batch is ONE group of more than 64 genomes (exp1_big_group)
        bool applicable = true;
        {
            u64 acc_b = 0;
            int acc_n = 0, acc_g = 0, acc_bins = 0;
            for (int g = 0; g < ngroups && applicable; ++g) {
                if (gcount[g] == 0) { applicable = false; break; }
                if (gcount[g] > KH_TAG_MAX_OPS) {   // wider than the genome mask: a batch of its own, taken in sub-batches
                    if (acc_n) { batch_end.push_back(g); batch_big.push_back(0); acc_b = 0; acc_n = acc_g = acc_bins = 0; }
                    batch_end.push_back(g + 1);
                    batch_big.push_back(1);
                    continue;
                }
                // (a batch above the memory budget is run as key-range waves, see below: groups are
                // only split into batches by the width of the genome mask)
                const bool fits = acc_n + gcount[g] <= KH_TAG_MAX_OPS && acc_g + 1 <= KH_TAG_MAX_OPS &&
                                  acc_bins + gcount[g] + 1 + (acc_g + 2) <= KH_TAG_MAX_BINS &&
                                  (acc_b + gbases[g] <= fbudget || acc_n == 0);
                if (!fits) { batch_end.push_back(g); batch_big.push_back(0); acc_b = 0; acc_n = acc_g = acc_bins = 0; }
                acc_b += gbases[g]; acc_n += gcount[g]; acc_g += 1; acc_bins += gcount[g] + 1;
            }
            if (acc_n || batch_end.empty()) { batch_end.push_back(ngroups); batch_big.push_back(0); }
        }
        // a batch whose bases exceed the budget: key-range waves (HBM-spill partitioning of BASELINE
        // configs[4]) — every wave re-extracts the batch's bases but keeps one slice of the key space,
        // so the memory in flight is 1/waves of the keys and the histograms of the waves add up
        auto waves_for = [&](u64 bases) -> u32 { return (u32)std::max<u64>(1, (bases + fbudget - 1) / fbudget); };
        // One group of more than 64 genomes (exp_type_1.smk:36-61 lists whatever data/dataset_N holds): sub-batches of
        // up to 64 genomes, each a fused build + tagged union in which every genome is a group of its own, so that the
        // set a sub-batch emits carries "in how many of its genomes"; one counter-summing union of those sets is the
        // group's step_3 database (its histogram fused: step_4).
        auto big_group = [&](int g, uint64_t* whist, uint64_t* dist_out /* [nseq] or null */, kh_set** group_set, bool* done) -> int {
            *done = false;
            std::vector<int> members;
            for (int i = 0; i < nseq; ++i)
                if (group_of[i] == g) members.push_back(i);
            if (!group_set && whist) {   // histogram and distinct counts only: the super-k-mer form in phases
                std::vector<const uint8_t*> ms(members.size());
                std::vector<uint64_t> ml(members.size()), md(members.size(), 0);
                for (size_t j = 0; j < members.size(); ++j) { ms[j] = seqs[members[j]]; ml[j] = lens[members[j]]; }
                const Stats before = c->stat;
                KHCHK(exp1_big_group_skm(c, (int)members.size(), ms.data(), ml.data(), on_device, k, cs, whist, hist_len, md.data(), done));
                if (*done) {
                    if (dist_out)
                        for (size_t j = 0; j < members.size(); ++j) dist_out[members[j]] = md[j];
                    return KH_OK;
                }
                const u64 tries = c->stat.retries;   // (work counted by an attempt that was given up is forgotten, the attempt is not)
                c->stat = before;
                c->stat.retries = tries;
            }
            std::vector<kh_set*> subs;
            struct G { std::vector<kh_set*>& v; ~G() { for (auto* x : v) kh_set_free(x); } } guard{subs};
            for (size_t i0 = 0; i0 < members.size(); i0 += KH_TAG_MAX_OPS) {
                const size_t m = std::min<size_t>(KH_TAG_MAX_OPS, members.size() - i0);
                std::vector<const uint8_t*> bs(m);
                std::vector<uint64_t> bl(m), bd(m, 0);
                std::vector<int> bg(m);
                u64 sb = 0;
                for (size_t j = 0; j < m; ++j) { bs[j] = seqs[members[i0 + j]]; bl[j] = lens[members[i0 + j]]; bg[j] = (int)j; sb += bl[j]; }
                kh_set* aset = nullptr;
                bool d = false;
                double scale = 0.0;
                KHCHK(exp1_fused(c, (int)m, bs.data(), bl.data(), on_device, bg.data(), (int)m, k, 0x7fffffffu, nullptr, nullptr,
                                 hist_len, bd.data(), &aset, &d, waves_for(sb), 1.0, &scale, (u32)m));
                if (!d && scale > 1.0 && scale < 16.0)   // one more try with finer slots
                    KHCHK(exp1_fused(c, (int)m, bs.data(), bl.data(), on_device, bg.data(), (int)m, k, 0x7fffffffu, nullptr, nullptr,
                                     hist_len, bd.data(), &aset, &d, waves_for(sb), scale, nullptr, (u32)m));
                if (!d) return KH_OK;
                subs.push_back(aset);
                if (dist_out)
                    for (size_t j = 0; j < m; ++j) dist_out[members[i0 + j]] = bd[j];
            }
            if (group_set) KHCHK(kh_union_sum(c, subs.data(), (int)subs.size(), cs, group_set, whist, whist ? hist_len : 0));
            else if (whist) KHCHK(kh_union_histogram(c, subs.data(), (int)subs.size(), cs, whist, hist_len));
            *done = true;
            return KH_OK;
        };
        if (applicable && batch_end.size() == 1 && !batch_big[0]) {
            u64 all_bases = 0;
            for (int g = 0; g < ngroups; ++g) all_bases += gbases[g];
            bool done = false;
            double scale = 0.0;
            if (!across_set) {   // histograms and distinct counts only: the super-k-mer form
                KHCHK(exp1_skm(c, nseq, seqs, lens, on_device, group_of, ngroups, k, cs, within_hist, across_hist,
                               hist_len, distinct_per_seq, &done));
                if (done) return KH_OK;
                forget_attempt();
            }
            KHCHK(exp1_fused(c, nseq, seqs, lens, on_device, group_of, ngroups, k, cs, within_hist, across_hist,
                             hist_len, distinct_per_seq, across_set, &done, waves_for(all_bases), 1.0, &scale));
            if (done) return KH_OK;
            forget_attempt();
            if (scale > 1.0 && scale < 16.0) {   // one more try with finer slots before the general path
                KHCHK(exp1_fused(c, nseq, seqs, lens, on_device, group_of, ngroups, k, cs, within_hist, across_hist,
                                 hist_len, distinct_per_seq, across_set, &done, waves_for(all_bases), scale, nullptr));
                if (done) return KH_OK;
                forget_attempt();
            }
        } else if (applicable) {
            const bool want_across = across_hist || across_set;
            // Histograms only, more than 64 genomes: every batch in the super-k-mer form for the within-group
            // questions, then ONE more pass over all genomes whose records carry the group number for the
            // across-group one.  Anything it cannot take (a region overflow, k outside its range) -> the key arrays.
            if (!across_set && across_hist && ngroups <= KH_TAG_MAX_OPS && !getenv("KHOICE_NO_SKM_TWO_PASS")) {
                bool ok2 = true;
                int g0b = 0;
                std::vector<uint64_t> wtmp(within_hist ? (size_t)ngroups * hist_len : 0), dtmp(nseq, 0), atmp(hist_len, 0);
                for (size_t b = 0; b < batch_end.size() && ok2; ++b) {
                    const int g1b = batch_end[b];
                    if (batch_big[b]) {
                        bool done = false;
                        KHCHK(big_group(g0b, within_hist ? wtmp.data() + (size_t)g0b * hist_len : nullptr, dtmp.data(), nullptr, &done));
                        if (!done) { ok2 = false; break; }
                        g0b = g1b;
                        continue;
                    }
                    std::vector<int> idx;
                    for (int i = 0; i < nseq; ++i)
                        if (group_of[i] >= g0b && group_of[i] < g1b) idx.push_back(i);
                    std::vector<const uint8_t*> bs(idx.size());
                    std::vector<uint64_t> bl(idx.size()), bd(idx.size(), 0);
                    std::vector<int> bg(idx.size());
                    for (size_t j = 0; j < idx.size(); ++j) { bs[j] = seqs[idx[j]]; bl[j] = lens[idx[j]]; bg[j] = group_of[idx[j]] - g0b; }
                    bool done = false;
                    KHCHK(exp1_skm(c, (int)idx.size(), bs.data(), bl.data(), on_device, bg.data(), g1b - g0b, k, cs,
                                   within_hist ? wtmp.data() + (size_t)g0b * hist_len : nullptr, nullptr, hist_len, bd.data(), &done));
                    if (!done) { ok2 = false; break; }
                    for (size_t j = 0; j < idx.size(); ++j) dtmp[idx[j]] = bd[j];
                    g0b = g1b;
                }
                if (ok2) {
                    bool done = false;
                    KHCHK(exp1_skm(c, nseq, seqs, lens, on_device, group_of, ngroups, k, cs, nullptr, atmp.data(), hist_len, nullptr,
                                   &done, /*by_group=*/true));
                    ok2 = done;
                }
                if (ok2) {
                    if (within_hist) memcpy(within_hist, wtmp.data(), 8 * wtmp.size());
                    memcpy(across_hist, atmp.data(), 8 * (size_t)hist_len);
                    if (distinct_per_seq) memcpy(distinct_per_seq, dtmp.data(), 8 * (size_t)nseq);
                    return KH_OK;
                }
                forget_attempt();
            }
            std::vector<kh_set*> asets;
            auto drop = [&]() { for (auto* s : asets) kh_set_free(s); asets.clear(); };
            bool ok = true;
            int g0 = 0;
            for (size_t b = 0; b < batch_end.size() && ok; ++b) {
                const int g1 = batch_end[b];
                if (batch_big[b]) {   // its across-group set: the group's k-mers, each counted once
                    kh_set* gs = nullptr;
                    bool done = false;
                    int r = big_group(g0, within_hist ? within_hist + (size_t)g0 * hist_len : nullptr, distinct_per_seq,
                                      want_across ? &gs : nullptr, &done);
                    if (r == KH_OK && done && gs) {
                        kh_set* one = nullptr;
                        r = kh_set_counts(c, gs, 1, &one);
                        kh_set_free(gs);
                        if (r == KH_OK) asets.push_back(one);
                    }
                    if (r != KH_OK) { drop(); return r; }
                    if (!done) { ok = false; break; }
                    g0 = g1;
                    continue;
                }
                std::vector<int> idx;
                for (int i = 0; i < nseq; ++i)
                    if (group_of[i] >= g0 && group_of[i] < g1) idx.push_back(i);
                std::vector<const uint8_t*> bs(idx.size());
                std::vector<uint64_t> bl(idx.size()), bd(idx.size(), 0);
                std::vector<int> bg(idx.size());
                for (size_t j = 0; j < idx.size(); ++j) { bs[j] = seqs[idx[j]]; bl[j] = lens[idx[j]]; bg[j] = group_of[idx[j]] - g0; }
                kh_set* aset = nullptr;
                bool done = false;
                u64 bbases = 0;
                for (int g = g0; g < g1; ++g) bbases += gbases[g];
                int r = KH_OK;
                if (!want_across)   // steps 1-4 only (more than 64 genomes, no across-group step): the batches are independent
                    r = exp1_skm(c, (int)idx.size(), bs.data(), bl.data(), on_device, bg.data(), g1 - g0, k, cs,
                                 within_hist ? within_hist + (size_t)g0 * hist_len : nullptr, nullptr, hist_len, bd.data(), &done);
                if (r == KH_OK && !done)
                    r = exp1_fused(c, (int)idx.size(), bs.data(), bl.data(), on_device, bg.data(), g1 - g0, k, cs,
                                   within_hist ? within_hist + (size_t)g0 * hist_len : nullptr, nullptr, hist_len,
                                   bd.data(), want_across ? &aset : nullptr, &done, waves_for(bbases));
                if (r != KH_OK) { drop(); return r; }
                if (!done) { ok = false; break; }
                if (aset) asets.push_back(aset);
                if (distinct_per_seq)
                    for (size_t j = 0; j < idx.size(); ++j) distinct_per_seq[idx[j]] = bd[j];
                g0 = g1;
            }
            if (ok) {
                int r = KH_OK;
                if (across_set) r = kh_union_sum(c, asets.data(), (int)asets.size(), cs, across_set, across_hist, hist_len);
                else if (across_hist) r = kh_union_histogram(c, asets.data(), (int)asets.size(), cs, across_hist, hist_len);
                drop();
                return r;
            }
            drop();
            forget_attempt();
        }
    }
    std::vector<kh_set*> gsets(nseq, nullptr), unions(ngroups, nullptr), usets(ngroups, nullptr);
    kh_set* across = nullptr;
    auto cleanup = [&]() {
        for (auto* s : gsets) kh_set_free(s);
        for (auto* s : unions) kh_set_free(s);
        for (auto* s : usets) kh_set_free(s);
        kh_set_free(across);
    };
    // Groups are independent until step 7, so they are processed in waves that fit a device
    // memory budget (all groups at once for the benchmark sizes; wave by wave for inputs whose
    // k-mers would not fit HBM together): per wave, steps 1+2 build every genome of the wave as
    // a plain set in ONE batched launch sequence, steps 3+4+6 enqueue the wave's group unions
    // back to back before the host waits once; the genome sets are released before the next wave.
    // A base in flight costs ~8W bytes in the partition array + 8W (+4) in the output arrays +
    // its share of the group union: budget = 1/64 (W=1) or 1/96 (W=2) of the free HBM, in bases.
    u64 budget = 4ull << 30;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            budget = std::max<u64>(1u << 20, (free_b + c->pool.cached_bytes) / (k <= 32 ? 64 : 96));
    }
    if (const char* e = getenv("KHOICE_WAVE_BASES")) budget = std::max<u64>(1, strtoull(e, nullptr, 10));
    std::vector<u64> group_bases(ngroups, 0);
    for (int i = 0; i < nseq; ++i) group_bases[group_of[i]] += lens[i];
    int r = KH_OK;
    const double t_begin = g_trace ? now_ms() : 0;
    double t_unions_submitted = 0, t_unions_synced = 0, t_groups_done = 0;
    for (int g0 = 0; g0 < ngroups;) {
        if (group_bases[g0] > budget) {             // one group larger than a wave: sub-waves of genomes
            std::vector<int> members;
            for (int i = 0; i < nseq; ++i)
                if (group_of[i] == g0) members.push_back(i);
            r = group_union_incremental(c, members, seqs, lens, on_device, k, cs, budget,
                                        within_hist ? within_hist + (size_t)g0 * hist_len : nullptr, hist_len,
                                        distinct_per_seq, &unions[g0]);
            if (r == KH_OK) r = kh_set_counts(c, unions[g0], 1, &usets[g0]);
            if (r != KH_OK) { cleanup(); return r; }
            ++g0;
            continue;
        }
        int g1 = g0;
        u64 acc = 0;
        while (g1 < ngroups && acc + group_bases[g1] <= budget) acc += group_bases[g1++];
        std::vector<int> idx;                       // sequences of groups [g0, g1)
        for (int i = 0; i < nseq; ++i)
            if (group_of[i] >= g0 && group_of[i] < g1) idx.push_back(i);
        std::vector<const uint8_t*> wseqs(idx.size());
        std::vector<uint64_t> wlens(idx.size());
        std::vector<kh_set*> wsets(idx.size(), nullptr);
        for (size_t j = 0; j < idx.size(); ++j) { wseqs[j] = seqs[idx[j]]; wlens[j] = lens[idx[j]]; }
        r = kh_build_batch(c, (int)idx.size(), wseqs.data(), wlens.data(), on_device, k, 1, KH_NO_MAX,
                           KH_KMC_DEFAULT_CS, 0, wsets.data());
        if (r != KH_OK) { cleanup(); return r; }
        for (size_t j = 0; j < idx.size(); ++j) {
            gsets[idx[j]] = wsets[j];
            if (distinct_per_seq) distinct_per_seq[idx[j]] = wsets[j]->n;
        }
        std::vector<SetopJob> jobs(g1 - g0);
        for (int g = g0; g < g1; ++g) {
            SetopJob& j = jobs[g - g0];
            j.c = c; j.op = KH_OP_UNION; j.mode = KH_OC_SUM; j.cs = cs;
            j.hist = within_hist ? within_hist + (size_t)g * hist_len : nullptr;
            j.hist_len = hist_len;
            for (int i : idx)
                if (group_of[i] == g) j.in.push_back(gsets[i]);
            if (j.in.empty()) { cleanup(); return kh_fail(KH_E_ARG, "group %d has no sequences", g); }
            if ((int)j.in.size() > KH_MAX_INPUT_SETS) {   // beyond one launch's fan-in: the general path
                r = kh_union_sum(c, j.in.data(), (int)j.in.size(), cs, &unions[g], j.hist, hist_len);
                if (r != KH_OK) { cleanup(); return r; }
                j.in.clear();
                continue;
            }
            r = setop_prepare(j);
            if (r == KH_OK) r = setop_plan(j);
            if (r != KH_OK) { cleanup(); return r; }
        }
        // the slot bounds of all unions of the wave in one launch, then the unions back to back
        Tmp d_bjobs;
        struct PinGuard { kh_ctx* c; void* p = nullptr; size_t n = 0; ~PinGuard() { if (p) c->pin_release(p, n); } } bpin{c};
        r = setop_bounds_batch(c, jobs, d_bjobs, &bpin.p, &bpin.n);
        if (r == KH_OK) {
            std::vector<SetopJob*> run;
            for (auto& j : jobs) run.push_back(&j);
            r = setop_run_batch(c, run.data(), run.size());
        }
        if (r != KH_OK) { cleanup(); return r; }
        if (g_trace) t_unions_submitted = now_ms();
        if (hipStreamSynchronize(c->st) != hipSuccess) { cleanup(); return kh_fail(KH_E_HIP, "stream sync failed"); }
        if (g_trace) t_unions_synced = now_ms();
        for (int g = g0; g < g1; ++g) {
            if (jobs[g - g0].in.empty()) continue;
            r = setop_finish(jobs[g - g0], &unions[g]);
            if (r != KH_OK) { cleanup(); return r; }
        }
        jobs.clear();
        for (int g = g0; g < g1; ++g) {
            r = kh_set_counts(c, unions[g], 1, &usets[g]);
            if (r != KH_OK) { cleanup(); return r; }
        }
        for (int i : idx) { kh_set_free(gsets[i]); gsets[i] = nullptr; }   // genome sets of this wave
        g0 = g1;
    }
    // steps 7+8 (skipped when the caller wants neither output: the multi-GPU path does them
    // after exchanging the group sets, khoice_amd/dist.py)
    if (g_trace) t_groups_done = now_ms();
    if (across_set) {
        r = kh_union_sum(c, usets.data(), ngroups, cs, &across, across_hist, hist_len);
        if (r != KH_OK) { cleanup(); return r; }
    } else if (across_hist) {
        r = kh_union_histogram(c, usets.data(), ngroups, cs, across_hist, hist_len);
        if (r != KH_OK) { cleanup(); return r; }
    }
    if (g_trace)
        fprintf(stderr, "[khoice trace] build submit %.3f wait %.3f | unions submit %.3f wait %.3f finish %.3f | "
                        "across %.3f | total %.3f ms\n",
                g_t_build_submitted - t_begin, g_t_build_synced - g_t_build_submitted,
                t_unions_submitted - g_t_build_synced, t_unions_synced - t_unions_submitted,
                t_groups_done - t_unions_synced, now_ms() - t_groups_done, now_ms() - t_begin);
    if (group_sets)
        for (int g = 0; g < ngroups; ++g) { group_sets[g] = unions[g]; unions[g] = nullptr; }
    if (across_set) { *across_set = across; across = nullptr; }
    cleanup();
    return KH_OK;
}

// ------------------------------------------------------------------------------ exchange form of steps 7-8
// Multi-GPU (khoice_amd/dist.py): every rank turns its genomes into records tagged with the LOCAL group number,
// merges identical ones and packs them by the rank that owns their slot (kh_skm_pack); the packed arrays travel
// (all-to-all); the owner runs ONE phased union over the pieces it received (kh_skm_phased_histogram) and the small
// histograms are all-reduced.  Slot geometry must be the same on all ranks: kh_skm_exchange_plan works it out from
// numbers the ranks have agreed on (the largest rank's k-mer positions, the largest group).
static bool skm_exchange_k(int k) { return k >= KH_SKM_MIN_K && k <= KH_SKM_MAX_K; }

// slots for `nparts` pieces of at most positions_max k-mer positions each; false: too many k-mers per piece
static bool skm_exchange_geometry(int k, uint64_t positions_max, int nparts, u32 fan, u64* nslots, double* per_kmer_out,
                                  double eff_parts = 0.0 /* pieces that share most k-mers count as fewer; 0: nparts */) {
    const int m = skm_minimizer_len(k);
    const u32 w = (u32)(k - m + 1);
    // k-mer instances per slot and rank: the pack kernel takes 1024 records of a slot; the owner's table (4096 entries)
    // has to hold the slot's distinct k-mers of ALL ranks — sized for unrelated groups: nparts x the per-rank mean
    // below three quarters of it
    const double per_kmer = 2.0 / (double)(w + 1) + 1.0 / 48.0;
    // k-mer instances per slot and rank.  The owner's table (4096 entries) takes ~3000 instances per round: with
    // groups of related genomes about 0.55 of a rank's instances survive the merge of identical records, so
    // 2600 / (0.55 x nparts) per rank keeps the owner at one round (unrelated genomes: two).  A rank's slot must also
    // fit the pack kernel's 1024 records, and the two partition levels give 512 x 512 slots at most (beyond: the
    // slots grow and the owner takes more rounds).
    // The regions of a rank's slots are sized as exp1_skm sizes them: the mean with the slack of five sigma, where the
    // `fan` genomes of a group bring their copies of a locus together (at least 1.7): that must stay below the pack
    // kernel's 1024 records.
    const double clump = 0.5 * (double)(w + 1) * (double)std::max<u32>(1, fan);
    auto region = [&](double mean) { return mean * per_kmer * std::max(1.7, 1.0 + 5.0 * std::sqrt(clump / mean)) + 128.0; };
    if (eff_parts <= 0.0) eff_parts = (double)nparts;
    double mean = std::min(2600.0 / (0.55 * eff_parts), (1024.0 - 128.0) / (1.7 * per_kmer));
    while (mean > 64.0 && region(mean) > 1008.0) mean *= 0.95;
    if (const char* e = getenv("KHOICE_SKM_EXCHANGE_MEAN")) mean = std::max(16.0, atof(e));   // tests: rounds on the owner
    u64 ns = std::max<u64>((u64)nparts, (u64)((double)std::max<u64>(1, positions_max) / mean) + 1);
    ns = std::min<u64>(ns, (u64)KH_SKM2_MAX_COARSE * KH_SKM_MAX_FINE);
    if (region((double)positions_max / (double)ns) > 1024.0) return false;
    *nslots = ns;
    *per_kmer_out = per_kmer;
    return true;
}
extern "C" int kh_skm_exchange_plan(kh_ctx* c, int k, uint64_t positions_max, uint32_t fan_max, int nparts, uint32_t* nslots,
                                    uint32_t* slots_per_part, uint64_t* part_cap) {
    if (!c || !nslots || !slots_per_part || !part_cap || nparts < 1) return kh_fail(KH_E_ARG, "kh_skm_exchange_plan: bad argument");
    if (!skm_exchange_k(k)) return kh_fail(KH_E_ARG, "the exchange form takes k = %d .. %d", KH_SKM_MIN_K, KH_SKM_MAX_K);
    u64 ns = 0;
    double per_kmer = 0;
    if (!skm_exchange_geometry(k, positions_max, nparts, fan_max, &ns, &per_kmer))
        return kh_fail(KH_E_CAPACITY, "too many k-mers per rank for the exchange form (%llu positions)", (unsigned long long)positions_max);
    *nslots = (u32)ns;
    *slots_per_part = (u32)((ns + nparts - 1) / nparts);
    *part_cap = ((u64)((double)positions_max * per_kmer * 1.3 / nparts) + 8192 + 63) & ~63ull;
    return KH_OK;
}

// fan_hint / inst_out / dup_out: the one-GPU use (a group of more than 64 genomes in sub-batches, exp1_big_group_skm):
// the tags are genomes of ONE group, their instance counts and the repeats under one tag are wanted.  soft: what does
// not fit is *done = false instead of an error.
static int skm_pack_impl(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens, int on_device,
                         const int* tag_of, int k, uint32_t nslots, int nparts, uint64_t part_cap, void* rec_out,
                         uint32_t* mask_out, uint32_t* count_out, uint32_t* off_out, uint64_t* part_n, u32 fan_hint,
                         std::vector<u64>* inst_out, u64* dup_out /* [32] host */, bool* soft_done, u32 nsub = 1,
                         SkmRecords* keep = nullptr /* the caller's: asks for and receives the side list */);
extern "C" int kh_skm_pack(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens, int on_device,
                           const int* tag_of, int k, uint32_t nslots, int nparts, uint64_t part_cap, void* rec_out,
                           uint32_t* mask_out, uint32_t* count_out, uint32_t* off_out, uint64_t* part_n) {
    return skm_pack_impl(c, nseq, seqs, lens, on_device, tag_of, k, nslots, nparts, part_cap, rec_out, mask_out, count_out, off_out,
                         part_n, 0, nullptr, nullptr, nullptr);
}
static int skm_pack_impl(kh_ctx* c, int nseq, const uint8_t* const* seqs, const uint64_t* lens, int on_device,
                         const int* tag_of, int k, uint32_t nslots, int nparts, uint64_t part_cap, void* rec_out,
                         uint32_t* mask_out, uint32_t* count_out, uint32_t* off_out, uint64_t* part_n, u32 fan_hint,
                         std::vector<u64>* inst_out, u64* dup_out, bool* soft_done, u32 nsub, SkmRecords* keep) {
    if (soft_done) *soft_done = false;
    if (!c || !seqs || !lens || !tag_of || nseq <= 0 || nparts < 1 || !rec_out || !mask_out || !count_out || !off_out || !part_n)
        return kh_fail(KH_E_ARG, "kh_skm_pack: bad argument");
    if (!skm_exchange_k(k)) return kh_fail(KH_E_ARG, "the exchange form takes k = %d .. %d", KH_SKM_MIN_K, KH_SKM_MAX_K);
    int ntags = 0;
    for (int i = 0; i < nseq; ++i) {
        if (tag_of[i] < 0 || tag_of[i] >= 32) return kh_fail(KH_E_ARG, "kh_skm_pack: tag %d outside 0..31", tag_of[i]);
        ntags = std::max(ntags, tag_of[i] + 1);
    }
    HIPCHK(hipSetDevice(c->dev));
    hipStream_t st = c->st;
    SkmRecords own;
    SkmRecords& rec = keep ? *keep : own;
    rec.force_slots = nslots;
    rec.fan_hint = fan_hint;
    rec.inst = inst_out;
    rec.want_spill = keep != nullptr;
    bool done = false;
    {   // every tag needs a sequence for the geometry code (groups without genomes are refused there): tags are dense here
        std::vector<int> seen(ntags, 0);
        for (int i = 0; i < nseq; ++i) seen[tag_of[i]] = 1;
        for (int t = 0; t < ntags; ++t)
            if (!seen[t]) return kh_fail(KH_E_ARG, "kh_skm_pack: tag %d has no sequence", t);
    }
    KHCHK(exp1_skm(c, nseq, seqs, lens, on_device, tag_of, ntags, k, 1, nullptr, nullptr, 2, nullptr, &done, /*by_group=*/true, &rec));
    if (!done) {
        if (soft_done) return KH_OK;
        return kh_fail(KH_E_CAPACITY, "kh_skm_pack: the records did not fit their regions (low-complexity input?)");
    }
    const u32 spp = (rec.nslots + (u32)nparts - 1) / (u32)nparts;
    Tmp d_ctl;
    nsub = std::max<u32>(1, nsub);
    // The caller wants every part without gaps (it travels): packed through 64 cursors into a buffer of our own and
    // moved together afterwards — one cursor per part is a queue of returning atomics (78 K slots: 0.8 ms).
    const bool compact = nsub == 1 && spp >= 2048 && !getenv("KHOICE_SKM_PACK_ONE_CURSOR");
    if (compact) nsub = 64;
    const size_t off_pn = (64 + 4 * (size_t)nparts * nsub + 7) & ~(size_t)7, off_dup = off_pn + ((4 * (size_t)nparts + 7) & ~(size_t)7),
                 ctl_bytes = off_dup + 8 * 32;
    Tmp d_tmp;
    if (compact) TMP_ALLOC(d_tmp, c, (size_t)nparts * part_cap * 20);
    TMP_ALLOC(d_ctl, c, ctl_bytes);
    HIPCHK(hipMemsetAsync(d_ctl.b->p, 0, ctl_bytes, st));
    HIPCHK(hipMemsetAsync(count_out, 0, 4 * (size_t)spp * nparts, st));   // (slots past the last one: nothing)
    HIPCHK(hipMemsetAsync(off_out, 0, 4 * (size_t)spp * nparts, st));
    KhSkmPackJob job;
    job.reg2 = reinterpret_cast<const uint4*>(rec.reg2->p);
    job.cur2 = rec.cur2;
    job.out_rec = compact ? d_tmp.as<uint4>() : static_cast<uint4*>(rec_out);
    job.out_mask = compact ? reinterpret_cast<u32*>(d_tmp.as<u8>() + 16 * (size_t)nparts * part_cap) : mask_out;
    job.part_cursor = reinterpret_cast<u32*>(d_ctl.as<u8>() + 64);
    job.slot_count = count_out;
    job.slot_off = off_out;
    job.ctl = d_ctl.as<u32>();
    job.dup = dup_out ? reinterpret_cast<unsigned long long*>(d_ctl.as<u8>() + off_dup) : nullptr;
    job.part_cap = part_cap;
    job.cap2 = rec.cap2;
    job.nslots = rec.nslots;
    job.spp = spp;
    job.nsub = nsub;
    c->prof_begin(KC_SKM_PACK);
    kh_launch_skm_pack(job, st);
    if (compact) {
        KhSkmCompactJob cj;
        cj.tmp_rec = job.out_rec;
        cj.tmp_mask = job.out_mask;
        cj.out_rec = static_cast<uint4*>(rec_out);
        cj.out_mask = mask_out;
        cj.cursors = job.part_cursor;
        cj.slot_off = off_out;
        cj.part_n = reinterpret_cast<u32*>(d_ctl.as<u8>() + off_pn);
        cj.part_cap = part_cap;
        cj.nslots = rec.nslots;
        cj.spp = spp;
        cj.nsub = nsub;
        cj.nparts = (u32)nparts;
        kh_launch_skm_pack_compact(cj, st);
    }
    c->prof_end();
    HIPCHK(hipGetLastError());
    std::vector<u32> h(ctl_bytes / 4);
    HIPCHK(hipMemcpyAsync(h.data(), d_ctl.b->p, ctl_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h[0] & KH_ERR_ORDER) return kh_fail(KH_E_INTERNAL, "kh_skm_pack: a record carried a tag above 31");
    if (h[0] & KH_ERR_CAPACITY) {
        if (soft_done) return KH_OK;
        return kh_fail(KH_E_CAPACITY, "kh_skm_pack: a slot or a part overflowed");
    }
    for (int p = 0; p < nparts; ++p) {   // (with several cursors per part: the records, which then lie with gaps)
        part_n[p] = 0;
        for (u32 q = 0; q < nsub; ++q) part_n[p] += h[16 + (size_t)p * nsub + q];
    }
    if (dup_out) memcpy(dup_out, reinterpret_cast<const u8*>(h.data()) + off_dup, 8 * 32);
    if (soft_done) *soft_done = true;
    return KH_OK;
}

static int skm_phased_impl(kh_ctx* c, int k, int npieces, const void* const* recs, const uint32_t* const* masks,
                           const uint32_t* const* counts, const uint32_t* const* offs, uint32_t nslots, uint32_t cs,
                           uint64_t* hist, uint32_t hist_len, u64* dup_out /* [npieces][32] host or null */, bool* soft_done,
                           const u32* dup_row = nullptr, const u32* join_next = nullptr, u32 share_q8 = 256);
extern "C" int kh_skm_phased_histogram(kh_ctx* c, int k, int npieces, const void* const* recs, const uint32_t* const* masks,
                                       const uint32_t* const* counts, const uint32_t* const* offs, uint32_t nslots, uint32_t cs,
                                       uint64_t* hist, uint32_t hist_len) {
    return skm_phased_impl(c, k, npieces, recs, masks, counts, offs, nslots, cs, hist, hist_len, nullptr, nullptr);
}
static int skm_phased_impl(kh_ctx* c, int k, int npieces, const void* const* recs, const uint32_t* const* masks,
                           const uint32_t* const* counts, const uint32_t* const* offs, uint32_t nslots, uint32_t cs,
                           uint64_t* hist, uint32_t hist_len, u64* dup_out, bool* soft_done, const u32* dup_row, const u32* join_next,
                           u32 share_q8) {
    if (soft_done) *soft_done = false;
    if (!c || npieces < 1 || !recs || !masks || !counts || !offs || !hist || hist_len < 2 || cs < 1)
        return kh_fail(KH_E_ARG, "kh_skm_phased_histogram: bad argument");
    if (!skm_exchange_k(k)) return kh_fail(KH_E_ARG, "the exchange form takes k = %d .. %d", KH_SKM_MIN_K, KH_SKM_MAX_K);
    HIPCHK(hipSetDevice(c->dev));
    hipStream_t st = c->st;
    const size_t off_pieces = 64, off_hist = (off_pieces + sizeof(KhSkmPiece) * (size_t)npieces + 63) & ~(size_t)63;
    const size_t off_pdup = off_hist + 8 * (size_t)hist_len, ws_bytes = off_pdup + (dup_out ? 8 * 32 * (size_t)npieces : 0);
    Tmp d_ws;
    TMP_ALLOC(d_ws, c, ws_bytes);
    std::vector<u8> up(off_hist, 0);
    KhSkmPiece* hp = reinterpret_cast<KhSkmPiece*>(up.data() + off_pieces);
    for (int i = 0; i < npieces; ++i) {
        hp[i].rec = static_cast<const uint4*>(recs[i]);
        hp[i].mask = masks[i];
        hp[i].count = counts[i];
        hp[i].off = offs[i];
        hp[i].dup_row = dup_row ? dup_row[i] : (u32)i;
        hp[i].join_next = join_next ? join_next[i] : 0u;
    }
    HIPCHK(hipMemsetAsync(d_ws.b->p, 0, ws_bytes, st));
    HIPCHK(hipMemcpyAsync(d_ws.b->p, up.data(), off_hist, hipMemcpyHostToDevice, st));
    KhSkmPhasedJob job;
    job.pieces = reinterpret_cast<const KhSkmPiece*>(d_ws.as<u8>() + off_pieces);
    job.hist = reinterpret_cast<unsigned long long*>(d_ws.as<u8>() + off_hist);
    job.ctl = d_ws.as<u32>();
    job.dup = dup_out ? reinterpret_cast<unsigned long long*>(d_ws.as<u8>() + off_pdup) : nullptr;
    job.npieces = (u32)npieces;
    job.share_q8 = std::min<u32>(256, std::max<u32>(1, share_q8));
    job.nslots = nslots;
    job.hist_len = hist_len;
    job.cs = cs;
    job.k = k;
    c->prof_begin(KC_SKM_PHASED);
    kh_launch_skm_phased(job, std::min<u32>(std::max<u32>(1, nslots), 2u * (u32)std::max(1, c->cus)), st);
    c->prof_end();
    HIPCHK(hipGetLastError());
    u32 h_ctl[4];
    HIPCHK(hipMemcpyAsync(h_ctl, d_ws.b->p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hist, d_ws.as<u8>() + off_hist, 8 * (size_t)hist_len, hipMemcpyDeviceToHost, st));
    if (dup_out) HIPCHK(hipMemcpyAsync(dup_out, d_ws.as<u8>() + off_pdup, 8 * 32 * (size_t)npieces, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h_ctl[0] & KH_ERR_CAPACITY) {
        if (soft_done) return KH_OK;
        return kh_fail(KH_E_CAPACITY, "kh_skm_phased_histogram: a slot held more k-mers than its table");
    }
    c->stat.setops++;
    if (soft_done) *soft_done = true;
    return KH_OK;
}

// One group of more than 64 genomes in the super-k-mer form (one-word keys): sub-batches of up to 32 genomes, each
// turned into records tagged with the genome's number in the sub-batch, identical records merged (k_skm_pack, one
// part); the sub-batches are the PHASES of one phased union (k_skm_phased): the counter of an entry ends as "in how
// many genomes of the group", which is the group's step_4 histogram.  Distinct k-mers of a genome: its instances minus
// the repeats under its tag (whole records at the merge, single k-mers at the insertion).  *done == false: the
// caller takes the key-array sub-batches.
static int exp1_big_group_skm(kh_ctx* c, int n, const uint8_t* const* seqs, const uint64_t* lens, int on_device, int k, u32 cs,
                              uint64_t* whist, u32 hist_len, uint64_t* dist /* [n] or null */, bool* done) {
    *done = false;
    if (!skm_exchange_k(k) || getenv("KHOICE_NO_SKM") || getenv("KHOICE_NO_SKM_PHASED") || !whist) return KH_OK;
    constexpr int SUB = 32;
    const int P = (n + SUB - 1) / SUB;
    if (P > (int)KH_SKM_PHASED_MAX_DUP_PIECES) return KH_OK;
    std::vector<int> first(P + 1, 0);
    for (int p = 0; p < P; ++p) first[p + 1] = first[p] + (n - first[p] + (P - p) - 1) / (P - p);   // balanced sizes
    u64 pos_max = 0;
    std::vector<u64> pos(P, 0);
    for (int p = 0; p < P; ++p) {
        for (int i = first[p]; i < first[p + 1]; ++i) pos[p] += lens[i] >= (u64)k ? lens[i] - k + 1 : 0;
        pos_max = std::max(pos_max, pos[p]);
    }
    if (!pos_max) return KH_OK;
    u64 ns = 0;
    double per_kmer = 0;
    const bool dbg = getenv("KHOICE_SKM_DEBUG") != nullptr;
    // How much the sub-batches of a group share decides how large a slot may be: its table holds one sub-batch's k-mers
    // plus what each further one adds (`novelty` of its own).  Measured: the synthetic genomes (1 % divergence from one
    // ancestor, independent per genome) add 0.7 — with 0.3 assumed a table overfilled and the repeated launch cost more
    // than the 2.5 x larger slots saved (12.1 -> 14.0 ms) — so the default assumes unrelated sub-batches (1.0); real
    // collections with a tree-like history share more: KHOICE_SKM_PHASED_NOVELTY (an overfilled table is caught and the
    // launch repeated with rounds sized for unrelated pieces).
    double novelty = 1.0;
    if (const char* e = getenv("KHOICE_SKM_PHASED_NOVELTY")) novelty = std::min(1.0, std::max(0.05, atof(e)));
    const double eff_parts = 1.0 + novelty * (double)(P - 1);
    const u32 share_q8 = (u32)std::min(256.0, std::ceil(256.0 * eff_parts / (double)P));
    if (!skm_exchange_geometry(k, pos_max, P, (u32)SUB, &ns, &per_kmer, eff_parts)) {
        if (dbg) fprintf(stderr, "[skm phased] %d genomes in %d phases: no geometry for %llu positions per phase\n", n, P, (unsigned long long)pos_max);
        return KH_OK;
    }
    const u32 nslots = (u32)ns;
    if (dbg) fprintf(stderr, "[skm phased] %d genomes in %d phases, %u slots, %.0f positions per slot and phase\n", n, P, nslots, (double)pos_max / nslots);
    HIPCHK(hipSetDevice(c->dev));
    std::vector<std::unique_ptr<Tmp>> bufs;
    std::vector<const void*> recs;
    std::vector<const uint32_t*> masks, counts, offs;
    std::vector<u32> rows, joins;   // per piece: the sub-batch it belongs to, "the next piece goes on in the same phase"
    std::vector<u64> inst_all(n, 0), dup_all(n, 0);
    hipStream_t st = c->st;
    for (int p = 0; p < P; ++p) {
        const int m = first[p + 1] - first[p];
        const u64 cap = ((u64)((double)pos[p] * per_kmer * 1.3) + 8192 + 63) & ~63ull;
        const size_t off_mask = 16 * (size_t)cap, off_count = off_mask + 4 * (size_t)cap, off_off = off_count + 4 * (size_t)((nslots + 3) & ~3u),
                     bytes = off_off + 4 * (size_t)((nslots + 3) & ~3u);
        bufs.emplace_back(new Tmp);
        TMP_ALLOC(*bufs.back(), c, bytes);
        u8* base = bufs.back()->as<u8>();
        std::vector<int> tag(m);
        for (int j = 0; j < m; ++j) tag[j] = j;
        std::vector<u64> inst;
        u64 dup[32], part_n = 0;
        bool ok = false;
        SkmRecords keep;
        KHCHK(skm_pack_impl(c, m, seqs + first[p], lens + first[p], on_device, tag.data(), k, nslots, 1, cap, base,
                            reinterpret_cast<u32*>(base + off_mask), reinterpret_cast<u32*>(base + off_count),
                            reinterpret_cast<u32*>(base + off_off), &part_n, (u32)m, &inst, dup, &ok, 64, &keep));
        if (!ok) {
            if (dbg) fprintf(stderr, "[skm phased] phase %d: the records did not fit\n", p);
            return KH_OK;
        }
        if (dbg) fprintf(stderr, "[skm phased] phase %d: %llu records travel (cap %llu), %u on the side list\n", p, (unsigned long long)part_n,
                         (unsigned long long)cap, keep.spill_n);
        recs.push_back(base);
        masks.push_back(reinterpret_cast<const u32*>(base + off_mask));
        counts.push_back(reinterpret_cast<const u32*>(base + off_count));
        offs.push_back(reinterpret_cast<const u32*>(base + off_off));
        rows.push_back((u32)p);
        joins.push_back(0);
        for (int j = 0; j < m; ++j) { inst_all[first[p] + j] = inst[j]; dup_all[first[p] + j] = dup[j]; }
        if (keep.spill_n) {
            // Overfull slots (low-complexity sequence): what the regions could not hold is on the side list, unmerged and
            // in no order.  It joins the sub-batch's phase as extra pieces — sorted by slot on the host (rare, small),
            // at most 1024 records of a slot per piece.
            const u32 ns = keep.spill_n;
            std::vector<uint4> hrec(ns);
            std::vector<u32> hslot(ns), order(ns);
            HIPCHK(hipMemcpyAsync(hrec.data(), keep.spill->p, 16 * (size_t)ns, hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(hslot.data(), static_cast<const u8*>(keep.spill->p) + 16 * (size_t)keep.spill_cap, 4 * (size_t)ns,
                                  hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (u32 i = 0; i < ns; ++i) {
                order[i] = i;
                if (hslot[i] >= nslots) return kh_fail(KH_E_INTERNAL, "side list: slot %u of %u", hslot[i], nslots);
            }
            std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return hslot[a] < hslot[b]; });
            std::vector<u32> run(nslots, 0);
            u32 longest = 0;
            for (u32 i = 0; i < ns; ++i) longest = std::max(longest, ++run[hslot[i]]);
            const u32 extra = (longest + 1023u) / 1024u;
            joins.back() = 1;
            for (u32 e = 0; e < extra; ++e) {
                std::vector<uint4> prec;
                std::vector<u32> pmask, pcount(nslots, 0), poff(nslots, 0);
                for (u32 i = 0; i < ns;) {   // the sorted list, a slot's run at a time
                    const u32 sl = hslot[order[i]], len = run[sl];
                    const u32 lo = std::min(len, e * 1024u), hi = std::min(len, (e + 1) * 1024u);
                    poff[sl] = (u32)prec.size();
                    pcount[sl] = hi - lo;
                    for (u32 j = lo; j < hi; ++j) {
                        const uint4 r = hrec[order[i + j]];
                        prec.push_back(r);
                        pmask.push_back(1u << ((r.w >> 21) & 31u));
                    }
                    i += len;
                }
                const size_t nrec_e = std::max<size_t>(1, prec.size());
                const size_t e_mask = 16 * nrec_e, e_count = e_mask + ((4 * nrec_e + 15) & ~(size_t)15), e_off = e_count + 4 * (size_t)((nslots + 3) & ~3u),
                             e_bytes = e_off + 4 * (size_t)((nslots + 3) & ~3u);
                bufs.emplace_back(new Tmp);
                TMP_ALLOC(*bufs.back(), c, e_bytes);
                u8* eb = bufs.back()->as<u8>();
                if (!prec.empty()) {
                    HIPCHK(hipMemcpyAsync(eb, prec.data(), 16 * prec.size(), hipMemcpyHostToDevice, st));
                    HIPCHK(hipMemcpyAsync(eb + e_mask, pmask.data(), 4 * pmask.size(), hipMemcpyHostToDevice, st));
                }
                HIPCHK(hipMemcpyAsync(eb + e_count, pcount.data(), 4 * (size_t)nslots, hipMemcpyHostToDevice, st));
                HIPCHK(hipMemcpyAsync(eb + e_off, poff.data(), 4 * (size_t)nslots, hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));   // (the host vectors go out of scope)
                recs.push_back(eb);
                masks.push_back(reinterpret_cast<const u32*>(eb + e_mask));
                counts.push_back(reinterpret_cast<const u32*>(eb + e_count));
                offs.push_back(reinterpret_cast<const u32*>(eb + e_off));
                rows.push_back((u32)p);
                joins.push_back(e + 1 < extra ? 1u : 0u);
            }
            c->stat.big_slots += (u64)std::count_if(run.begin(), run.end(), [](u32 v) { return v != 0; });
        }
    }
    std::vector<u64> pdup((size_t)recs.size() * 32, 0), hist(hist_len, 0);
    bool ok = false;
    KHCHK(skm_phased_impl(c, k, (int)recs.size(), recs.data(), masks.data(), counts.data(), offs.data(), nslots, cs, hist.data(), hist_len,
                          pdup.data(), &ok, rows.data(), joins.data(), share_q8));
    if (!ok && share_q8 < 256) {   // a table overfilled: the genomes share less than assumed — rounds for unrelated pieces
        if (dbg) fprintf(stderr, "[skm phased] a table overfilled with rounds sized for shared k-mers: once more\n");
        c->stat.retries++;
        KHCHK(skm_phased_impl(c, k, (int)recs.size(), recs.data(), masks.data(), counts.data(), offs.data(), nslots, cs, hist.data(),
                              hist_len, pdup.data(), &ok, rows.data(), joins.data(), 256));
    }
    if (!ok) {
        if (dbg) fprintf(stderr, "[skm phased] the phased union overflowed\n");
        return KH_OK;
    }
    memcpy(whist, hist.data(), 8 * (size_t)hist_len);
    u64 bases = 0, isum = 0, dsum = 0;
    for (int p = 0; p < P; ++p)
        for (int i = first[p]; i < first[p + 1]; ++i) {
            const u64 d = inst_all[i] - dup_all[i] - pdup[(size_t)p * 32 + (i - first[p])];
            if (dist) dist[i] = d;
            bases += lens[i]; isum += inst_all[i]; dsum += d;
        }
    c->stat.bases += bases;
    c->stat.builds += n;
    c->stat.kmers += isum;
    c->stat.distinct += dsum;
    c->stat.setop_in += dsum;
    *done = true;
    return KH_OK;
}

// ------------------------------------------------------------------------------ host mix
extern "C" void kh_mix_host(int k, const uint64_t* in, uint64_t* out) {
    if (k <= 32) {
        KmerKey<1> a{in[0]};
        a = kh_mix(a, k);
        out[0] = a.lo;
    } else {
        KmerKey<2> a{in[0], in[1]};
        a = kh_mix(a, k);
        out[0] = a.lo;
        out[1] = a.hi;
    }
}
extern "C" void kh_unmix_host(int k, const uint64_t* in, uint64_t* out) {
    if (k <= 32) {
        KmerKey<1> a{in[0]};
        a = kh_unmix(a, k);
        out[0] = a.lo;
    } else {
        KmerKey<2> a{in[0], in[1]};
        a = kh_unmix(a, k);
        out[0] = a.lo;
        out[1] = a.hi;
    }
}
