// khoice_amd — internal host-side types of libkhoice_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "kh_common.h"

struct kh_ctx;

struct Pool {
    std::multimap<size_t, void*> free_;
    size_t total_bytes = 0, cached_bytes = 0;
    void* alloc(size_t bytes, size_t* got);
    void release(void* p, size_t bytes);
    void trim();
};

struct DevBuf {   // reference-counted device allocation, shared by the sets cut out of it
    void* p;
    size_t bytes;
    std::atomic<int> refs;
    kh_ctx* ctx;
};
void buf_ref(DevBuf* b);
void buf_unref(DevBuf* b);

enum KernelClass {
    KC_EXTRACT_HIST = 0, KC_BUCKET_PLAN, KC_EXTRACT_SCATTER, KC_BUCKET_SORT, KC_RANGE_BOUNDS,
    KC_SETOP, KC_HISTOGRAM, KC_REMIX, KC_COPY_IN, KC_UNION_TAGGED, KC_SKM_SCATTER, KC_SKM_REGROUP, KC_SKM_UNION,
    KC_SKM_BIG, KC_SKM_PACK, KC_SKM_PHASED, KC_COUNT
};
struct ProfEvt { int cls; hipEvent_t a, b; };
struct Stats {
    u64 builds = 0, bases = 0, kmers = 0, distinct = 0, setops = 0, setop_in = 0, setop_out = 0,
        retries = 0, order_fallbacks = 0, skm_records = 0, big_slots = 0;
};

struct kh_ctx {
    int dev = 0;
    int cus = 0;
    std::string arch;
    hipStream_t st = nullptr;
    Pool pool;
    Stats stat;
    bool profile = false;
    std::vector<ProfEvt> evts;
    std::vector<hipEvent_t> free_events;
    double cls_ms[KC_COUNT] = {0};
    u64 cls_n[KC_COUNT] = {0};
    // Sets may outlive kh_ctx_destroy (a binding's garbage collector frees them late): the
    // context object stays behind as a closed shell until its last buffer is released.
    std::atomic<long> live_bufs{0};
    bool closed = false;
    bool dynamic_order = false;   // KhLookback::dynamic for this context's launches
    DevBuf* buf_alloc(size_t bytes);
    // pinned host staging (small read-backs and descriptor uploads must not be pageable:
    // a pageable hipMemcpyAsync waits for the stream and would serialise queued operations)
    std::multimap<size_t, void*> pinned_free;
    void* pin_alloc(size_t bytes, size_t* got);
    void pin_release(void* p, size_t bytes);
    void prof_begin(int cls);
    void prof_end();
    void prof_collect();
};

// A k-mer database resident in HBM: n distinct MIXED keys sorted ascending, and either a
// counter per key (cb) or one uniform counter for all of them.
struct kh_set {
    int k, W;
    u64 n;
    DevBuf* kb;
    size_t koff;   // byte offset of key 0 inside kb
    DevBuf* cb;    // nullptr => uniform counter
    size_t coff;
    u32 uniform;
    u32 counter_max;   // saturation value the counters were produced with (histogram range)
    const void* keys_ptr() const { return kb ? static_cast<const u8*>(kb->p) + koff : nullptr; }
    const u32* counts_ptr() const {
        return cb ? reinterpret_cast<const u32*>(static_cast<const u8*>(cb->p) + coff) : nullptr;
    }
};

int kh_fail(int code, const char* fmt, ...);
int kh_set_from_mixed_host(kh_ctx* c, int k, u64 n, const void* keys_mixed_sorted, const u32* counts,
                           u32 uniform, u32 counter_max, kh_set** out);
