// khoice_amd — the step_7/8 exchange of experiment type 1 behind the C ABI (SURVEY.md §8b
// `kh_comm_init`, §8e): RCCL over xGMI without Python in the way.  Same protocol as
// khoice_amd/dist.py::across_set_exchange:
//   every rank holds its LOCAL across-group set (keys sorted by mixed key + counter = number of
//   its groups holding the k-mer, what the fused kh_exp1_run emits);
//   the mixed key space is cut into `nranks` equal-width slots, so what rank j owns of a set is one
//   contiguous slice and the set's own storage is the send buffer;
//   all-gather of the slice bounds -> grouped ncclSend / ncclRecv of keys and counters (a full-mesh
//   pattern: every xGMI link carries one peer's stream) -> one counter-summing union with the
//   histogram fused -> ncclAllReduce(sum) of the histogram.
// RCCL is loaded with dlopen on the first kh_comm_* call: processes that never leave one GPU
// (the CLIs, the tests) do not pay for it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/khoice_hip.h"
#include "kh_engine.h"
#include "kh_launch.h"

namespace {
struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

bool rccl_load() {
    std::call_once(g_rccl_once, []() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.so) break;
        }
        if (!g_rccl.so) { g_rccl.error = std::string("cannot load librccl: ") + dlerror(); return; }
#define KH_SYM(field, sym)                                                                  \
        g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.so, #sym));    \
        if (!g_rccl.field && g_rccl.error.empty()) g_rccl.error = "librccl lacks " #sym;
        KH_SYM(GetUniqueId, ncclGetUniqueId)
        KH_SYM(CommInitRank, ncclCommInitRank)
        KH_SYM(CommDestroy, ncclCommDestroy)
        KH_SYM(GetErrorString, ncclGetErrorString)
        KH_SYM(AllGather, ncclAllGather)
        KH_SYM(AllReduce, ncclAllReduce)
        KH_SYM(Send, ncclSend)
        KH_SYM(Recv, ncclRecv)
        KH_SYM(GroupStart, ncclGroupStart)
        KH_SYM(GroupEnd, ncclGroupEnd)
#undef KH_SYM
    });
    return g_rccl.error.empty();
}
}  // namespace

struct kh_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
};

#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r__ = (expr);                                                                 \
        if (r__ != ncclSuccess) return kh_fail(KH_E_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r__)); \
    } while (0)
#define HIPCHK2(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return kh_fail(KH_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

extern "C" int kh_comm_unique_id(char id[KH_COMM_ID_BYTES]) {
    if (!id) return kh_fail(KH_E_ARG, "kh_comm_unique_id: NULL argument");
    if (!rccl_load()) return kh_fail(KH_E_HIP, "%s", g_rccl.error.c_str());
    static_assert(sizeof(ncclUniqueId) == KH_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return KH_OK;
}

extern "C" int kh_comm_init(kh_ctx* c, int rank, int nranks, const char id[KH_COMM_ID_BYTES], kh_comm** out) {
    if (!c || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return kh_fail(KH_E_ARG, "kh_comm_init: bad argument");
    if (!rccl_load()) return kh_fail(KH_E_HIP, "%s", g_rccl.error.c_str());
    HIPCHK2(hipSetDevice(c->dev));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    kh_comm* k = new kh_comm;
    k->rank = rank;
    k->nranks = nranks;
    const ncclResult_t r = g_rccl.CommInitRank(&k->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete k;
        return kh_fail(KH_E_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    *out = k;
    return KH_OK;
}

extern "C" void kh_comm_destroy(kh_comm* k) {
    if (!k) return;
    if (k->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(k->comm);
    delete k;
}

extern "C" int kh_across_exchange_histogram(kh_ctx* c, kh_comm* k, const kh_set* local, uint32_t cs, uint64_t* hist,
                                            uint32_t hist_len) {
    if (!c || !k || !local || !hist || hist_len < 2 || cs < 1) return kh_fail(KH_E_ARG, "kh_across_exchange_histogram: bad argument");
    HIPCHK2(hipSetDevice(c->dev));
    hipStream_t st = c->st;
    const int P = k->nranks, me = k->rank;
    const size_t kb = 8 * (size_t)local->W;
    // ---- slice bounds of the local set, then everybody's
    std::vector<uint64_t> myb((size_t)P + 1);
    int r = kh_set_partition_bounds(c, local, (uint32_t)P, myb.data());
    if (r != KH_OK) return r;
    struct Dev {
        void* p = nullptr;
        ~Dev() { if (p) (void)hipFree(p); }
    } d_myb, d_allb, d_cnt_fill, d_rkeys, d_rcnt, d_hist;
    const size_t bwords = (size_t)P + 1;
    HIPCHK2(hipMalloc(&d_myb.p, 8 * bwords));
    HIPCHK2(hipMalloc(&d_allb.p, 8 * bwords * P));
    HIPCHK2(hipMemcpyAsync(d_myb.p, myb.data(), 8 * bwords, hipMemcpyHostToDevice, st));
    NCCLCHK(g_rccl.AllGather(d_myb.p, d_allb.p, bwords, ncclUint64, k->comm, st));
    std::vector<uint64_t> allb(bwords * P);
    HIPCHK2(hipMemcpyAsync(allb.data(), d_allb.p, 8 * bwords * P, hipMemcpyDeviceToHost, st));
    HIPCHK2(hipStreamSynchronize(st));
    // what rank p sends to me: its slice [allb[p][me], allb[p][me + 1])
    std::vector<uint64_t> rn(P), roff(P + 1, 0);
    for (int p = 0; p < P; ++p) {
        rn[p] = allb[(size_t)p * bwords + me + 1] - allb[(size_t)p * bwords + me];
        roff[p + 1] = roff[p] + rn[p];
    }
    const uint64_t rtotal = roff[P];
    // ---- counters to send: the set's own array, or a materialised uniform counter
    const uint32_t* scnt = local->counts_ptr();
    if (!scnt && local->n) {
        HIPCHK2(hipMalloc(&d_cnt_fill.p, 4 * local->n));
        kh_launch_fill_u32(static_cast<u32*>(d_cnt_fill.p), local->n, local->uniform, st);
        scnt = static_cast<const uint32_t*>(d_cnt_fill.p);
    }
    HIPCHK2(hipMalloc(&d_rkeys.p, std::max<size_t>(16, kb * rtotal)));
    HIPCHK2(hipMalloc(&d_rcnt.p, std::max<size_t>(16, 4 * rtotal)));
    // ---- keys and counters: one grouped exchange (the own slice travels device-to-device inside it)
    const uint8_t* skeys = static_cast<const uint8_t*>(local->keys_ptr());
    NCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < P; ++p) {
        const uint64_t lo = myb[p], n = myb[p + 1] - myb[p];
        if (n) {
            NCCLCHK(g_rccl.Send(skeys + lo * kb, n * local->W, ncclUint64, p, k->comm, st));
            NCCLCHK(g_rccl.Send(scnt + lo, n, ncclUint32, p, k->comm, st));
        }
        if (rn[p]) {
            NCCLCHK(g_rccl.Recv(static_cast<uint8_t*>(d_rkeys.p) + roff[p] * kb, rn[p] * local->W, ncclUint64, p, k->comm, st));
            NCCLCHK(g_rccl.Recv(static_cast<uint32_t*>(d_rcnt.p) + roff[p], rn[p], ncclUint32, p, k->comm, st));
        }
    }
    NCCLCHK(g_rccl.GroupEnd());
    // ---- every received slice is sorted, distinct and inside this rank's slot: sum them in one pass
    std::vector<kh_set*> slices;
    auto drop = [&]() { for (auto* s : slices) kh_set_free(s); };
    for (int p = 0; p < P; ++p) {
        if (!rn[p]) continue;
        kh_set* s = nullptr;
        r = kh_set_wrap_device(c, local->k, rn[p], static_cast<uint8_t*>(d_rkeys.p) + roff[p] * kb,
                               static_cast<uint32_t*>(d_rcnt.p) + roff[p], 1, &s);
        if (r != KH_OK) { drop(); return r; }
        slices.push_back(s);
    }
    std::vector<uint64_t> mine(hist_len, 0);
    if (!slices.empty()) {
        r = kh_union_histogram(c, slices.data(), (int)slices.size(), cs, mine.data(), hist_len);   // runs on the same stream, behind the exchange
        if (r != KH_OK) { drop(); return r; }
    }
    drop();
    // ---- global histogram
    HIPCHK2(hipMalloc(&d_hist.p, 8 * (size_t)hist_len));
    HIPCHK2(hipMemcpyAsync(d_hist.p, mine.data(), 8 * (size_t)hist_len, hipMemcpyHostToDevice, st));
    NCCLCHK(g_rccl.AllReduce(d_hist.p, d_hist.p, hist_len, ncclUint64, ncclSum, k->comm, st));
    HIPCHK2(hipMemcpyAsync(hist, d_hist.p, 8 * (size_t)hist_len, hipMemcpyDeviceToHost, st));
    HIPCHK2(hipStreamSynchronize(st));
    return KH_OK;
}
