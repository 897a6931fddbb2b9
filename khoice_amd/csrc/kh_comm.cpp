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
    decltype(&ncclCommAbort) CommAbort = nullptr;   // (optional)
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
        g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.so, "ncclCommAbort"));
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

// All ranks go through the SAME three collectives whatever happens locally: a rank that fails before one of them
// (slice bounds, a small allocation, a set that cannot be wrapped) takes part with empty slices and raises a flag
// that travels with the histogram's all-reduce, so that every rank returns an error together instead of one rank
// leaving while its peers block in the next collective.  Inside the grouped exchange the first error is kept and
// ncclGroupEnd is still called.  Only a rank that cannot even allocate its receive buffers aborts the communicator
// (its peers then fail in RCCL rather than hang).  Buffers come from the context's pool.
extern "C" int kh_across_exchange_histogram(kh_ctx* c, kh_comm* k, const kh_set* local, uint32_t cs, uint64_t* hist,
                                            uint32_t hist_len) {
    if (!c || !k || !local || !hist || hist_len < 2 || cs < 1) return kh_fail(KH_E_ARG, "kh_across_exchange_histogram: bad argument");
    hipStream_t st = c->st;
    const int P = k->nranks, me = k->rank;
    const size_t kb = 8 * (size_t)local->W;
    int first_err = KH_OK;
    std::string first_msg;
    auto note = [&](int code, const char* what) {   // keep the first local failure, go on
        if (first_err == KH_OK) { first_err = code; first_msg = what ? what : kh_last_error(); }
    };
    auto note_hip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess) note(KH_E_HIP, (std::string(what) + " failed: " + hipGetErrorString(e)).c_str());
    };
    auto note_nccl = [&](ncclResult_t e, const char* what) {
        if (e != ncclSuccess) note(KH_E_HIP, (std::string(what) + " failed: " + g_rccl.GetErrorString(e)).c_str());
    };
    struct Buf {
        DevBuf* b = nullptr;
        ~Buf() { buf_unref(b); }
        void* p() const { return b ? b->p : nullptr; }
    } d_myb, d_allb, d_cnt_fill, d_rkeys, d_rcnt, d_hist;
    note_hip(hipSetDevice(c->dev), "hipSetDevice");
    // ---- slice bounds of the local set (none on failure), then everybody's
    const size_t bwords = (size_t)P + 1;
    std::vector<uint64_t> myb(bwords, 0);
    if (first_err == KH_OK) {
        const int r = kh_set_partition_bounds(c, local, (uint32_t)P, myb.data());
        if (r != KH_OK) { note(r, nullptr); std::fill(myb.begin(), myb.end(), 0); }
    }
    d_myb.b = c->buf_alloc(8 * bwords);
    d_allb.b = c->buf_alloc(8 * bwords * P);
    d_hist.b = c->buf_alloc(8 * ((size_t)hist_len + 1));
    if (!d_myb.b || !d_allb.b || !d_hist.b) {   // not even a few hundred bytes: this rank cannot take part at all
        if (g_rccl.CommAbort) (void)g_rccl.CommAbort(k->comm);
        k->comm = nullptr;
        return kh_fail(KH_E_NOMEM, "kh_across_exchange_histogram: device allocation failed; communicator aborted");
    }
    note_hip(hipMemcpyAsync(d_myb.p(), myb.data(), 8 * bwords, hipMemcpyHostToDevice, st), "upload of the slice bounds");
    note_nccl(g_rccl.AllGather(d_myb.p(), d_allb.p(), bwords, ncclUint64, k->comm, st), "ncclAllGather");
    std::vector<uint64_t> allb(bwords * P, 0);
    note_hip(hipMemcpyAsync(allb.data(), d_allb.p(), 8 * bwords * P, hipMemcpyDeviceToHost, st), "download of the slice bounds");
    note_hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    // what rank p sends to me: its slice [allb[p][me], allb[p][me + 1])
    std::vector<uint64_t> rn(P), roff(P + 1, 0);
    for (int p = 0; p < P; ++p) {
        const uint64_t lo = allb[(size_t)p * bwords + me], hi = allb[(size_t)p * bwords + me + 1];
        rn[p] = hi >= lo ? hi - lo : 0;
        roff[p + 1] = roff[p] + rn[p];
    }
    const uint64_t rtotal = roff[P];
    // ---- counters to send: the set's own array, or a materialised uniform counter
    const uint32_t* scnt = local->counts_ptr();
    const uint64_t nsend = myb[P];   // 0 after a local failure: nothing is sent (the peers were told so)
    if (!scnt && nsend) {
        d_cnt_fill.b = c->buf_alloc(4 * local->n);
        if (d_cnt_fill.b) {
            kh_launch_fill_u32(static_cast<u32*>(d_cnt_fill.p()), local->n, local->uniform, st);
            scnt = static_cast<const uint32_t*>(d_cnt_fill.p());
        }
    }
    d_rkeys.b = c->buf_alloc(std::max<size_t>(16, kb * rtotal));
    d_rcnt.b = c->buf_alloc(std::max<size_t>(16, 4 * rtotal));
    if (!d_rkeys.b || !d_rcnt.b || (!scnt && nsend)) {
        // the peers' sends are already decided (they have seen my bounds): without buffers they cannot be matched
        if (g_rccl.CommAbort) (void)g_rccl.CommAbort(k->comm);
        k->comm = nullptr;
        return kh_fail(KH_E_NOMEM, "kh_across_exchange_histogram: no device memory for %llu received records; communicator aborted",
                       (unsigned long long)rtotal);
    }
    // ---- keys and counters: one grouped exchange (the own slice travels device-to-device inside it)
    const uint8_t* skeys = static_cast<const uint8_t*>(local->keys_ptr());
    {
        ncclResult_t gerr = g_rccl.GroupStart();
        for (int p = 0; p < P && gerr == ncclSuccess; ++p) {
            const uint64_t lo = myb[p], n = myb[p + 1] - myb[p];
            if (n) {
                gerr = g_rccl.Send(skeys + lo * kb, n * local->W, ncclUint64, p, k->comm, st);
                if (gerr == ncclSuccess) gerr = g_rccl.Send(scnt + lo, n, ncclUint32, p, k->comm, st);
            }
            if (rn[p] && gerr == ncclSuccess) {
                gerr = g_rccl.Recv(static_cast<uint8_t*>(d_rkeys.p()) + roff[p] * kb, rn[p] * local->W, ncclUint64, p, k->comm, st);
                if (gerr == ncclSuccess) gerr = g_rccl.Recv(static_cast<uint32_t*>(d_rcnt.p()) + roff[p], rn[p], ncclUint32, p, k->comm, st);
            }
        }
        const ncclResult_t eend = g_rccl.GroupEnd();   // always: an open group would poison the communicator
        note_nccl(gerr, "ncclSend / ncclRecv");
        note_nccl(eend, "ncclGroupEnd");
    }
    // ---- every received slice is sorted, distinct and inside this rank's slot: sum them in one pass
    std::vector<uint64_t> mine((size_t)hist_len + 1, 0);
    if (first_err == KH_OK) {
        std::vector<kh_set*> slices;
        for (int p = 0; p < P && first_err == KH_OK; ++p) {
            if (!rn[p]) continue;
            kh_set* s = nullptr;
            const int r = kh_set_wrap_device(c, local->k, rn[p], static_cast<uint8_t*>(d_rkeys.p()) + roff[p] * kb,
                                             static_cast<uint32_t*>(d_rcnt.p()) + roff[p], 1, &s);
            if (r != KH_OK) note(r, nullptr);
            else slices.push_back(s);
        }
        if (first_err == KH_OK && !slices.empty()) {
            const int r = kh_union_histogram(c, slices.data(), (int)slices.size(), cs, mine.data(), hist_len);   // on the same stream, behind the exchange
            if (r != KH_OK) note(r, nullptr);
        }
        for (auto* s : slices) kh_set_free(s);
    }
    if (first_err != KH_OK) std::fill(mine.begin(), mine.end(), 0);
    mine[hist_len] = first_err != KH_OK ? 1 : 0;   // the flag every rank will see
    // ---- global histogram (+ the number of ranks that failed)
    note_hip(hipMemcpyAsync(d_hist.p(), mine.data(), 8 * ((size_t)hist_len + 1), hipMemcpyHostToDevice, st), "upload of the histogram");
    note_nccl(g_rccl.AllReduce(d_hist.p(), d_hist.p(), (size_t)hist_len + 1, ncclUint64, ncclSum, k->comm, st), "ncclAllReduce");
    note_hip(hipMemcpyAsync(mine.data(), d_hist.p(), 8 * ((size_t)hist_len + 1), hipMemcpyDeviceToHost, st), "download of the histogram");
    note_hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (first_err != KH_OK) return kh_fail(first_err, "%s", first_msg.c_str());
    if (mine[hist_len]) return kh_fail(KH_E_INTERNAL, "kh_across_exchange_histogram: %llu other rank(s) failed", (unsigned long long)mine[hist_len]);
    memcpy(hist, mine.data(), 8 * (size_t)hist_len);
    return KH_OK;
}
