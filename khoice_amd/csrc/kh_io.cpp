// khoice_amd — file side of the drop-in boundary: (gz) multi-FASTA ingest, the
// <prefix>.kmc_pre/.kmc_suf container, histogram and sorted-dump text files.
// File names and text formats are the ones khoice's Snakemake rules and consumers expect
// (workflow/rules/exp_type_1.smk:160-161,210-212; src/merge_lists.py:19-22).
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/khoice_hip.h"
#include "kh_engine.h"

// ------------------------------------------------------------------------------ FASTA
// -fm semantics: '>' at line start opens a header line, sequence lines of
// a record are concatenated, k-mers never span records.  The cleaned text keeps every
// sequence byte verbatim (so N / IUPAC / anything non-ACGT still breaks runs on the device)
// and puts one '\n' between records.
extern "C" int kh_read_fasta(const char* path, uint8_t** seq, uint64_t* len) {
    if (!path || !seq || !len) return kh_fail(KH_E_ARG, "kh_read_fasta: NULL argument");
    gzFile f = gzopen(path, "rb");   // transparently reads plain files too
    if (!f) return kh_fail(KH_E_IO, "cannot open %s", path);
    gzbuffer(f, 1 << 20);
    size_t cap = 1 << 24, n = 0;
    uint8_t* out = static_cast<uint8_t*>(malloc(cap));
    if (!out) { gzclose(f); return kh_fail(KH_E_NOMEM, "out of host memory"); }
    std::vector<uint8_t> buf(1 << 20);
    bool line_start = true, in_header = false, have_record = false;
    for (;;) {
        const int got = gzread(f, buf.data(), (unsigned)buf.size());
        if (got < 0) {
            int e;
            const char* msg = gzerror(f, &e);
            free(out);
            gzclose(f);
            return kh_fail(KH_E_IO, "read error in %s: %s", path, msg);
        }
        if (got == 0) break;
        if (n + (size_t)got + 2 > cap) {
            while (n + (size_t)got + 2 > cap) cap *= 2;
            uint8_t* p = static_cast<uint8_t*>(realloc(out, cap));
            if (!p) { free(out); gzclose(f); return kh_fail(KH_E_NOMEM, "out of host memory"); }
            out = p;
        }
        for (int i = 0; i < got; ++i) {
            const uint8_t ch = buf[i];
            if (in_header) {
                if (ch == '\n') { in_header = false; line_start = true; }
                continue;
            }
            if (ch == '\n') { line_start = true; continue; }
            if (ch == '\r') continue;
            if (line_start && ch == '>') {
                in_header = true;
                if (have_record && n && out[n - 1] != '\n') out[n++] = '\n';
                have_record = true;
                continue;
            }
            line_start = false;
            have_record = true;
            out[n++] = ch;
        }
    }
    gzclose(f);
    *seq = out;
    *len = n;
    return KH_OK;
}
extern "C" void kh_free_host(void* p) { free(p); }

extern "C" int kh_build_fasta(kh_ctx* c, const char* path, int k, uint32_t ci, uint32_t cx, uint32_t cs,
                              kh_set** out) {
    if (!c || !out) return kh_fail(KH_E_ARG, "kh_build_fasta: NULL argument");
    uint8_t* seq = nullptr;
    uint64_t len = 0;
    int r = kh_read_fasta(path, &seq, &len);
    if (r != KH_OK) return r;
    const uint8_t* seqs[1] = {seq};
    r = kh_build_batch(c, 1, seqs, &len, 0, k, ci, cx, cs, 1, out);
    free(seq);
    return r;
}

// ------------------------------------------------------------------------------ batched ingest
// kh_ingest_fasta: many (gz) multi-FASTA files -> cleaned sequence text resident in HBM, the form
// kh_build_batch / kh_exp1_run read in place (inputs of exp_type_1.smk:44-47,158).
//   worker threads   inflate one file each into a host buffer (sized from the gz trailer) and sweep its line starts with memchr: one "the line running into
//                    this 4 KB tile is a header line" flag per tile — the only sequential part of
//                    FASTA cleaning;
//   calling thread   as files complete (any order): one H2D copy of raw bytes + flags, then the
//                    three small kernels of kh_ingest.hip (count, scan, write) on the context's
//                    stream, while the other files are still being inflated.
// The result is byte-identical to kh_read_fasta's text (tests/test_gpu_ingest.py).
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

size_t kh_fasta_clean_workspace(u64 raw_len);
u32 kh_fasta_tile_bytes();
void kh_launch_fasta_clean(const u8* raw, u64 n, const u8* entry_hdr, u8* out, void* ws,
                           unsigned long long* out_len, hipStream_t st);

struct kh_seqs {
    std::vector<DevBuf*> bufs;
    std::vector<uint64_t> lens;
};

namespace {
struct IngestFile {
    std::string path;
    uint8_t* host = nullptr;      // [raw bytes][pad to 16][entry flags]
    size_t cap = 0;
    uint64_t raw_len = 0, flags_off = 0, ntiles = 0;
    std::string error;
};

// expected size of the inflated data: the ISIZE trailer of a single-member gzip file, the file
// size for plain text; 0 = unknown (the buffer then grows as needed)
uint64_t inflated_size_hint(const char* path, bool* is_gz) {
    *is_gz = false;
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    unsigned char magic[2] = {0, 0};
    const size_t got = fread(magic, 1, 2, f);
    struct stat sb;
    uint64_t hint = 0;
    if (fstat(fileno(f), &sb) == 0) {
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            *is_gz = true;
            unsigned char t[4];
            if (sb.st_size >= 18 && fseek(f, -4, SEEK_END) == 0 && fread(t, 1, 4, f) == 4) {
                hint = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
                if (hint < (uint64_t)sb.st_size) hint = 0;   // multi-member or > 4 GiB: the trailer is no guide
            }
        } else {
            hint = (uint64_t)sb.st_size;
        }
    }
    fclose(f);
    return hint;
}

// inflate into f.host (grown through `grow` when the hint was short), then the per-tile flags
void ingest_read(IngestFile& f, const std::function<bool(IngestFile&, size_t)>& grow) {
    gzFile g = gzopen(f.path.c_str(), "rb");
    if (!g) { f.error = "cannot open " + f.path; return; }
    gzbuffer(g, 1 << 20);
    const uint32_t tile = kh_fasta_tile_bytes();
    uint64_t n = 0;
    for (;;) {
        // room for the next chunk, the padding and the flags that follow the data
        const size_t need = (size_t)n + (4u << 20) + 64 + (size_t)((n + (4u << 20)) / tile + 2);
        if (need > f.cap && !grow(f, need + need / 4)) { f.error = "out of host memory"; gzclose(g); return; }
        const int got = gzread(g, f.host + n, 4u << 20);
        if (got < 0) {
            int e;
            f.error = std::string("read error in ") + f.path + ": " + gzerror(g, &e);
            gzclose(g);
            return;
        }
        if (got == 0) break;
        n += (uint64_t)got;
    }
    gzclose(g);
    f.raw_len = n;
    f.flags_off = (n + 15) & ~15ull;
    f.ntiles = (n + tile - 1) / tile;
    memset(f.host + n, '\n', (size_t)(f.flags_off - n));
    uint8_t* flags = f.host + f.flags_off;
    memset(flags, 0, (size_t)f.ntiles + 1);
    // line sweep: a tile boundary p with ls < p <= le lies inside the line [ls, le]
    const uint8_t* raw = f.host;
    uint64_t ls = 0;
    while (ls < n) {
        const void* nl = memchr(raw + ls, '\n', (size_t)(n - ls));
        const uint64_t le = nl ? (uint64_t)(static_cast<const uint8_t*>(nl) - raw) : n - 1;
        uint64_t q = ls;
        while (q <= le && raw[q] == '\r') ++q;
        const bool hdr = q <= le && raw[q] == '>';
        if (hdr)
            for (uint64_t t = ls / tile + 1; t * tile <= le && t < f.ntiles; ++t) flags[t] = 1;
        ls = le + 1;
    }
}
}  // namespace

extern "C" int kh_ingest_fasta(kh_ctx* c, int nfiles, const char* const* paths, int nthreads, kh_seqs** out) {
    if (!c || !paths || !out || nfiles <= 0) return kh_fail(KH_E_ARG, "kh_ingest_fasta: bad argument");
    if (hipSetDevice(c->dev) != hipSuccess) return kh_fail(KH_E_HIP, "hipSetDevice failed");
    if (nthreads <= 0) nthreads = (int)std::min<unsigned>(32, std::max(1u, std::thread::hardware_concurrency()));
    nthreads = std::min(nthreads, nfiles);
    std::vector<IngestFile> files(nfiles);
    // Plain host buffers: page-locking 25 x 9 MB costs ~75 ms on this host — far more than what
    // staged (pageable) copies lose over 125 MB — and a run ingests its files once.
    auto grow = [&](IngestFile& f, size_t want) -> bool {
        void* p = realloc(f.host, want);
        if (!p) return false;
        f.host = static_cast<uint8_t*>(p);
        f.cap = want;
        return true;
    };
    const uint32_t tile = kh_fasta_tile_bytes();
    auto cleanup_host = [&]() { for (auto& f : files) { free(f.host); f.host = nullptr; } };
    for (int i = 0; i < nfiles; ++i) {
        if (!paths[i]) { cleanup_host(); return kh_fail(KH_E_ARG, "kh_ingest_fasta: path %d is NULL", i); }
        files[i].path = paths[i];
        bool gz;
        const uint64_t hint = inflated_size_hint(paths[i], &gz);
        if (hint) {   // exact-size buffer up front (the workers then never reallocate)
            const size_t want = (size_t)hint + (4u << 20) + 64 + (size_t)((hint + (4u << 20)) / tile + 2) + 4096;
            if (!grow(files[i], want)) { cleanup_host(); return kh_fail(KH_E_NOMEM, "host allocation of %zu bytes failed", want); }
        }
    }
    // workers: files in order; completion is signalled through `ready`
    std::atomic<int> next{0};
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> ready;
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= nfiles) break;
                ingest_read(files[i], grow);
                { std::lock_guard<std::mutex> lk(mu); ready.push_back(i); }
                cv.notify_one();
            }
        });
    kh_seqs* res = new kh_seqs;
    res->bufs.assign(nfiles, nullptr);
    res->lens.assign(nfiles, 0);
    int rc = KH_OK;
    std::string err;
    // device-side lengths, read back once at the end
    DevBuf* d_len = c->buf_alloc(8 * (size_t)nfiles);
    if (!d_len) { rc = KH_E_NOMEM; err = "device allocation failed"; }
    int done = 0;
    while (done < nfiles) {
        int i;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return !ready.empty(); });
            i = ready.back();
            ready.pop_back();
        }
        ++done;
        IngestFile& f = files[i];
        if (rc != KH_OK) continue;
        if (!f.error.empty()) { rc = KH_E_IO; err = f.error; continue; }
        const size_t ship = (size_t)(f.flags_off + f.ntiles + 1);
        DevBuf* d_raw = c->buf_alloc(ship + 16);
        DevBuf* d_out = c->buf_alloc((size_t)f.raw_len + 256);      // 16-byte loads may run past the last base
        DevBuf* d_ws = c->buf_alloc(kh_fasta_clean_workspace(f.raw_len));
        if (!d_raw || !d_out || !d_ws) {
            buf_unref(d_raw); buf_unref(d_out); buf_unref(d_ws);
            rc = KH_E_NOMEM; err = "device allocation failed";
            continue;
        }
        c->prof_begin(KC_COPY_IN);
        hipError_t e = hipMemcpyAsync(d_raw->p, f.host, ship, hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) {
            kh_launch_fasta_clean(static_cast<const u8*>(d_raw->p), f.raw_len, static_cast<const u8*>(d_raw->p) + f.flags_off,
                                  static_cast<u8*>(d_out->p), d_ws->p,
                                  reinterpret_cast<unsigned long long*>(d_len->p) + i, c->st);
            e = hipGetLastError();
        }
        c->prof_end();
        buf_unref(d_raw);      // stream-ordered pool: reusable by whatever is queued after the kernels
        buf_unref(d_ws);
        if (e != hipSuccess) { buf_unref(d_out); rc = KH_E_HIP; err = hipGetErrorString(e); continue; }
        res->bufs[i] = d_out;
    }
    for (auto& th : pool) th.join();
    if (rc == KH_OK) {
        if (hipMemcpyAsync(res->lens.data(), d_len->p, 8 * (size_t)nfiles, hipMemcpyDeviceToHost, c->st) != hipSuccess ||
            hipStreamSynchronize(c->st) != hipSuccess) { rc = KH_E_HIP; err = "read-back of the text lengths failed"; }
    } else {
        (void)hipStreamSynchronize(c->st);
    }
    buf_unref(d_len);
    cleanup_host();
    if (rc != KH_OK) {
        for (auto* b : res->bufs) buf_unref(b);
        delete res;
        return kh_fail(rc, "kh_ingest_fasta: %s", err.c_str());
    }
    *out = res;
    return KH_OK;
}
extern "C" int kh_seqs_count(const kh_seqs* s) { return s ? (int)s->bufs.size() : 0; }
extern "C" int kh_seqs_get(const kh_seqs* s, int i, const uint8_t** dev_ptr, uint64_t* len) {
    if (!s || i < 0 || i >= (int)s->bufs.size()) return kh_fail(KH_E_ARG, "kh_seqs_get: index out of range");
    if (dev_ptr) *dev_ptr = static_cast<const uint8_t*>(s->bufs[i]->p);
    if (len) *len = s->lens[i];
    return KH_OK;
}
extern "C" void kh_seqs_free(kh_seqs* s) {
    if (!s) return;
    for (auto* b : s->bufs) buf_unref(b);
    delete s;
}

// ------------------------------------------------------------------------------ container
// <prefix>.kmc_pre : KhFileHeader
// <prefix>.kmc_suf : "KHAMDSUF" | n*W u64 mixed keys, ascending | n u32 counters (if any)
struct KhFileHeader {
    char magic[8];        // "KHAMDPRE"
    uint32_t version;     // 1
    uint32_t k;
    uint32_t words;       // W
    uint32_t has_counts;
    uint32_t uniform;
    uint32_t counter_max;
    uint64_t n;
    uint64_t mix_id;      // fingerprint of the key mixing function
};
static uint64_t mix_fingerprint() {
    uint64_t a[2] = {0x0123456789abcdefull, 0x1f}, o1[2], o2[2];
    kh_mix_host(31, a, o1);
    kh_mix_host(41, a, o2);
    return o1[0] ^ (o2[0] * 3) ^ (o2[1] * 5);
}

static int write_all(FILE* f, const void* p, size_t n, const char* path) {
    if (n && fwrite(p, 1, n, f) != n) return kh_fail(KH_E_IO, "short write to %s", path);
    return KH_OK;
}
// write through a temporary + rename so that a failed rule never leaves a partial output
struct AtomicFile {
    std::string final_path, tmp_path;
    FILE* f = nullptr;
    int open(const std::string& path) {
        final_path = path;
        tmp_path = path + ".tmp." + std::to_string((long)getpid());
        f = fopen(tmp_path.c_str(), "wb");
        if (!f) return kh_fail(KH_E_IO, "cannot create %s", tmp_path.c_str());
        setvbuf(f, nullptr, _IOFBF, 1 << 22);
        return KH_OK;
    }
    int commit() {
        if (fclose(f) != 0) { f = nullptr; unlink(tmp_path.c_str()); return kh_fail(KH_E_IO, "close failed for %s", tmp_path.c_str()); }
        f = nullptr;
        if (rename(tmp_path.c_str(), final_path.c_str()) != 0) {
            unlink(tmp_path.c_str());
            return kh_fail(KH_E_IO, "rename to %s failed", final_path.c_str());
        }
        return KH_OK;
    }
    ~AtomicFile() {
        if (f) { fclose(f); unlink(tmp_path.c_str()); }
    }
};

extern "C" int kh_save(kh_ctx* c, const kh_set* s, const char* prefix) {
    if (!c || !s || !prefix) return kh_fail(KH_E_ARG, "kh_save: NULL argument");
    if (hipSetDevice(c->dev) != hipSuccess) return kh_fail(KH_E_HIP, "hipSetDevice failed");
    const size_t kb = 8 * (size_t)s->W;
    std::vector<uint8_t> keys(kb * s->n);
    std::vector<uint32_t> counts(s->cb ? s->n : 0);
    if (s->n) {
        if (hipMemcpyAsync(keys.data(), s->keys_ptr(), kb * s->n, hipMemcpyDeviceToHost, c->st) != hipSuccess)
            return kh_fail(KH_E_HIP, "download of keys failed");
        if (s->cb && hipMemcpyAsync(counts.data(), s->counts_ptr(), 4 * s->n, hipMemcpyDeviceToHost, c->st) != hipSuccess)
            return kh_fail(KH_E_HIP, "download of counters failed");
        if (hipStreamSynchronize(c->st) != hipSuccess) return kh_fail(KH_E_HIP, "stream sync failed");
    }
    KhFileHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "KHAMDPRE", 8);
    h.version = 1;
    h.k = (uint32_t)s->k;
    h.words = (uint32_t)s->W;
    h.has_counts = s->cb ? 1 : 0;
    h.uniform = s->uniform;
    h.counter_max = s->counter_max;
    h.n = s->n;
    h.mix_id = mix_fingerprint();
    const std::string pre = std::string(prefix) + ".kmc_pre", suf = std::string(prefix) + ".kmc_suf";
    AtomicFile fs, fp;
    int r;
    if ((r = fs.open(suf)) != KH_OK) return r;
    if ((r = write_all(fs.f, "KHAMDSUF", 8, suf.c_str())) != KH_OK) return r;
    if ((r = write_all(fs.f, keys.data(), keys.size(), suf.c_str())) != KH_OK) return r;
    if ((r = write_all(fs.f, counts.data(), 4 * counts.size(), suf.c_str())) != KH_OK) return r;
    if ((r = fp.open(pre)) != KH_OK) return r;
    if ((r = write_all(fp.f, &h, sizeof h, pre.c_str())) != KH_OK) return r;
    if ((r = fs.commit()) != KH_OK) return r;
    return fp.commit();
}

extern "C" int kh_load(kh_ctx* c, const char* prefix, kh_set** out) {
    if (!c || !prefix || !out) return kh_fail(KH_E_ARG, "kh_load: NULL argument");
    if (hipSetDevice(c->dev) != hipSuccess) return kh_fail(KH_E_HIP, "hipSetDevice failed");
    const std::string pre = std::string(prefix) + ".kmc_pre", suf = std::string(prefix) + ".kmc_suf";
    FILE* fp = fopen(pre.c_str(), "rb");
    if (!fp) return kh_fail(KH_E_IO, "cannot open %s", pre.c_str());
    KhFileHeader h;
    const size_t got = fread(&h, 1, sizeof h, fp);
    fclose(fp);
    if (got != sizeof h || memcmp(h.magic, "KHAMDPRE", 8) != 0 || h.version != 1)
        return kh_fail(KH_E_FORMAT, "%s is not a khoice_amd database (was it written by KMC?)", pre.c_str());
    if (h.k < 1 || h.k > 64 || h.words != (h.k <= 32 ? 1u : 2u))
        return kh_fail(KH_E_FORMAT, "%s: inconsistent header", pre.c_str());
    if (h.mix_id != mix_fingerprint())
        return kh_fail(KH_E_FORMAT, "%s was written with a different key-mixing function", pre.c_str());
    FILE* fs = fopen(suf.c_str(), "rb");
    if (!fs) return kh_fail(KH_E_IO, "cannot open %s", suf.c_str());
    char magic[8];
    const size_t kb = 8 * (size_t)h.words;
    // the header's record count is checked against the file before anything of that size is
    // allocated: a truncated or corrupt pair must be an error return, never a bad_alloc thrown
    // through this extern "C" frame (it would take a resident khoice_server down with it)
    {
        struct stat sb;
        const uint64_t rec = kb + (h.has_counts ? 4 : 0);
        if (fstat(fileno(fs), &sb) != 0 || (uint64_t)sb.st_size < 8 || h.n > ((uint64_t)sb.st_size - 8) / rec ||
            (uint64_t)sb.st_size != 8 + h.n * rec) {
            fclose(fs);
            return kh_fail(KH_E_FORMAT, "%s is truncated or does not match its header (%llu records expected)",
                           suf.c_str(), (unsigned long long)h.n);
        }
    }
    std::vector<uint8_t> keys;
    std::vector<uint32_t> counts;
    try {
        keys.resize(kb * h.n);
        counts.resize(h.has_counts ? h.n : 0);
    } catch (const std::exception&) {
        fclose(fs);
        return kh_fail(KH_E_NOMEM, "%s: not enough host memory for %llu records", suf.c_str(), (unsigned long long)h.n);
    }
    bool ok = fread(magic, 1, 8, fs) == 8 && memcmp(magic, "KHAMDSUF", 8) == 0;
    ok = ok && (keys.empty() || fread(keys.data(), 1, keys.size(), fs) == keys.size());
    ok = ok && (counts.empty() || fread(counts.data(), 4, counts.size(), fs) == counts.size());
    fclose(fs);
    if (!ok) return kh_fail(KH_E_FORMAT, "%s is truncated or not a khoice_amd database", suf.c_str());
    return kh_set_from_mixed_host(c, (int)h.k, h.n, keys.data(), h.has_counts ? counts.data() : nullptr,
                                  h.uniform, h.counter_max ? h.counter_max : KH_KMC_DEFAULT_CS, out);
}

// ------------------------------------------------------------------------------ text outputs
extern "C" int kh_histogram_file(kh_ctx* c, const kh_set* s, uint32_t cmax, const char* path) {
    if (!c || !s || !path || cmax < 1) return kh_fail(KH_E_ARG, "kh_histogram_file: bad argument");
    // one line per counter value: 2^24 lines (a three-byte counter) is the most this writes; a set
    // saturated beyond that would mean a text file of gigabytes nobody reads
    if (cmax > 0xffffffu) return kh_fail(KH_E_ARG, "kh_histogram_file: %u histogram lines requested (limit 16777215)", cmax);
    std::vector<uint64_t> h;
    try { h.resize((size_t)cmax + 1); } catch (const std::exception&) { return kh_fail(KH_E_NOMEM, "histogram of %u lines", cmax); }
    int r = kh_histogram(c, s, h.data(), cmax + 1);
    if (r != KH_OK) return r;
    AtomicFile f;
    if ((r = f.open(path)) != KH_OK) return r;
    for (uint32_t i = 1; i <= cmax; ++i) fprintf(f.f, "%u\t%llu\n", i, (unsigned long long)h[i]);
    return f.commit();
}

extern "C" int kh_dump_sorted(kh_ctx* c, const kh_set* s, const char* path) {
    if (!c || !s || !path) return kh_fail(KH_E_ARG, "kh_dump_sorted: NULL argument");
    const int W = s->W, k = s->k;
    std::vector<uint64_t> keys((size_t)s->n * W);
    std::vector<uint32_t> counts(s->n);
    int r = kh_set_download(c, s, keys.data(), counts.data());
    if (r != KH_OK) return r;
    std::vector<uint64_t> idx(s->n);
    for (uint64_t i = 0; i < s->n; ++i) idx[i] = i;
    if (W == 1)
        std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return keys[a] < keys[b]; });
    else
        std::sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) {
            return keys[2 * a + 1] < keys[2 * b + 1] ||
                   (keys[2 * a + 1] == keys[2 * b + 1] && keys[2 * a] < keys[2 * b]);
        });
    AtomicFile f;
    if ((r = f.open(path)) != KH_OK) return r;
    char line[128];
    for (uint64_t ii = 0; ii < s->n; ++ii) {
        const uint64_t i = idx[ii];
        for (int b = 0; b < k; ++b) {
            const int bit = 2 * (k - 1 - b);
            const uint64_t w = keys[i * W + (bit >> 6)];
            line[b] = "ACGT"[(w >> (bit & 63)) & 3];
        }
        const int m = snprintf(line + k, sizeof line - k, "\t%u\n", counts[i]);
        if (fwrite(line, 1, (size_t)k + m, f.f) != (size_t)k + m) return kh_fail(KH_E_IO, "short write to %s", path);
    }
    return f.commit();
}

// text form of a histogram held in host memory (the fused kh_exp1_run hands back arrays, the
// Snakemake rules declare step_4 / step_8 files): lines "c<TAB>n" for c = 1..cmax, counters past
// the array read 0
extern "C" int kh_write_histogram_text(const char* path, const uint64_t* hist, uint32_t hist_len, uint32_t cmax) {
    if (!path || !hist || cmax < 1) return kh_fail(KH_E_ARG, "kh_write_histogram_text: bad argument");
    // (the same limit as kh_histogram_file: the text of a three-byte counter's histogram is the longest this writes;
    // an unbounded cmax would be a bad_alloc thrown through this extern "C" frame)
    if (cmax > 0xffffffu) return kh_fail(KH_E_ARG, "kh_write_histogram_text: %u histogram lines requested (limit 16777215)", cmax);
    AtomicFile f;
    int r;
    if ((r = f.open(path)) != KH_OK) return r;
    // 65535 lines per file and six files per k: formatted by hand into one buffer (fprintf cost ~3 ms a file)
    std::vector<char> buf;
    try { buf.reserve((size_t)cmax * 14 + 64); } catch (const std::exception&) { return kh_fail(KH_E_NOMEM, "histogram text of %u lines", cmax); }
    char tmp[24];
    auto put = [&](unsigned long long v) {
        int m = 0;
        do { tmp[m++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (m) buf.push_back(tmp[--m]);
    };
    for (uint32_t i = 1; i <= cmax; ++i) {
        put(i);
        buf.push_back('\t');
        put(i < hist_len ? hist[i] : 0ull);
        buf.push_back('\n');
    }
    if ((r = write_all(f.f, buf.data(), buf.size(), path)) != KH_OK) return r;
    return f.commit();
}

