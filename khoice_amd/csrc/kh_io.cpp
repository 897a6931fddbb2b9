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
