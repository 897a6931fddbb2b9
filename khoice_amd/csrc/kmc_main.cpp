// bin/kmc — drop-in for KMC 3's `kmc` at khoice's call sites, e.g.
//   kmc -fm -m64 -k{k} -ci1 IN.fna.gz OUT_PREFIX tmp/        (workflow/rules/exp_type_1.smk:163)
// A thin argv parser over libkhoice_hip.so (include/khoice_hip.h); all counting runs on the GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "khoice_hip.h"

static int usage(const char* why) {
    fprintf(stderr,
            "%s%skmc (khoice_amd, MI355X) — canonical k-mer counter\n"
            "Usage: kmc [options] <input_file|@file_list> <output_prefix> <working_directory>\n"
            "  -k<len>   k-mer length, 1..64 (default 25)\n"
            "  -ci<v>    exclude k-mers occurring fewer than v times (default 2)\n"
            "  -cx<v>    exclude k-mers occurring more than v times (default 1e9)\n"
            "  -cs<v>    counter saturation value (default 255)\n"
            "  -fm|-fa   (multi-)FASTA input, optionally gzip'ed (default)\n"
            "  -m<GB> -t<n> -sm -p<n> -r -v -hp -w -e -n<n>   accepted and ignored\n"
            "  -b (non-canonical counting), -fq, -fbam, -fkmc are not supported by this build\n",
            why ? "kmc: " : "", why ? why : "");
    if (why) fputc('\n', stderr);
    return 1;
}

static bool parse_u64(const char* s, unsigned long long* out) {
    if (!*s) return false;
    char* end = nullptr;
    const double d = strtod(s, &end);   // KMC accepts forms like 1e9
    if (*end || d < 0) return false;
    *out = (unsigned long long)d;
    return true;
}

int main(int argc, char** argv) {
    int k = 25;
    unsigned long long ci = 2, cx = 1000000000ull, cs = 255;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (a[0] != '-' || !a[1]) { pos.push_back(a); continue; }
        unsigned long long v = 0;
        if (!strncmp(a, "-ci", 3)) { if (!parse_u64(a + 3, &ci)) return usage("bad -ci value"); }
        else if (!strncmp(a, "-cx", 3)) { if (!parse_u64(a + 3, &cx)) return usage("bad -cx value"); }
        else if (!strncmp(a, "-cs", 3)) { if (!parse_u64(a + 3, &cs)) return usage("bad -cs value"); }
        else if (a[1] == 'k') { if (!parse_u64(a + 2, &v)) return usage("bad -k value"); k = (int)v; }
        else if (a[1] == 'f') {
            const std::string f = a + 2;
            if (f != "m" && f != "a") return usage("only -fm / -fa input is supported by this build");
        }
        else if (!strcmp(a, "-b")) return usage("-b (non-canonical counting) is not supported by this build");
        else if (a[1] == 'm' || a[1] == 't' || a[1] == 'p' || a[1] == 'n' || a[1] == 'r' || a[1] == 'v' ||
                 a[1] == 'w' || a[1] == 'e' || !strcmp(a, "-sm") || !strcmp(a, "-hp") || a[1] == 'j' ||
                 a[1] == 'o') { /* performance / reporting knobs: no semantic effect */ }
        else return usage((std::string("unknown option ") + a).c_str());
    }
    if (pos.size() != 3) return usage("expected <input> <output_prefix> <working_directory>");
    if (k < 1 || k > 64) return usage("k must be in 1..64 for this build");
    if (ci < 1) ci = 1;
    const uint32_t cx32 = cx >= 0xffffffffull ? KH_NO_MAX : (uint32_t)cx;
    const uint32_t cs32 = cs >= 0xffffffffull ? 0xfffffffeu : (uint32_t)cs;
    if (cs32 < 1) return usage("-cs must be >= 1");

    std::vector<std::string> inputs;
    if (pos[0][0] == '@') {
        std::ifstream fl(pos[0].substr(1));
        if (!fl) { fprintf(stderr, "kmc: cannot open file list %s\n", pos[0].c_str() + 1); return 1; }
        std::string line;
        while (std::getline(fl, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (!line.empty()) inputs.push_back(line);
        }
    } else {
        inputs.push_back(pos[0]);
    }
    if (inputs.empty()) { fprintf(stderr, "kmc: no input files\n"); return 1; }

    // host ingest: all files form ONE database (records never joined: '\n' between files)
    std::vector<uint8_t> text;
    for (const auto& path : inputs) {
        uint8_t* seq = nullptr;
        uint64_t len = 0;
        if (kh_read_fasta(path.c_str(), &seq, &len) != KH_OK) {
            fprintf(stderr, "kmc: %s\n", kh_last_error());
            return 1;
        }
        if (!text.empty()) text.push_back('\n');
        text.insert(text.end(), seq, seq + len);
        kh_free_host(seq);
    }

    const char* dev_env = getenv("KHOICE_GPU_DEVICE");
    kh_ctx* ctx = nullptr;
    if (kh_ctx_create(dev_env ? atoi(dev_env) : 0, &ctx) != KH_OK) {
        fprintf(stderr, "kmc: %s\n", kh_last_error());
        return 1;
    }
    const uint8_t* seqs[1] = {text.data()};
    const uint64_t lens[1] = {text.size()};
    kh_set* set = nullptr;
    int rc = kh_build_batch(ctx, 1, seqs, lens, 0, k, (uint32_t)ci, cx32, cs32, 1, &set);
    if (rc == KH_OK) rc = kh_save(ctx, set, pos[1].c_str());
    if (rc != KH_OK) {
        fprintf(stderr, "kmc: %s\n", kh_last_error());
        kh_set_free(set);
        kh_ctx_destroy(ctx);
        return 1;
    }
    uint64_t n = 0;
    kh_set_info(set, &n, nullptr, nullptr, nullptr, nullptr);
    char stats[8192];
    unsigned long long kmers = 0;
    if (kh_stats(ctx, stats, sizeof stats) == KH_OK) {
        const char* p = strstr(stats, "\"kmers\":");
        if (p) kmers = strtoull(p + 8, nullptr, 10);
    }
    printf("Stats (khoice_amd kmc on MI355X):\n");
    printf("   No. of unique counted k-mers       : %12llu\n", (unsigned long long)n);
    printf("   Total no. of k-mers                : %12llu\n", kmers);
    printf("   Total no. of bases                 : %12llu\n", (unsigned long long)text.size());
    kh_set_free(set);
    kh_ctx_destroy(ctx);
    return 0;
}
