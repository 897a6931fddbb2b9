// khoice_amd — the `kmc` and `kmc_tools` front ends (argv / ops-file parsing over the C ABI).
// Reference call sites: workflow/rules/exp_type_1.smk:163,173,182,191; exp_type_2.smk:363-379;
// exp_type_4.smk:255-257.  Every operation runs on the GPU through libkhoice_hip.so.
#include "kh_cli.h"

#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------ kmc
static int kmc_usage(std::string& err, const char* why) {
    char buf[2048];
    snprintf(buf, sizeof buf,
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
    err += buf;
    if (why) err += "\n";
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

int kh_cli_kmc(kh_ctx* ctx, const std::vector<std::string>& args, std::string& out, std::string& err) {
    int k = 25;
    unsigned long long ci = 2, cx = 1000000000ull, cs = 255;
    std::vector<std::string> pos;
    for (size_t i = 0; i < args.size(); ++i) {
        const char* a = args[i].c_str();
        if (a[0] != '-' || !a[1]) { pos.push_back(a); continue; }
        unsigned long long v = 0;
        if (!strncmp(a, "-ci", 3)) { if (!parse_u64(a + 3, &ci)) return kmc_usage(err, "bad -ci value"); }
        else if (!strncmp(a, "-cx", 3)) { if (!parse_u64(a + 3, &cx)) return kmc_usage(err, "bad -cx value"); }
        else if (!strncmp(a, "-cs", 3)) { if (!parse_u64(a + 3, &cs)) return kmc_usage(err, "bad -cs value"); }
        else if (a[1] == 'k') { if (!parse_u64(a + 2, &v)) return kmc_usage(err, "bad -k value"); k = (int)v; }
        else if (a[1] == 'f') {
            const std::string f = a + 2;
            if (f != "m" && f != "a") return kmc_usage(err, "only -fm / -fa input is supported by this build");
        }
        else if (!strcmp(a, "-b")) return kmc_usage(err, "-b (non-canonical counting) is not supported by this build");
        else if (a[1] == 'm' || a[1] == 't' || a[1] == 'p' || a[1] == 'n' || a[1] == 'r' || a[1] == 'v' ||
                 a[1] == 'w' || a[1] == 'e' || !strcmp(a, "-sm") || !strcmp(a, "-hp") || a[1] == 'j' ||
                 a[1] == 'o') { /* performance / reporting knobs: no semantic effect */ }
        else return kmc_usage(err, (std::string("unknown option ") + a).c_str());
    }
    if (pos.size() != 3) return kmc_usage(err, "expected <input> <output_prefix> <working_directory>");
    if (k < 1 || k > 64) return kmc_usage(err, "k must be in 1..64 for this build");
    if (ci < 1) ci = 1;
    // KMC's own default (-cx1e9) means "no upper cut-off" for every counter this engine can hold;
    // passing it through as a number made long inputs (chunked builds) refuse `-ci1` runs
    const uint32_t cx32 = cx >= 1000000000ull ? KH_NO_MAX : (uint32_t)cx;
    const uint32_t cs32 = cs >= 0xffffffffull ? 0xfffffffeu : (uint32_t)cs;
    if (cs32 < 1) return kmc_usage(err, "-cs must be >= 1");

    std::vector<std::string> inputs;
    if (pos[0][0] == '@') {
        std::ifstream fl(pos[0].substr(1));
        if (!fl) { err += "kmc: cannot open file list " + pos[0].substr(1) + "\n"; return 1; }
        std::string line;
        while (std::getline(fl, line)) {
            while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
            if (!line.empty()) inputs.push_back(line);
        }
    } else {
        inputs.push_back(pos[0]);
    }
    if (inputs.empty()) { err += "kmc: no input files\n"; return 1; }

    // host ingest: all files form ONE database (records never joined: '\n' between files)
    std::vector<uint8_t> text;
    for (const auto& path : inputs) {
        uint8_t* seq = nullptr;
        uint64_t len = 0;
        if (kh_read_fasta(path.c_str(), &seq, &len) != KH_OK) {
            err += std::string("kmc: ") + kh_last_error() + "\n";
            return 1;
        }
        if (!text.empty()) text.push_back('\n');
        text.insert(text.end(), seq, seq + len);
        kh_free_host(seq);
    }

    const uint8_t* seqs[1] = {text.data()};
    const uint64_t lens[1] = {text.size()};
    kh_set* set = nullptr;
    int rc = kh_build_batch(ctx, 1, seqs, lens, 0, k, (uint32_t)ci, cx32, cs32, 1, &set);
    if (rc == KH_OK) rc = kh_save(ctx, set, pos[1].c_str());
    if (rc != KH_OK) {
        err += std::string("kmc: ") + kh_last_error() + "\n";
        kh_set_free(set);
        return 1;
    }
    uint64_t n = 0;
    kh_set_info(set, &n, nullptr, nullptr, nullptr, nullptr);
    char line[256];
    out += "Stats (khoice_amd kmc on MI355X):\n";
    snprintf(line, sizeof line, "   No. of unique counted k-mers       : %12llu\n", (unsigned long long)n);
    out += line;
    snprintf(line, sizeof line, "   Total no. of bases                 : %12llu\n", (unsigned long long)text.size());
    out += line;
    kh_set_free(set);
    return 0;
}

// ------------------------------------------------------------------------------ kmc_tools
namespace {

struct Fail { std::string msg; };
[[noreturn]] void die(const std::string& m) { throw Fail{m}; }
void chk(int rc) { if (rc != KH_OK) die(kh_last_error()); }

struct Set {   // owning handle
    kh_set* h = nullptr;
    Set() = default;
    explicit Set(kh_set* p) : h(p) {}
    Set(const Set&) = delete;
    Set& operator=(const Set&) = delete;
    Set(Set&& o) noexcept : h(o.h) { o.h = nullptr; }
    Set& operator=(Set&& o) noexcept { if (this != &o) { kh_set_free(h); h = o.h; o.h = nullptr; } return *this; }
    ~Set() { kh_set_free(h); }
};

thread_local kh_ctx* g_ctx = nullptr;

Set load(const std::string& prefix) { kh_set* s = nullptr; chk(kh_load(g_ctx, prefix.c_str(), &s)); return Set(s); }

bool starts(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
uint32_t to_u32(const std::string& s, const char* what) {
    char* end = nullptr;
    const double d = strtod(s.c_str(), &end);
    if (s.empty() || *end || d < 0) die(std::string("bad value for ") + what + ": " + s);
    return d >= 4294967295.0 ? 0xffffffffu : (uint32_t)d;
}
// number of histogram lines KMC prints for a database: 2^(8 * counter bytes) - 1
uint32_t hist_lines(const kh_set* s) {
    uint32_t cm = 255;
    kh_set_counter_max(s, &cm);
    if (cm <= 0xffu) return 0xffu;
    if (cm <= 0xffffu) return 0xffffu;
    return 0xffffffu;   // counters saturated above three bytes: the histogram stops at 2^24 - 1 lines
}

struct Cut { uint32_t ci = 1, cx = KH_NO_MAX, cs = 0; int mode = -1; };
// consume -ci/-cx/-cs/-oc options starting at argv[i]
Cut take_opts(const std::vector<std::string>& a, size_t& i) {
    Cut c;
    while (i < a.size() && a[i].size() > 1 && a[i][0] == '-') {
        const std::string& o = a[i];
        if (starts(o, "-ci")) c.ci = to_u32(o.substr(3), "-ci");
        else if (starts(o, "-cx")) c.cx = to_u32(o.substr(3), "-cx");
        else if (starts(o, "-cs")) c.cs = to_u32(o.substr(3), "-cs");
        else if (starts(o, "-oc")) {
            static const std::map<std::string, int> m{{"min", KH_MODE_MIN}, {"max", KH_MODE_MAX}, {"sum", KH_MODE_SUM},
                                                      {"diff", KH_MODE_DIFF}, {"left", KH_MODE_LEFT}, {"right", KH_MODE_RIGHT}};
            auto it = m.find(o.substr(3));
            if (it == m.end()) die("unknown counter mode " + o);
            c.mode = it->second;
        } else break;
        ++i;
    }
    return c;
}
void no_cutoffs(const Cut& c, const char* where) {
    if (c.ci > 1 || c.cx != KH_NO_MAX)
        die(std::string("-ci/-cx on ") + where + " are not supported by this build");
}

// ---------------------------------------------------------------- transform
int do_transform(const std::vector<std::string>& a) {
    size_t i = 0;
    if (i >= a.size()) die("transform: missing input");
    const std::string in = a[i++];
    no_cutoffs(take_opts(a, i), "transform input");
    Set src = load(in);
    if (i >= a.size()) die("transform: missing operation");
    while (i < a.size()) {
        const std::string op = a[i++];
        if (op == "set_counts") {
            if (i + 1 >= a.size()) die("set_counts needs <value> <output>");
            const uint32_t v = to_u32(a[i++], "set_counts");
            const std::string out = a[i++];
            no_cutoffs(take_opts(a, i), "set_counts output");
            kh_set* r = nullptr;
            chk(kh_set_counts(g_ctx, src.h, v, &r));
            Set rs(r);
            chk(kh_save(g_ctx, rs.h, out.c_str()));
        } else if (op == "histogram") {
            if (i >= a.size()) die("histogram needs <output file>");
            const std::string out = a[i++];
            const Cut c = take_opts(a, i);
            if (c.ci > 1) die("histogram -ci is not supported by this build");
            const uint32_t lines = c.cx != KH_NO_MAX ? c.cx : hist_lines(src.h);
            chk(kh_histogram_file(g_ctx, src.h, lines, out.c_str()));
        } else if (op == "dump") {
            if (i < a.size() && a[i] == "-s") ++i;   // our dump is always sorted
            if (i >= a.size()) die("dump needs <output file>");
            const std::string out = a[i++];
            no_cutoffs(take_opts(a, i), "dump output");
            chk(kh_dump_sorted(g_ctx, src.h, out.c_str()));
        } else if (op == "sort" || op == "compact" || op == "reduce") {
            if (i >= a.size()) die(op + " needs <output>");
            const std::string out = a[i++];
            no_cutoffs(take_opts(a, i), "output");
            chk(kh_save(g_ctx, src.h, out.c_str()));   // databases are always sorted and compact
        } else {
            die("transform: unknown operation " + op);
        }
    }
    return 0;
}

// ---------------------------------------------------------------- simple
int do_simple(const std::vector<std::string>& a) {
    size_t i = 0;
    if (a.size() < 4) die("simple: expected <input1> <input2> <oper> <output> ...");
    const std::string in1 = a[i++];
    no_cutoffs(take_opts(a, i), "simple input");
    if (i >= a.size()) die("simple: missing second input");
    const std::string in2 = a[i++];
    no_cutoffs(take_opts(a, i), "simple input");
    Set A = load(in1), B = load(in2);
    if (i >= a.size()) die("simple: missing operation");
    while (i < a.size()) {
        const std::string op = a[i++];
        if (i >= a.size()) die("simple: operation " + op + " needs an output");
        const std::string out = a[i++];
        const Cut c = take_opts(a, i);
        no_cutoffs(c, "simple output");
        // no -cs on the output: the operands' larger counter range (PARITY UNPINNED — no KMC binary or
        // fixture to check against; the reference's `simple ... intersect OUT -ocsum` calls,
        // exp_type_2.smk:363-365, run on a -cs5000 group union and would saturate at 255 otherwise)
        uint32_t cma = KH_KMC_DEFAULT_CS, cmb = KH_KMC_DEFAULT_CS;
        kh_set_counter_max(A.h, &cma);
        kh_set_counter_max(B.h, &cmb);
        const uint32_t cs = c.cs ? c.cs : std::max(cma, cmb);
        int code, mode;
        const kh_set *x = A.h, *y = B.h;
        if (op == "intersect") { code = KH_INTERSECT; mode = KH_MODE_MIN; }
        else if (op == "union") { code = KH_UNION; mode = KH_MODE_SUM; }
        else if (op == "kmers_subtract") { code = KH_KMERS_SUBTRACT; mode = KH_MODE_LEFT; }
        else if (op == "counters_subtract") { code = KH_COUNTERS_SUBTRACT; mode = KH_MODE_DIFF; }
        else if (op == "reverse_kmers_subtract") { code = KH_KMERS_SUBTRACT; mode = KH_MODE_LEFT; std::swap(x, y); }
        else if (op == "reverse_counters_subtract") { code = KH_COUNTERS_SUBTRACT; mode = KH_MODE_DIFF; std::swap(x, y); }
        else die("simple: unknown operation " + op);
        if (c.mode >= 0) {
            if (code != KH_INTERSECT && code != KH_UNION) die("-oc applies to intersect and union only");
            mode = c.mode;
        }
        kh_set* r = nullptr;
        chk(kh_simple(g_ctx, x, y, code, mode, cs, &r));
        Set rs(r);
        chk(kh_save(g_ctx, rs.h, out.c_str()));
    }
    return 0;
}

// ---------------------------------------------------------------- complex
// ops file (written by exp_type_1.smk:52-61):
//   INPUT:
//   set1 = <prefix> [-ci.. -cx..]
//   OUTPUT:
//   <out_prefix> = (set1 + set2 + ... )          operators: + union(sum)  * intersect(min)
//   OUTPUT_PARAMS:                                           - kmers_subtract  ~ counters_subtract
//   -cs5000
struct Expr {
    char op = 0;   // 0 = leaf
    std::string name;
    std::vector<std::unique_ptr<Expr>> kids;
};
struct Parser {
    std::vector<std::string> tok;
    size_t at = 0;
    explicit Parser(const std::string& s) {
        std::string cur;
        for (char ch : s) {
            if (isspace((unsigned char)ch) || strchr("()+-*~", ch)) {
                if (!cur.empty()) { tok.push_back(cur); cur.clear(); }
                if (!isspace((unsigned char)ch)) tok.emplace_back(1, ch);
            } else cur.push_back(ch);
        }
        if (!cur.empty()) tok.push_back(cur);
    }
    bool peek(const char* t) const { return at < tok.size() && tok[at] == t; }
    std::unique_ptr<Expr> factor() {
        if (at >= tok.size()) die("complex: unexpected end of expression");
        if (peek("(")) { ++at; auto e = expr(); if (!peek(")")) die("complex: missing ')'"); ++at; return e; }
        auto e = std::make_unique<Expr>();
        e->name = tok[at++];
        return e;
    }
    std::unique_ptr<Expr> term() {
        auto l = factor();
        while (peek("*")) { ++at; auto n = std::make_unique<Expr>(); n->op = '*'; n->kids.push_back(std::move(l)); n->kids.push_back(factor()); l = std::move(n); }
        return l;
    }
    // KMC binds '*' tightest, then '-' and '~', then '+' (khoice only ever writes '+')
    std::unique_ptr<Expr> diff() {
        auto l = term();
        while (peek("-") || peek("~")) {
            const char op = tok[at++][0];
            auto n = std::make_unique<Expr>();
            n->op = op;
            n->kids.push_back(std::move(l));
            n->kids.push_back(term());
            l = std::move(n);
        }
        return l;
    }
    std::unique_ptr<Expr> expr() {
        auto l = diff();
        while (peek("+")) {
            ++at;
            auto r = diff();
            if (l->op == '+') { l->kids.push_back(std::move(r)); continue; }   // n-ary union
            auto n = std::make_unique<Expr>();
            n->op = '+';
            n->kids.push_back(std::move(l));
            n->kids.push_back(std::move(r));
            l = std::move(n);
        }
        return l;
    }
};

const uint32_t kNoSat = 0x7fffffffu;   // intermediate results never saturate

Set eval(const Expr& e, std::map<std::string, Set>& inputs, uint32_t cs, bool top) {
    if (!e.op) {
        auto it = inputs.find(e.name);
        if (it == inputs.end()) die("complex: undefined input " + e.name);
        // a bare input as the whole expression: union of one set (counters saturate at cs)
        const kh_set* one[1] = {it->second.h};
        kh_set* r = nullptr;
        chk(kh_union_sum(g_ctx, one, 1, top ? cs : kNoSat, &r, nullptr, 0));
        return Set(r);
    }
    const uint32_t sat = top ? cs : kNoSat;
    if (e.op == '+') {
        std::vector<Set> tmp;
        std::vector<const kh_set*> ops;
        for (auto& kid : e.kids) {
            if (!kid->op) {
                auto it = inputs.find(kid->name);
                if (it == inputs.end()) die("complex: undefined input " + kid->name);
                ops.push_back(it->second.h);
            } else {
                tmp.push_back(eval(*kid, inputs, cs, false));
                ops.push_back(tmp.back().h);
            }
        }
        kh_set* r = nullptr;
        chk(kh_union_sum(g_ctx, ops.data(), (int)ops.size(), sat, &r, nullptr, 0));
        return Set(r);
    }
    Set l = eval(*e.kids[0], inputs, cs, false), r = eval(*e.kids[1], inputs, cs, false);
    int code = KH_INTERSECT, mode = KH_MODE_MIN;
    if (e.op == '-') { code = KH_KMERS_SUBTRACT; mode = KH_MODE_LEFT; }
    if (e.op == '~') { code = KH_COUNTERS_SUBTRACT; mode = KH_MODE_DIFF; }
    kh_set* out = nullptr;
    chk(kh_simple(g_ctx, l.h, r.h, code, mode, sat, &out));
    return Set(out);
}

int do_complex(const std::vector<std::string>& a) {
    if (a.size() != 1) die("complex: expected <operations_definition_file>");
    std::ifstream f(a[0]);
    if (!f) die("cannot open " + a[0]);
    std::map<std::string, Set> inputs;
    std::string section, line, out_prefix, expr_text;
    uint32_t cs = 0;   // 0: no -cs given -> the largest counter range among the inputs (see do_simple)
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || isspace((unsigned char)line.back()))) line.pop_back();
        size_t b = 0;
        while (b < line.size() && isspace((unsigned char)line[b])) ++b;
        line = line.substr(b);
        if (line.empty()) continue;
        if (line == "INPUT:" || line == "OUTPUT:" || line == "OUTPUT_PARAMS:") { section = line; continue; }
        if (section == "INPUT:") {
            const size_t eq = line.find('=');
            if (eq == std::string::npos) die("complex: bad INPUT line: " + line);
            std::istringstream ls(line.substr(eq + 1));
            std::string name = line.substr(0, eq), path, opt;
            while (!name.empty() && isspace((unsigned char)name.back())) name.pop_back();
            ls >> path;
            std::vector<std::string> opts;
            while (ls >> opt) opts.push_back(opt);
            size_t oi = 0;
            no_cutoffs(take_opts(opts, oi), "complex input");
            inputs.emplace(name, load(path));
        } else if (section == "OUTPUT:") {
            const size_t eq = line.find('=');
            if (eq == std::string::npos) die("complex: bad OUTPUT line: " + line);
            out_prefix = line.substr(0, eq);
            while (!out_prefix.empty() && isspace((unsigned char)out_prefix.back())) out_prefix.pop_back();
            expr_text = line.substr(eq + 1);
        } else if (section == "OUTPUT_PARAMS:") {
            std::istringstream ls(line);
            std::vector<std::string> opts;
            std::string o;
            while (ls >> o) opts.push_back(o);
            size_t oi = 0;
            const Cut c = take_opts(opts, oi);
            no_cutoffs(c, "complex output");
            if (c.cs) cs = c.cs;
        } else {
            die("complex: text outside INPUT:/OUTPUT:/OUTPUT_PARAMS: sections");
        }
    }
    if (out_prefix.empty()) die("complex: no OUTPUT: line");
    if (!cs) {
        cs = KH_KMC_DEFAULT_CS;
        for (auto& kv : inputs) {
            uint32_t cm = KH_KMC_DEFAULT_CS;
            kh_set_counter_max(kv.second.h, &cm);
            cs = std::max(cs, cm);
        }
    }
    Parser p(expr_text);
    auto e = p.expr();
    if (p.at != p.tok.size()) die("complex: trailing text in expression");
    Set r = eval(*e, inputs, cs, true);
    chk(kh_save(g_ctx, r.h, out_prefix.c_str()));
    return 0;
}

int tools_usage(std::string& err) {
    err +=
            "kmc_tools (khoice_amd, MI355X)\n"
            "Usage: kmc_tools [-t<n>] [-v] [-hp] <mode> <mode params>\n"
            "  transform <input> <oper> [params] <output> ...   oper: set_counts <v> | histogram | dump [-s] | sort | compact\n"
            "  simple <input1> <input2> <oper> <output> [-cs<v>] [-oc<min|max|sum|diff|left|right>] ...\n"
            "         oper: intersect | union | kmers_subtract | counters_subtract | reverse_*\n"
            "  complex <operations_definition_file>\n";
    return 1;
}

}   // namespace

int kh_cli_kmc_tools(kh_ctx* ctx, const std::vector<std::string>& args, std::string& out, std::string& err) {
    (void)out;
    size_t i = 0;
    for (; i < args.size(); ++i) {   // global options
        const std::string& a = args[i];
        if (a.size() > 1 && a[0] == '-' && (a[1] == 't' || a[1] == 'v' || a == "-hp")) continue;
        break;
    }
    if (i >= args.size()) return tools_usage(err);
    const std::string mode = args[i++];
    std::vector<std::string> a(args.begin() + i, args.end());
    g_ctx = ctx;
    int rc = 1;
    try {
        if (mode == "transform") rc = do_transform(a);
        else if (mode == "simple") rc = do_simple(a);
        else if (mode == "complex") rc = do_complex(a);
        else rc = tools_usage(err);
    } catch (const Fail& f) {
        err += "kmc_tools: " + f.msg + "\n";
        rc = 1;
    } catch (const std::exception& e) {   // bad_alloc and friends must not take a resident server down
        err += std::string("kmc_tools: ") + e.what() + "\n";
        rc = 1;
    }
    g_ctx = nullptr;
    return rc;
}

// ------------------------------------------------------------------------------ server client
// wire format (all integers little-endian u32): request = tool, cwd, argc, args...; each string
// as length + bytes.  reply = status (i32), stdout text, stderr text.
static bool send_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t w = ::write(fd, c, n);
        if (w <= 0) return false;
        c += w;
        n -= (size_t)w;
    }
    return true;
}
static bool recv_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t r = ::read(fd, c, n);
        if (r <= 0) return false;
        c += r;
        n -= (size_t)r;
    }
    return true;
}
bool kh_wire_send_str(int fd, const std::string& s) {
    const uint32_t n = (uint32_t)s.size();
    return send_all(fd, &n, 4) && send_all(fd, s.data(), n);
}
bool kh_wire_recv_str(int fd, std::string& s) {
    uint32_t n = 0;
    if (!recv_all(fd, &n, 4) || n > (64u << 20)) return false;
    s.resize(n);
    return n == 0 || recv_all(fd, &s[0], n);
}
bool kh_wire_send_u32(int fd, uint32_t v) { return send_all(fd, &v, 4); }
bool kh_wire_recv_u32(int fd, uint32_t& v) { return recv_all(fd, &v, 4); }

bool kh_cli_try_server(const char* tool, const std::vector<std::string>& args, int* status) {
    const char* path = getenv("KHOICE_SERVER");
    if (!path || !*path) return false;
    const int fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return false;
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    strncpy(addr.sun_path, path, sizeof addr.sun_path - 1);
    if (::connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) != 0) { ::close(fd); return false; }
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) { ::close(fd); return false; }
    bool ok = kh_wire_send_str(fd, tool) && kh_wire_send_str(fd, cwd) && kh_wire_send_u32(fd, (uint32_t)args.size());
    for (size_t i = 0; ok && i < args.size(); ++i) ok = kh_wire_send_str(fd, args[i]);
    uint32_t st = 1;
    std::string out, err;
    ok = ok && kh_wire_recv_u32(fd, st) && kh_wire_recv_str(fd, out) && kh_wire_recv_str(fd, err);
    ::close(fd);
    if (!ok) {
        fprintf(stderr, "%s: lost the connection to the khoice_amd server at %s\n", tool, path);
        *status = 1;
        return true;
    }
    fputs(out.c_str(), stdout);
    fputs(err.c_str(), stderr);
    *status = (int)st;
    return true;
}
