// bin/khoice_server — resident engine behind bin/kmc and bin/kmc_tools (SURVEY §8f "next" #1).
// khoice's DAG is thousands of short `kmc` / `kmc_tools` processes (exp_type_1.smk:156-259);
// each would pay a HIP initialisation.  With this server running and $KHOICE_SERVER pointing at
// its socket, those processes become thin clients and every operation shares ONE context, its
// device memory pool and its loaded code objects.  Requests are served one at a time (the GPU
// serialises them anyway); relative paths are resolved against the client's working directory.
//   khoice_server <socket_path> [device]        stop with SIGTERM or `khoice_server --stop <socket>`
#include <signal.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kh_cli.h"

bool kh_wire_send_str(int fd, const std::string& s);
bool kh_wire_recv_str(int fd, std::string& s);
bool kh_wire_send_u32(int fd, uint32_t v);
bool kh_wire_recv_u32(int fd, uint32_t& v);

static volatile sig_atomic_t g_stop = 0;
static void on_term(int) { g_stop = 1; }

int main(int argc, char** argv) {
    if (argc == 3 && !strcmp(argv[1], "--stop")) {
        setenv("KHOICE_SERVER", argv[2], 1);
        int st = 1;
        if (!kh_cli_try_server("shutdown", {}, &st)) { fprintf(stderr, "no server at %s\n", argv[2]); return 1; }
        return st;
    }
    if (argc < 2) { fprintf(stderr, "usage: khoice_server <socket_path> [device] | --stop <socket_path>\n"); return 1; }
    const char* path = argv[1];
    kh_ctx* ctx = nullptr;
    if (kh_ctx_create(argc > 2 ? atoi(argv[2]) : 0, &ctx) != KH_OK) {
        fprintf(stderr, "khoice_server: %s\n", kh_last_error());
        return 1;
    }
    const int srv = ::socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    strncpy(addr.sun_path, path, sizeof addr.sun_path - 1);
    unlink(path);
    if (srv < 0 || ::bind(srv, reinterpret_cast<sockaddr*>(&addr), sizeof addr) != 0 || ::listen(srv, 64) != 0) {
        fprintf(stderr, "khoice_server: cannot listen on %s\n", path);
        kh_ctx_destroy(ctx);
        return 1;
    }
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_term;
    sigaction(SIGTERM, &sa, nullptr);
    sigaction(SIGINT, &sa, nullptr);
    signal(SIGPIPE, SIG_IGN);
    char home[4096];
    if (!getcwd(home, sizeof home)) strcpy(home, "/");
    fprintf(stderr, "khoice_server: ready on %s\n", path);
    unsigned long served = 0;
    while (!g_stop) {
        const int fd = ::accept(srv, nullptr, nullptr);
        if (fd < 0) continue;
        std::string tool, cwd;
        uint32_t n = 0;
        bool ok = kh_wire_recv_str(fd, tool) && kh_wire_recv_str(fd, cwd) && kh_wire_recv_u32(fd, n) && n < 4096;
        std::vector<std::string> args(ok ? n : 0);
        for (uint32_t i = 0; ok && i < n; ++i) ok = kh_wire_recv_str(fd, args[i]);
        int status = 1;
        std::string out, err;
        if (ok) {
            if (tool == "shutdown") { status = 0; g_stop = 1; }
            else if (chdir(cwd.c_str()) != 0) err = "khoice_server: cannot enter " + cwd + "\n";
            else if (tool == "kmc") status = kh_cli_kmc(ctx, args, out, err);
            else if (tool == "kmc_tools") status = kh_cli_kmc_tools(ctx, args, out, err);
            else err = "khoice_server: unknown tool " + tool + "\n";
            if (chdir(home) != 0) { /* stay */ }
            kh_wire_send_u32(fd, (uint32_t)status) && kh_wire_send_str(fd, out) && kh_wire_send_str(fd, err);
            ++served;
        }
        ::close(fd);
    }
    ::close(srv);
    unlink(path);
    kh_ctx_destroy(ctx);
    fprintf(stderr, "khoice_server: served %lu requests\n", served);
    return 0;
}
