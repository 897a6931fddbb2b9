// bin/kmc_tools — drop-in for KMC 3's `kmc_tools` (see kh_cli.cpp for the call forms).  With
// $KHOICE_SERVER set the request is served by the resident bin/khoice_server (one HIP context for
// the whole workflow); otherwise this process opens its own.
#include <cstdio>
#include <cstdlib>

#include "kh_cli.h"

int main(int argc, char** argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    int status = 1;
    if (kh_cli_try_server("kmc_tools", args, &status)) return status;
    const char* dev_env = getenv("KHOICE_GPU_DEVICE");
    kh_ctx* ctx = nullptr;
    if (kh_ctx_create(dev_env ? atoi(dev_env) : 0, &ctx) != KH_OK) {
        fprintf(stderr, "kmc_tools: %s\n", kh_last_error());
        return 1;
    }
    std::string out, err;
    status = kh_cli_kmc_tools(ctx, args, out, err);
    fputs(out.c_str(), stdout);
    fputs(err.c_str(), stderr);
    kh_ctx_destroy(ctx);
    return status;
}
