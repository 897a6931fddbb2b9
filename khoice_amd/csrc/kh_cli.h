// khoice_amd — command-line front ends as library functions, shared by the stand-alone
// executables (bin/kmc, bin/kmc_tools) and the resident server (bin/khoice_server).
#pragma once
#include <string>
#include <vector>

#include "khoice_hip.h"

// Both return the process exit status (0 = success); text for stdout / stderr is appended to
// `out` / `err`.  `ctx` must be a live engine context.
int kh_cli_kmc(kh_ctx* ctx, const std::vector<std::string>& args, std::string& out, std::string& err);
int kh_cli_kmc_tools(kh_ctx* ctx, const std::vector<std::string>& args, std::string& out, std::string& err);

// Client side of the resident server: when $KHOICE_SERVER names a Unix socket with a live
// server behind it, forward `tool args` (and the caller's working directory) to it and return
// true with the remote exit status in *status; otherwise return false (run locally).
bool kh_cli_try_server(const char* tool, const std::vector<std::string>& args, int* status);
