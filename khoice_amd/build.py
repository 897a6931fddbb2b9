"""Build the native pieces in-tree with hipcc / gcc (no cmake, no JIT cache).

    python -m khoice_amd.build            # library + CLIs (+ oracle checker)
    python -m khoice_amd.build --force

Outputs (git-ignored, but they travel to the GPU box with the snapshot):
    khoice_amd/lib/libkhoice_hip.so     HIP engine, gfx950 code objects only
    bin/kmc, bin/kmc_tools              drop-in executables (argv of the 7 call forms)
    bin/khoice_server                   resident engine the two can forward to ($KHOICE_SERVER)
    oracle/_build/libkh_oracle.so       C restatement used by tests / cpu_baseline only
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "khoice_amd", "csrc")
LIBDIR = os.path.join(ROOT, "khoice_amd", "lib")
BINDIR = os.path.join(ROOT, "bin")
LIB = os.path.join(LIBDIR, "libkhoice_hip.so")
ARCH = "gfx950"

HIP_SOURCES = ["kh_kernels.hip", "kh_skm.hip", "kh_skm2.hip", "kh_ingest.hip", "kh_engine.cpp", "kh_io.cpp", "kh_comm.cpp"]
HEADERS = ["kh_common.h", "kh_device.h", "kh_skm_device.h", "kh_launch.h", "kh_engine.h", os.path.join(ROOT, "include", "khoice_hip.h")]
CLIS = {"kmc": "kmc_main.cpp", "kmc_tools": "kmc_tools_main.cpp", "khoice_server": "kh_server_main.cpp"}
CLI_COMMON = "kh_cli.cpp"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; khoice_amd needs the ROCm toolchain to build")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_library(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in HIP_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, [path] + hdrs):
            _run([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
                  "-I", os.path.join(ROOT, "include"), "-c", path, "-o", obj])
    if force or _newer(LIB, objs):
        # --no-undefined: a launcher declared in kh_launch.h but not defined must fail the build,
        # not the first dlopen on the GPU box
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", LIB] + objs + ["-lz", "-ldl"])
    return LIB


def build_clis(force: bool = False):
    os.makedirs(BINDIR, exist_ok=True)
    hipcc = _hipcc()
    out = []
    for name, src in CLIS.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        exe = os.path.join(BINDIR, name)
        common = os.path.join(CSRC, CLI_COMMON)
        if force or _newer(exe, [path, common, os.path.join(CSRC, "kh_cli.h"), LIB,
                                 os.path.join(ROOT, "include", "khoice_hip.h")]):
            _run([hipcc, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, path, common,
                  "-o", exe, "-L", LIBDIR, "-lkhoice_hip", f"-Wl,-rpath,$ORIGIN/../khoice_amd/lib"])
        out.append(exe)
    return out


def build_oracle(force: bool = False):
    mk = os.path.join(ROOT, "oracle", "Makefile")
    if os.path.exists(mk):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle")] + (["-B"] if force else []))


def build_all(force: bool = False):
    lib = build_library(force)
    build_clis(force)
    build_oracle(force)
    return lib


if __name__ == "__main__":
    build_all("--force" in sys.argv)
