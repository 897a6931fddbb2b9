#!/bin/bash
# Diagnostic build with in-kernel phase stamps (-DKH_STAMPS): never shipped, never benchmarked.
# Usage: tools/build_diag.sh ; KHOICE_HIP_LIB=khoice_amd/lib/diag/libkhoice_hip_diag.so python bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/khoice_amd/lib/diag
mkdir -p $OUT
for f in kh_kernels.hip kh_skm.hip kh_skm2.hip kh_ingest.hip kh_engine.cpp kh_io.cpp kh_comm.cpp; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DKH_STAMPS -I $ROOT/include -c $ROOT/khoice_amd/csrc/$f -o $OUT/${f%.*}.o
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libkhoice_hip_diag.so $OUT/*.o -lz -ldl
echo built $OUT/libkhoice_hip_diag.so
