#!/bin/bash
# Variant build of the library for A/B timing: tools/build_variant.sh NAME -DFOO=1 ...
# -> khoice_amd/lib/variants/libkhoice_hip_NAME.so ; run with KHOICE_HIP_LIB=<that path>.  Never shipped.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/khoice_amd/lib/variants/$NAME
mkdir -p $OUT
for f in kh_kernels.hip kh_skm.hip kh_skm2.hip kh_ingest.hip kh_engine.cpp kh_io.cpp kh_comm.cpp; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip "$@" -I $ROOT/include -c $ROOT/khoice_amd/csrc/$f -o $OUT/${f%.*}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/khoice_amd/lib/variants/libkhoice_hip_$NAME.so $OUT/*.o -lz -ldl
echo built $ROOT/khoice_amd/lib/variants/libkhoice_hip_$NAME.so
