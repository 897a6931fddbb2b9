#!/bin/bash
# A/B timing on the GPU box: tools/ab.sh TAG [lib ...]  ("default" = the shipped library)
TAG=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset KHOICE_HIP_LIB; else export KHOICE_HIP_LIB=$v; fi
  out=gpurun_out/ab_${TAG}_$(basename "$v" .so).log
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out 2>&1
  python - "$out" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["ms_per_step"], {k:v for k,v in d["kernel_ms_per_step"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
