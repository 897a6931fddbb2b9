#!/bin/bash
# Headline shape at two-word k, super-k-mer form vs key-array form: tools/ab_k2.sh TAG K ...
TAG=$1; shift
for K in "$@"; do
  for mode in skm keyarray; do
    if [ $mode = keyarray ]; then export KHOICE_NO_SKM2=1; else unset KHOICE_NO_SKM2; fi
    out=gpurun_out/abk2_${TAG}_${K}_${mode}.log
    KHOICE_SKM_DEBUG=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --k $K > $out 2> $out.err
    python - "$out" "$K $mode" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["ms_per_step"], {k:v for k,v in d["kernel_ms_per_step"].items() if v}, "replans", d["replans"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    grep "skm\]" $out.err | tail -1 | cut -c1-330
  done
done
