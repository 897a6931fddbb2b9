#!/bin/bash
# tools/ab_env_k.sh TAG K "VAR=val ..." ...   ("-" = no variables): bench at k = K under environment variants
TAG=$1; K=$2; shift; shift
i=0
for v in "$@"; do
  i=$((i+1)); out=gpurun_out/abek_${TAG}_${K}_$i.log
  if [ "$v" = "-" ]; then v=""; fi
  env $v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --k $K > $out 2>&1
  python - "$out" "$K $v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2] or "-", d["ms_per_step"], {k:v for k,v in d["kernel_ms_per_step"].items() if v}, "replans", d["replans"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
