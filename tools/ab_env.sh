#!/bin/bash
# Timing under environment variants on the GPU box: tools/ab_env.sh TAG "VAR=val VAR2=val" ...  ("-" = no variables)
TAG=$1; shift
i=0
for v in "$@"; do
  i=$((i+1))
  out=gpurun_out/abe_${TAG}_$i.log
  if [ "$v" = "-" ]; then v=""; fi
  env $v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out 2>&1
  python - "$out" "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2] or "-", d["ms_per_step"], {k:v for k,v in d["kernel_ms_per_step"].items() if v}, "replans", d["replans"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
