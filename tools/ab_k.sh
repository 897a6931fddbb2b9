#!/bin/bash
# Timing of the headline shape at other k / minimizer lengths: tools/ab_k.sh TAG "K M" ...   (M = - : default)
TAG=$1; shift
for v in "$@"; do
  set -- $v; K=$1; M=$2
  if [ "$M" = "-" ]; then unset KHOICE_SKM_M; else export KHOICE_SKM_M=$M; fi
  out=gpurun_out/abk_${TAG}_${K}_${M}.log
  KHOICE_SKM_DEBUG=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --k $K > $out 2> $out.err
  python - "$out" "$K $M" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["ms_per_step"], {k:v for k,v in d["kernel_ms_per_step"].items() if v}, "replans", d["replans"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  grep "skm\]" $out.err | tail -1 | cut -c1-330
done
