for cfg in "15 9" "15 10" "16 10" "16 11" "17 11" "17 12" "18 12" "18 13" "19 13"; do set -- $cfg; KHOICE_SKM_M=$2 timeout -k 10 200 python bench.py --k $1 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k',d['config']['k'],'m',$2, d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if v and k.startswith('skm')}, 'replans', d['replans'])"; done
