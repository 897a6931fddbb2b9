#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats and HBM traffic counters of bench.py.
# Usage: tools/collect_profiles.sh <tag> [extra bench.py arguments, e.g. --k 41]      -> gpurun_out/<tag>_*.csv|json
set -e
TAG=${1:-rXX}
shift || true
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/write.log 2>&1
python3 tools/summarize_profiles.py $OUT $TAG
