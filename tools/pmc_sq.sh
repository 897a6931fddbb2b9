#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the bench kernels (one pass, 8 SQ slots).  tools/pmc_sq.sh TAG [bench args]
TAG=${1:-sq}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/b.log 2>&1
python3 tools/summarize_sq.py $OUT $TAG
