#!/bin/bash
# rocprofv3 passes for the headline kernel (run on the GPU box through gpurun):
#   1. --kernel-trace --stats       per-kernel durations
#   2. --pmc FETCH_SIZE             HBM read side   (own pass: TCC slots)
#   3. --pmc WRITE_SIZE             HBM write side  (own pass)
#   4. --pmc SQ_*                   VALU instruction / busy counters
# Counter passes are never combined with --sys-trace / hip / hsa tracing (see the pool rules).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${1:-r01}
N=${2:-262144}
mkdir -p $OUT
CMD="python3 bench.py --steps 1 --warmup 1 --elements $N --mix-elements 0 --skip-cpu"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
echo "rc=$?"
find $OUT -name "*.csv" | head -20
