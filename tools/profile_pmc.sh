#!/bin/bash
# rocprofv3 passes for the kernels that carry the numbers of bench.py (run on the GPU box through gpurun):
#   run A  the headline workload alone (k_modpow<Cfg<74,1>> at N elements)
#   run B  the proof legs (mix + prove at 2048 bits, CCPoS at 3072 bits, P-256 width 3) at N ciphertexts
# each as four separate passes:
#   1. --kernel-trace --stats       per-kernel durations
#   2. --pmc FETCH_SIZE             HBM read side   (own pass: TCC slots)
#   3. --pmc WRITE_SIZE             HBM write side  (own pass)
#   4. --pmc SQ_*                   VALU instruction / busy counters
# Counter passes are never combined with --sys-trace / hip / hsa tracing (see the pool rules).  The libraries are built
# BEFORE any rocprofv3 pass: the profiler's preloaded library initialises the GPU, and a process that has done so must
# not spawn the compilers.
#
# usage: tools/profile_pmc.sh <tag> [N]   ->  gpurun_out/pmc_<tag>/{A,B}/{trace,fetch,write,sq}
#        then: python3 tools/summarize_pmc.py gpurun_out/pmc_<tag> <N> profiles/<tag>_pmc_kernels.json
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${1:-r04}
N=${2:-262144}
mkdir -p $OUT
python3 __graft_entry__.py > $OUT/build.log 2>&1 || { echo "build failed"; tail -5 $OUT/build.log; exit 1; }
CMD_A="python3 bench.py --steps 1 --warmup 1 --elements $N --mix-elements 0 --ccpos-elements 0 --ec-elements 0 --decrypt-elements 0 --skip-cpu --no-shapes"
CMD_B="python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements $N --ccpos-elements $N --ec-elements $N --decrypt-elements $N --skip-cpu --no-e2e"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
run() {   # run <dir> <command...>
  local d=$1; shift
  mkdir -p $OUT/$d
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$d/trace -- "$@" > $OUT/$d/trace.log 2>&1 &&
  echo "$d trace done" &&
  timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$d/fetch -- "$@" > $OUT/$d/fetch.log 2>&1 &&
  echo "$d fetch done" &&
  timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$d/write -- "$@" > $OUT/$d/write.log 2>&1 &&
  echo "$d write done" &&
  timeout -k 10 500 rocprofv3 --pmc $SQ --output-format csv -d $OUT/$d/sq -- "$@" > $OUT/$d/sq.log 2>&1 &&
  echo "$d sq done"
}
run A $CMD_A && run B $CMD_B
echo "rc=$?"
find $OUT -name "*.csv" | head -40
