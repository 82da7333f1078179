#!/bin/bash
# The bench line of a round and the rocprofv3 per-kernel table of the same command.   usage: tools/final_bench.sh <tag>   (gpurun)
#   -> gpurun_out/<tag>.json, gpurun_out/<tag>_kernel_stats.csv      (copy both to profiles/)
tag=${1:-r03_bench_v1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 __graft_entry__.py > gpurun_out/fb_build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/fb_build.log; exit 1; }
python3 -u bench.py > gpurun_out/$tag.json 2> gpurun_out/$tag.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --skip-cpu > gpurun_out/${tag}_rocprof.json 2> gpurun_out/${tag}_rocprof.err || exit 1
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
head -c 600 gpurun_out/$tag.json; echo; head -5 gpurun_out/${tag}_kernel_stats.csv
