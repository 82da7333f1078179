cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 __graft_entry__.py > gpurun_out/fb_build.log 2>&1
python3 -u bench.py > gpurun_out/r02_bench_v2.json 2> gpurun_out/r02_bench_v2.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_bench_v2_prof -- python3 bench.py --skip-cpu > gpurun_out/r02_bench_v2_rocprof.json 2> gpurun_out/r02_bench_v2_rocprof.err || exit 1
f=$(find gpurun_out/r02_bench_v2_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02_bench_v2_kernel_stats.csv
rm -rf gpurun_out/r02_bench_v2_prof
head -c 600 gpurun_out/r02_bench_v2.json; echo; head -5 gpurun_out/r02_bench_v2_kernel_stats.csv
