#!/bin/bash
# Host scopes (VMN_TRACE_EVENTS, csrc/hosttrace.h) + rocprofv3 kernel trace of one bench.py run, kept for tools/timeline.py.
# usage: tools/trace_bench.sh OUTDIR [bench.py arguments ...]
out=${1:?outdir}; shift
rm -rf "$out"; mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
# build BEFORE the profiler is involved: under rocprofv3 the preloaded library has already initialised the GPU
python3 __graft_entry__.py > "$out/build.log" 2>&1 || { echo "build failed"; tail -5 "$out/build.log"; exit 1; }
VMN_TRACE_EVENTS=$out/events.csv rocprofv3 --kernel-trace --output-format csv -d "$out/prof" -- \
  python3 bench.py "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
kt=$(find "$out/prof" -name '*kernel_trace.csv' | head -1)
cp "$kt" "$out/kernel_trace.csv"; rm -rf "$out/prof"
ls -la "$out"
