#!/bin/bash
# GPU-idle attribution of the mix + prove leg at N ciphertexts (see tools/idle_gaps.py).  usage: tools/trace_gaps.sh N OUTDIR
n=${1:-10000}; out=${2:-gpurun_out/gaps}
rm -rf "$out"; mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
# build BEFORE the profiler is involved: under rocprofv3 the preloaded library has already initialised the GPU, and bench.py
# refuses to start compilers from such a process
python3 __graft_entry__.py > "$out/build.log" 2>&1 || { echo "build failed"; tail -5 "$out/build.log"; exit 1; }
VMN_TRACE_EVENTS=$out/events.csv rocprofv3 --kernel-trace --output-format csv -d "$out/prof" -- \
  python3 bench.py --elements 2000 --mix-elements $n --ec-elements 0 --ccpos-elements 0 --decrypt-elements 0 --skip-cpu --no-e2e --steps 2 --warmup 1 > "$out/bench.json" 2> "$out/bench.err" || exit 1
kt=$(find "$out/prof" -name '*kernel_trace.csv' | head -1)
for w in pos:precompute@2 pos:precompute@1 pos:commit_prepare pos:commit pos:reply pos:set_commitment pos:compute_af pos:verify; do
  echo "=== $w" >> "$out/gaps.txt"
  python3 tools/idle_gaps.py "$out/events.csv" "$kt" --window $w >> "$out/gaps.txt" 2>&1
done
cp "$kt" "$out/kernel_trace.csv"; rm -rf "$out/prof"
