#!/bin/bash
# GPU-idle attribution of the P-256 CCPoS leg at N ciphertexts of width 3 (see tools/idle_gaps.py).  usage: tools/trace_gaps_ec.sh N OUTDIR
n=${1:-1000000}; out=${2:-gpurun_out/gaps_ec}
rm -rf "$out"; mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python3 __graft_entry__.py > "$out/build.log" 2>&1 || { echo "build failed"; tail -5 "$out/build.log"; exit 1; }
VMN_TRACE_EVENTS=$out/events.csv rocprofv3 --kernel-trace --output-format csv -d "$out/prof" -- \
  python3 bench.py --elements 2000 --mix-elements 0 --ec-elements $n --ccpos-elements 0 --decrypt-elements 0 --skip-cpu --no-e2e --steps 1 --warmup 0 > "$out/bench.json" 2> "$out/bench.err" || exit 1
kt=$(find "$out/prof" -name '*kernel_trace.csv' | head -1)
for w in vmn_shuffle_reencrypt ccpos:commit ccpos:reply ccpos:set_commitment ccpos:compute_ab ccpos:verify; do
  echo "=== $w" >> "$out/gaps.txt"
  python3 tools/idle_gaps.py "$out/events.csv" "$kt" --window $w >> "$out/gaps.txt" 2>&1
done
rm -rf "$out/prof" "$out/events.csv"
cat "$out/gaps.txt"
