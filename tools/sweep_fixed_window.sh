#!/bin/bash
# Window of the fixed-base tables of the long-lived bases (g, the public key): VMN_FIXED_WINDOW_REUSE = 19 (default choice at
# 2048 bits: 17 GB per table), 20 (33 GB), 21 (61 GB) -- the PoS leg at N = 10^6.
cd "$GRAFT_REPO_ROOT"
for w in 19 20 21; do
  VMN_FIXED_WINDOW_REUSE=$w VMN_FIXED_CACHE_BYTES=220000000000 python3 bench.py --steps 1 --warmup 0 --elements 2048 --no-e2e --ccpos-elements 0 --ec-elements 0 \
      --decrypt-elements 0 --skip-cpu > gpurun_out/fixedw_$w.json 2> gpurun_out/fixedw_$w.err || { echo "w=$w failed"; tail -3 gpurun_out/fixedw_$w.err; continue; }
  python3 - "$w" <<'PY'
import json, sys
w = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/fixedw_{w}.json") if l.startswith("{")][-1])
mp = d["mix_prove"]
print(f"w={w}: PoS-2048 1M {mp['total_ms']:7.1f} ms  fixed {mp['kernel_ms_by_family'].get('fixed')}  fixed_table {mp['kernel_ms_by_family'].get('fixed_table')}", flush=True)
PY
done
