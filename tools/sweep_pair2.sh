#!/bin/bash
# Between the paired launch and the combined form of check (B): PoS leg at 24k .. 100k ciphertexts.   (gpurun)  -> gpurun_out/pair_sweep2.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pair_sweep2.txt
: > $out
for n in ${SIZES:-24000 32768 50000 70000 100000}; do
  for mode in "0 32768" "1000000 1000000000" "0 1000000000"; do
    set -- $mode
    VMN_PAIR_MAX=$1 VMN_COMBINED_MIN=$2 python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements $n --ccpos-elements 0 --ec-elements 0 \
        --decrypt-elements 0 --skip-cpu --no-e2e 2> gpurun_out/pair_sweep.err |
      python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); m=r['mix_prove']
print('N=$n VMN_PAIR_MAX=$1 VMN_COMBINED_MIN=$2 total_ms=%.2f  ct/s=%.4g  verify_ms=%.2f  accepted=%s  modpow_ms=%s' % (m['total_ms'], m['ciphertexts_per_s'], m['verify_ms'], m['accepted'], m['kernel_ms_by_family'].get('modpow')))" >> $out || exit 1
  done
done
cat $out
