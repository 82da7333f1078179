#!/bin/bash
# The PoS leg at small sizes with check (B)'s two powers as one launch (vmn_garray_exp_pair) and as two.   (gpurun)
#   -> gpurun_out/pair_sweep.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pair_sweep.txt
: > $out
for n in 2000 5000 10000 20000; do
  for pm in 0 20480; do
    VMN_PAIR_MAX=$pm python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements $n --ccpos-elements 0 --ec-elements 0 \
        --decrypt-elements 0 --skip-cpu --no-e2e 2> gpurun_out/pair_sweep.err |
      python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); m=r['mix_prove']
print('N=$n VMN_PAIR_MAX=$pm  total_ms=%.2f  ct/s=%.4g  verify_ms=%.2f  accepted=%s  kernel_ms=%s' % (m['total_ms'], m['ciphertexts_per_s'], m['verify_ms'], m['accepted'], m['kernel_ms_by_family']))" >> $out || exit 1
  done
done
cat $out
