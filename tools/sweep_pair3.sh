#!/bin/bash
# The paired launch's geometry rule: PoS leg at sizes around the wide-geometry boundaries.   (gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in ${SIZES:-3000 5000 8000 10000 14000 16000 18000 20000 24000 32000}; do
  python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements $n --ccpos-elements 0 --ec-elements 0 --decrypt-elements 0 --skip-cpu --no-e2e 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); m=r['mix_prove']
print('N=$n total_ms=%.2f ct/s=%.4g verify_ms=%.2f modpow=%s' % (m['total_ms'], m['ciphertexts_per_s'], m['verify_ms'], m['kernel_ms_by_family'].get('modpow')))"
done
