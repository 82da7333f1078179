#!/bin/bash
# P-256 CCPoS leg with the multi-exponentiation's windows unsigned / signed (VMN_SIGNED_WINDOWS).   (gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sw in 0 1 0 1; do
  VMN_SIGNED_WINDOWS=$sw python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements 0 --ccpos-elements 0 --ec-elements ${1:-1000000} --decrypt-elements 0 --skip-cpu --no-e2e 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); m=r['mix_ec_p256']
print('VMN_SIGNED_WINDOWS=$sw online_ms=%.2f ct/s=%.4g kernels=%s' % (m['online_ms'], m['ciphertexts_per_s_online'], m['kernel_ms_by_family']))"
done
