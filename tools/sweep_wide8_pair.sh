cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w8 in 6144 12000 20000 40000; do
  for n in 5000 10000; do
  VMN_WIDE8_MAX=$w8 python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements $n --ccpos-elements 0 --ec-elements 0 --decrypt-elements 0 --skip-cpu --no-e2e 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); m=r['mix_prove']
print('N=$n WIDE8_MAX=$w8 total_ms=%.2f verify_ms=%.2f modpow=%s fixed=%s expprod=%s' % (m['total_ms'], m['verify_ms'], m['kernel_ms_by_family'].get('modpow'), m['kernel_ms_by_family'].get('fixed'), m['kernel_ms_by_family'].get('expprod')))"
  done
done
