cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wg in 2 4 8 16; do
  VMN_BUCKET_AGG_WEIGHT=$wg python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements 0 --ccpos-elements 0 --ec-elements 1000000 --decrypt-elements 0 --skip-cpu --no-e2e 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); m=r['mix_ec_p256']
k=m['kernel_ms_by_family']
print('signed, VMN_BUCKET_AGG_WEIGHT=$wg online_ms=%.2f ct/s=%.4g expprod=%.2f agg=%.2f sort=%.2f' % (m['online_ms'], m['ciphertexts_per_s_online'], k['expprod'], k['expprod_agg'], k['expprod_sort']))"
done
