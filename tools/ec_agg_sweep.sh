for w in 3 6 10 16 24; do
  VMN_BUCKET_AGG_WEIGHT=$w python3 -u bench.py --elements 2000 --mix-elements 0 --ec-elements 1000000 --ccpos-elements 0 --skip-cpu --no-e2e --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d['mix_ec_p256']
print('w=$w', round(m['online_ms'],1), {k:v for k,v in m['kernel_ms_by_family'].items() if v>3})"
done
