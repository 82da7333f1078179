for n in 2000 10000 16000 25000; do
 for w8 in 0 1000000000; do
  echo "== N=$n VMN_WIDE8_MAX=$w8"
  VMN_WIDE8_MAX=$w8 python3 -u bench.py --elements $n --mix-elements $n --ec-elements 0 --ccpos-elements 0 --skip-cpu --no-e2e --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d['mix_prove']
print(json.dumps({'modpow_ms':round(d['ms_per_step'],2),'mix_ct_per_s':round(m['ciphertexts_per_s']),'mix_total_ms':round(m['total_ms'],1),'kernel_ms':{k:v for k,v in m['kernel_ms_by_family'].items() if v>0.5}}))"
 done
done
