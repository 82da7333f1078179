#!/bin/bash
# the weight of one bucket in the aggregation against one first-level addition (pick_bucket_bits), over the three legs
for w in 0.5 1 2 3 5; do
  VMN_BUCKET_AGG_WEIGHT=$w python3 -u bench.py --elements 2000 --skip-cpu --no-e2e --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('w=$w', 'mix %.1f' % d['mix_prove']['total_ms'], 'n10k %.1f' % d['mix_prove_n10000']['total_ms'], 'ccpos %.1f' % d['mix_ccpos_3072']['online_ms'], 'ec %.1f' % d['mix_ec_p256']['online_ms'],
      {k:v for k,v in d['mix_ec_p256']['kernel_ms_by_family'].items() if v>3})"
done
