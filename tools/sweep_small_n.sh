#!/bin/bash
# Small-N sweep: the mix + prove leg (PoS, 2048-bit), the configs[2] leg (3072-bit PoSC + CCPoS) and the headline modpow at N
# elements, with the wide geometry off (VMN_WIDE_MAX=0) and forced (huge).  usage: tools/sweep_small_n.sh OUTFILE N [N ...]
out=$1; shift
: > "$out"
for n in "$@"; do
  for w in 0 1000000000; do
    echo "== N=$n VMN_WIDE_MAX=$w" >> "$out"
    VMN_WIDE_MAX=$w python3 -u bench.py --elements $n --mix-elements $n --ec-elements 0 --ccpos-elements ${CCPOS_N:-0} --skip-cpu --no-e2e --steps 3 --warmup 1 2>/dev/null \
      | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
m=d.get('mix_prove',{})
c=d.get('mix_ccpos_3072',{})
print(json.dumps({'modexp_per_s':d['value'],'ms':d['ms_per_step'],'mix_ct_per_s':m.get('ciphertexts_per_s'),'mix_total_ms':m.get('total_ms'),'prove_ms':m.get('prove_ms'),'verify_ms':m.get('verify_ms'),'kernel_ms':m.get('kernel_ms_by_family'),
 'ccpos_online_ct_per_s':c.get('ciphertexts_per_s_online'),'ccpos_online_ms':c.get('online_ms'),'ccpos_offline_ms':c.get('offline_ms'),'ccpos_kernel_ms':c.get('kernel_ms_by_family')}))" >> "$out" || exit 1
  done
done
