#!/bin/bash
# PoS-2048 and CCPoS-3072 legs with the window width of the multi-exponentiation forced (VMN_WINDOW_BITS).   (gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in default ${WIDTHS:-13 14 15 16}; do
  if [ "$c" = default ]; then unset VMN_WINDOW_BITS; else export VMN_WINDOW_BITS=$c; fi
  python3 bench.py --steps 1 --warmup 0 --elements 2048 --mix-elements ${1:-1000000} --ccpos-elements ${1:-1000000} --ec-elements 0 --decrypt-elements 0 --skip-cpu --no-e2e 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); m=r['mix_prove']; c3=r['mix_ccpos_3072']
k=m['kernel_ms_by_family']; k3=c3['kernel_ms_by_family']
print('c=$c PoS total_ms=%.1f expprod=%.1f sort=%.1f agg=%.1f scan=%.1f reduce=%.1f | ccpos online_ms=%.1f expprod=%.1f sort=%.1f scan=%.1f' % (m['total_ms'], k['expprod'], k.get('expprod_sort',0), k.get('expprod_agg',0), k.get('scan',0), k.get('reduce',0), c3['online_ms'], k3['expprod'], k3.get('expprod_sort',0), k3.get('scan',0)))"
done
