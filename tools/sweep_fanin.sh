#!/bin/bash
# fan-in of the per-bucket product trees (VMN_TREE_FANIN): the mix + prove leg at 2048 bits and the P-256 width-3 leg
cd "$GRAFT_REPO_ROOT"
for f in ${FANINS:-8 12 16 24 32}; do
  VMN_TREE_FANIN=$f python3 bench.py --steps 1 --warmup 0 --elements 2048 --no-e2e --ccpos-elements 0 --decrypt-elements 0 --skip-cpu \
      > gpurun_out/fanin_$f.json 2> gpurun_out/fanin_$f.err || { echo "F=$f failed"; tail -3 gpurun_out/fanin_$f.err; exit 1; }
  python3 - "$f" <<'PY'
import json, sys
f = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/fanin_{f}.json") if l.startswith("{")][-1])
mp, ec, sm = d["mix_prove"], d["mix_ec_p256"], d["mix_prove_n10000"]
print(f"F={f:>3}: PoS-2048 1M {mp['total_ms']:7.1f} ms (expprod {mp['kernel_ms_by_family'].get('expprod')})  |  P-256 w3 1M online {ec['online_ms']:6.1f} ms "
      f"(expprod {ec['kernel_ms_by_family'].get('expprod')}, scan {ec['kernel_ms_by_family'].get('scan')})  |  PoS n=10^4 {sm['total_ms']:5.1f} ms", flush=True)
PY
done
