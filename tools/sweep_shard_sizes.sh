#!/bin/bash
# One GPU's share of a strong-scaled 1M-ciphertext job at G = 1, 2, 4, 8: the single-GPU legs at N = 10^6 / G (what a rank of
# the sharded drivers executes, minus the scalar exchanges) -- input of the MODELLED scaling table of DESIGN.md §7.
cd "$GRAFT_REPO_ROOT"
for g in 1 2 4 8; do
  n=$((1000000 / g))
  python3 bench.py --steps 2 --warmup 1 --elements $n --mix-elements $n --ec-elements $n --ccpos-elements $n --decrypt-elements 0 --no-e2e --skip-cpu \
      > gpurun_out/shard_$g.json 2> gpurun_out/shard_$g.err || { echo "G=$g failed"; tail -3 gpurun_out/shard_$g.err; exit 1; }
  python3 - "$g" "$n" <<'PY'
import json, sys
g, n = sys.argv[1], int(sys.argv[2])
d = json.loads([l for l in open(f"gpurun_out/shard_{g}.json") if l.startswith("{")][-1])
print(f"G={g} n/GPU={n:>8}: modpow {d['ms_per_step']:8.2f} ms/step | PoS-2048 {d['mix_prove']['total_ms']:8.1f} ms | CCPoS-3072 online {d['mix_ccpos_3072']['online_ms']:7.1f} ms "
      f"| P-256 w3 online {d['mix_ec_p256']['online_ms']:6.1f} ms", flush=True)
PY
done
