#!/bin/bash
# Below how many points (k arrays x n) a multi-exponentiation over a curve adds its rows as they are instead of normalising them
# first (VMN_EC_NORMALISE_MIN) against the P-256 legs (tools/ec_quick.sh).   usage (GPU box): bash tools/sweep_ec_normalise_min.sh "0 32768 131072 524288"
for t in ${1:-0 32768 131072 524288}; do
  echo "min=$t: $(VMN_EC_NORMALISE_MIN=$t bash tools/ec_quick.sh r04_ec_normmin_$t | tr '\n' ' ')"
done
