#!/bin/bash
# Values per lane and level of the batched inversion that normalises curve rows (VMN_EC_NORMALISE_CHUNK) against the P-256 legs.
# usage (GPU box): bash tools/sweep_ec_normalise_chunk.sh "2 4 8 16"
for k in ${1:-2 4 8 16}; do
  echo "K=$k: $(VMN_EC_NORMALISE_CHUNK=$k bash tools/ec_quick.sh r04_ec_normchunk_$k | tr '\n' ' ')"
done
