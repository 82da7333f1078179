#!/bin/bash
# Fan-in of the bucket trees (VMN_TREE_FANIN) against the P-256 legs (tools/ec_quick.sh).  usage (GPU box): bash tools/sweep_ec_fanin.sh "6 8 12 16"
for f in ${1:-6 8 12 16}; do
  echo "F=$f: $(VMN_TREE_FANIN=$f bash tools/ec_quick.sh r04_ec_fanin_$f | tr '\n' ' ')"
done
