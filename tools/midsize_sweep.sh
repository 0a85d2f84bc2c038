#!/bin/bash
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
# tools/midsize_sweep.sh: one prove_entities call of b proofs (tools/bench_midsize_one.py) under the lanes-per-list / small-call knobs
cd "$(dirname "$0")/.."
for b in ${BS:-128 256 512 1024 1536}; do
  for cfg in "-" "DAPOL_LPL=16" "DAPOL_LPL=8" "DAPOL_SMALL_TAIL=1" "DAPOL_NO_QUAD=1"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    echo "$b $cfg $(env $e python tools/bench_midsize_one.py $b 2>&1 | tail -1)"
  done
done
