#!/bin/bash
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
# Interleaved A/B of environment settings on one box: tools/ab.sh <repeats> <log2_entities> "ENV1=.. ENV2=.." "ENV=.." ...
# ("-" = defaults).  Prints one line per run: <config> <entities/s>.
reps=$1; shift; lg=$1; shift
for i in $(seq 1 $reps); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    v=$(env $e python bench.py --no-cpu-baseline --no-secondary --log2-entities $lg --warmup 1 --steps 1 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*' | cut -d' ' -f2)
    echo "$cfg $v"
  done
done
