#!/bin/bash
# A/B of two builds on ONE box, interleaved: tools/ab_libs.sh <tag> <other .so> [log2 entities] [rounds]
# (the other build is loaded through DAPOL_HIP_LIB; box-to-box spread is +-3 %, so builds are only compared within a call)
set -o pipefail
tag=${1:-ab}; other=$2; lg=${3:-19}; rounds=${4:-2}
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/${tag}_ab.txt
for i in $(seq 1 $rounds); do
  for which in new other; do
    if [ $which = other ]; then export DAPOL_HIP_LIB=$other; else unset DAPOL_HIP_LIB; fi
    v=$(python3 bench.py --no-cpu-baseline --no-secondary --log2-entities $lg --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['checksum'])") || exit 1
    echo "$which $v" | tee -a $OUT/${tag}_ab.txt
  done
done
