#!/bin/bash
# round 6, call c: verification in the few-party regime, the sweep's crossover for individual proofs, tail choices for 4 / 8 parties
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out/r6c; mkdir -p $OUT
export DAPOL_ENV_KNOBS=1
timeout -k 10 400 python tools/bench_small_parties.py --only verify --aggs 32,24,0 --reps 2 > $OUT/verify.log 2>&1 || { tail -20 $OUT/verify.log; exit 1; }
grep "^verify" $OUT/verify.log
for lg in 10 11 12 13 14; do
  echo "== m=1, 2^$lg proofs: default / forced sweep"
  timeout -k 10 200 python tools/bench_small_parties.py --only batch --ms 1 --proofs $lg --reps 3 2>&1 | grep "^batch" | cut -c1-160
  DAPOL_GS_SMALL_MIN=64 timeout -k 10 200 python tools/bench_small_parties.py --only batch --ms 1 --proofs $lg --reps 3 2>&1 | grep "^batch" | cut -c1-160
done
for m in 2 4 8; do
  echo "== m=$m 2^16: default, TAIL_N=32, NO_TAIL, TAIL_N=64"
  timeout -k 10 200 python tools/bench_small_parties.py --only batch --ms $m --proofs 16 --reps 2 2>&1 | grep "^batch" | cut -c1-160
  DAPOL_TAIL_N=32 timeout -k 10 200 python tools/bench_small_parties.py --only batch --ms $m --proofs 16 --reps 2 2>&1 | grep "^batch" | cut -c1-160
  DAPOL_NO_TAIL=1 timeout -k 10 200 python tools/bench_small_parties.py --only batch --ms $m --proofs 16 --reps 2 2>&1 | grep "^batch" | cut -c1-160
  DAPOL_TAIL_N=64 timeout -k 10 200 python tools/bench_small_parties.py --only batch --ms $m --proofs 16 --reps 2 2>&1 | grep "^batch" | cut -c1-160
done
