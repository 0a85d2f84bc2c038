#!/bin/bash
# round 6, call g: lane-per-proof A commitment, Fiat-Shamir shapes for bulk short proofs, the sweep below 1,024 proofs
set -o pipefail
export DAPOL_ENV_KNOBS=1
OUT=gpurun_out/r6g; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_small_parties.py -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
{
for i in 1 2; do
echo "== m=1 2^17: default | NO_A_LANE | FS_SHAPE=1 | FS_SHAPE=2"
python3 tools/bench_small_parties.py --only batch --ms 1 --reps 3 2>&1 | grep "^batch" | cut -c1-130
DAPOL_NO_A_LANE=1 python3 tools/bench_small_parties.py --only batch --ms 1 --reps 3 2>&1 | grep "^batch" | cut -c1-130
DAPOL_FS_SHAPE=1 python3 tools/bench_small_parties.py --only batch --ms 1 --reps 3 2>&1 | grep "^batch" | cut -c1-130
done
echo "== m=2 2^17: default | FS_SHAPE=1"
python3 tools/bench_small_parties.py --only batch --ms 2 --reps 2 2>&1 | grep "^batch" | cut -c1-130
DAPOL_FS_SHAPE=1 python3 tools/bench_small_parties.py --only batch --ms 2 --reps 2 2>&1 | grep "^batch" | cut -c1-130
for m in 1 2 4 8; do for lg in 8 9; do
  echo "== m=$m 2^$lg: default | forced sweep"
  python3 tools/bench_small_parties.py --only batch --ms $m --proofs $lg --reps 3 2>&1 | grep "^batch" | cut -c1-110
  DAPOL_GS_SMALL_MIN=64 python3 tools/bench_small_parties.py --only batch --ms $m --proofs $lg --reps 3 2>&1 | grep "^batch" | cut -c1-110
done; done
} 2>&1 | tee $OUT/ab.txt
