#!/bin/bash
# round 6, call h: grouped verification of few-party proofs
set -o pipefail
export DAPOL_ENV_KNOBS=1
OUT=gpurun_out/r6h; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_small_parties.py -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
{
echo "== grouped (default)"; python3 tools/bench_small_parties.py --only verify --aggs 32,24,8,0 --reps 3 2>&1 | grep "^verify"
echo "== one check per sub-proof (DAPOL_NO_GROUP=1)"; DAPOL_NO_GROUP=1 python3 tools/bench_small_parties.py --only verify --aggs 24,8,0 --reps 3 2>&1 | grep "^verify"
} | tee $OUT/verify_grouped_ab.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q -k "verif or batch or wire or blake2b" > $OUT/verify_tests.log 2>&1 || { tail -30 $OUT/verify_tests.log; exit 1; }
tail -2 $OUT/verify_tests.log
