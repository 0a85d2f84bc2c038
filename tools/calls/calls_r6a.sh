#!/bin/bash
# round 6, call a: new small-party tests + the regime's first numbers (old path via knobs vs new defaults)
set -o pipefail
mkdir -p gpurun_out/r6a
timeout -k 10 900 python -m pytest tests/test_gpu_small_parties.py -x -q > gpurun_out/r6a/tests.log 2>&1 || { tail -30 gpurun_out/r6a/tests.log; exit 1; }
tail -3 gpurun_out/r6a/tests.log
echo "== old path (no grouping, no short-list sweep)"
DAPOL_NO_GROUP=1 DAPOL_GS_SMALL_MIN=1000000000 timeout -k 10 300 python tools/bench_small_parties.py --reps 1 --aggs 32,24,0 --ms 1,8 --proofs 15 > gpurun_out/r6a/old.log 2>&1 || { tail -20 gpurun_out/r6a/old.log; exit 1; }
grep -v "^{" gpurun_out/r6a/old.log
echo "== grouped only"
DAPOL_GS_SMALL_MIN=1000000000 timeout -k 10 300 python tools/bench_small_parties.py --reps 1 --aggs 24,0 --only policy > gpurun_out/r6a/grouped.log 2>&1 || { tail -20 gpurun_out/r6a/grouped.log; exit 1; }
grep -v "^{" gpurun_out/r6a/grouped.log
echo "== new defaults"
timeout -k 10 300 python tools/bench_small_parties.py --reps 2 > gpurun_out/r6a/new.log 2>&1 || { tail -20 gpurun_out/r6a/new.log; exit 1; }
grep -v "^{" gpurun_out/r6a/new.log
