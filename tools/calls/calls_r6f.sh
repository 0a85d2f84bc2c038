#!/bin/bash
# round 6, call f: the whole GPU suite on the new sources, the sweep's crossover for 2 / 4 / 8 parties, the default bench line
set -o pipefail
export DAPOL_ENV_KNOBS=1
OUT=gpurun_out/r6f; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
for m in 2 4 8; do
  for lg in 10 11 12 13; do
    echo "== m=$m 2^$lg: default | forced sweep"
    timeout -k 10 120 python3 tools/bench_small_parties.py --only batch --ms $m --proofs $lg --reps 3 2>&1 | grep "^batch" | cut -c1-110
    DAPOL_GS_SMALL_MIN=64 timeout -k 10 120 python3 tools/bench_small_parties.py --only batch --ms $m --proofs $lg --reps 3 2>&1 | grep "^batch" | cut -c1-110
  done
done 2>&1 | tee $OUT/midsize_small_parties.txt
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
tail -1 $OUT/bench_default.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['steps'], d['ms_per_step'], d['phases_ms'], d['complete'], d['wall_s_since_process_start'])"
