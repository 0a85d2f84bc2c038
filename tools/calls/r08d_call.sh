#!/bin/bash
# round 5, GPU call D: tape modes + Blake2b tests, configs[4] (verification only) with the forked own-point branch and the aggregated counting sort
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_tape.py tests/test_gpu_blake2b.py -x -q > $OUT/r08d_new_tests.txt 2>&1; tail -12 $OUT/r08d_new_tests.txt
python3 -m pytest tests/test_gpu_parity.py -x -q -k "verif or rlc or bucket or cancelling or config4 or batching or transcript" > $OUT/r08d_verify_tests.txt 2>&1; tail -4 $OUT/r08d_verify_tests.txt
for i in 1 2; do
  python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('forked', d['ms_per_step'], d['value'], d['all_verified'], d['one_bad_proof_turns_the_job_verdict'])" | tee -a $OUT/r08d_verify_ab.txt
  DAPOL_ENV_KNOBS=1 DAPOL_VERIFY_ONE_STREAM=1 python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_stream', d['ms_per_step'], d['value'], d['all_verified'])" | tee -a $OUT/r08d_verify_ab.txt
done
