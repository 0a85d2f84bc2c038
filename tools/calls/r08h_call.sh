#!/bin/bash
# round 5: commitments uploaded in column blocks under the phased transcript replay: verdict tests, A/B against one copy up front, timeline
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_parity.py -x -q -k "verif or rlc or bucket or cancelling or config4 or batching or transcript" > $OUT/r08i_verify_tests.txt 2>&1; tail -4 $OUT/r08i_verify_tests.txt
for i in 1 2 3; do
  python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined_kept_io', d['ms_per_step'], d['value'], d['all_verified'], d['one_bad_proof_turns_the_job_verdict'])" | tee -a $OUT/r08i_verify_ab.txt
  DAPOL_ENV_KNOBS=1 DAPOL_VERIFY_NO_PIPELINE=1 python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_copy_kept_io', d['ms_per_step'], d['value'], d['all_verified'])" | tee -a $OUT/r08i_verify_ab.txt
done
bash tools/r08f_call.sh > /dev/null 2>&1; cp $OUT/r08f_verify_timeline.txt $OUT/r08i_verify_timeline.txt; cat $OUT/r08i_verify_timeline.txt
