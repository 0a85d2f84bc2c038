#!/bin/bash
# round 5: the doubling-free generator MSM of the batched verifier: verdict tests, A/B against the Straus kernel, timeline
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_parity.py -x -q -k "verif or rlc or bucket or cancelling or config4 or batching or transcript" > $OUT/r08g_verify_tests.txt 2>&1; tail -4 $OUT/r08g_verify_tests.txt
for i in 1 2 3; do
  python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('gen_sweep', d['ms_per_step'], d['value'], d['all_verified'], d['one_bad_proof_turns_the_job_verdict'])" | tee -a $OUT/r08g_verify_ab.txt
  DAPOL_ENV_KNOBS=1 DAPOL_VERIFY_GEN_STRAUS=1 python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('straus', d['ms_per_step'], d['value'], d['all_verified'])" | tee -a $OUT/r08g_verify_ab.txt
done
bash tools/r08f_call.sh > /dev/null 2>&1; cp $OUT/r08f_verify_timeline.txt $OUT/r08g_verify_timeline.txt; cat $OUT/r08g_verify_timeline.txt
