#!/bin/bash
# round 5: four-lane doubling chains at the end of the window sums + half-wavefront fixed-base products in k_rvb_finish:
# verdict tests, three runs of the configs[4] verifier and of the 65,536 x 32-party batch, kernel timeline
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_parity.py -x -q -k "verif or rlc or bucket or cancelling or config4 or batching or transcript" > $OUT/r09a_verify_tests.txt 2>&1 || { tail -20 $OUT/r09a_verify_tests.txt; exit 1; }
tail -2 $OUT/r09a_verify_tests.txt
for i in 1 2 3; do
  python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('quad_chain', d['ms_per_step'], d['value'], d['all_verified'], d['one_bad_proof_turns_the_job_verdict'])" | tee -a $OUT/r09a_verify_ab.txt
done
bash tools/r08f_call.sh > /dev/null 2>&1; cp $OUT/r08f_verify_timeline.txt $OUT/r09a_verify_timeline.txt; cat $OUT/r09a_verify_timeline.txt
