#!/bin/bash
# round 5: s_setprio on the transcript's latency chains, squeezes by v_readlane, message words absorbed whole: parity tests, three runs of configs[4], timeline
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 -m pytest tests/test_gpu_parity.py -x -q > $OUT/r09f_verify_tests.txt 2>&1 || { tail -20 $OUT/r09f_verify_tests.txt; exit 1; }
tail -2 $OUT/r09f_verify_tests.txt
for i in 1 2 3; do
  python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('wordwise', d['ms_per_step'], d['value'], d['ms_per_step_pinned_host_buffers'], d['all_verified'], d['one_bad_proof_turns_the_job_verdict'])" | tee -a $OUT/r09f_verify_ab.txt
done
bash tools/r08f_call.sh > /dev/null 2>&1; cp $OUT/r08f_verify_timeline.txt $OUT/r09f_verify_timeline.txt; cat $OUT/r09f_verify_timeline.txt
