#!/bin/bash
# round 5: the wavefront's Keccak permutation (constant at the top of the round, alignbit rho, two rounds per trip), the absorbed
# blocks assembled from three aligned words a block ahead, the transcript head on the wavefront: ubench, verdict tests, all tests
# that replay a transcript on a wavefront (small prover calls), three runs of the configs[4] verifier, kernel timeline
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
./build/ubench_keccak 1024 > $OUT/r09c_keccak_ubench.txt 2>&1 || { cat $OUT/r09c_keccak_ubench.txt; exit 1; }
cat $OUT/r09c_keccak_ubench.txt
python3 -m pytest tests/test_gpu_parity.py -x -q > $OUT/r09c_parity_tests.txt 2>&1 || { tail -30 $OUT/r09c_parity_tests.txt; exit 1; }
tail -2 $OUT/r09c_parity_tests.txt
for i in 1 2 3; do
  python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('keccak_r5', d['ms_per_step'], d['value'], d['ms_per_step_pinned_host_buffers'], d['all_verified'], d['one_bad_proof_turns_the_job_verdict'])" | tee -a $OUT/r09c_verify_ab.txt
done
bash tools/r08f_call.sh > /dev/null 2>&1; cp $OUT/r08f_verify_timeline.txt $OUT/r09c_verify_timeline.txt; cat $OUT/r09c_verify_timeline.txt
