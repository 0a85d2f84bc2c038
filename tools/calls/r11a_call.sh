#!/bin/bash
# round 5: the stream's closed form shared with the host test (hash.h), the permutation without a loop in k_rv_absorb_V:
# ubenches, parity tests, three runs of configs[4]
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
./build/ubench_absorb 1024 1024 > $OUT/r11a_absorb_ubench.txt 2>&1 && ./build/ubench_absorb 1024 256 >> $OUT/r11a_absorb_ubench.txt 2>&1 || { cat $OUT/r11a_absorb_ubench.txt; exit 1; }
cat $OUT/r11a_absorb_ubench.txt
./build/ubench_keccak 1024 > $OUT/r11a_keccak_ubench.txt 2>&1 || { cat $OUT/r11a_keccak_ubench.txt; exit 1; }
tail -2 $OUT/r11a_keccak_ubench.txt
python3 -m pytest tests/test_gpu_parity.py -x -q > $OUT/r11a_parity_tests.txt 2>&1 || { tail -30 $OUT/r11a_parity_tests.txt; exit 1; }
tail -2 $OUT/r11a_parity_tests.txt
python3 tools/soak_verify_wave.py 60 612 | tail -1
for i in 1 2 3; do
  python3 bench.py --mode verify --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('unrolled', d['ms_per_step'], d['value'], d['ms_per_step_pinned_host_buffers'], d['all_verified'], d['one_bad_proof_turns_the_job_verdict'])" | tee -a $OUT/r11a_verify_ab.txt
done
