#!/bin/bash
# round 6, refresh of the end-of-round evidence after the lanes change (kernel-source hash changed: host code only, but the hash covers csrc/):
# the whole GPU suite, then part A (tools/final_round.sh r15)
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/r15_pytest_gpu.txt 2>&1; tail -3 $OUT/r15_pytest_gpu.txt
grep -q " passed" $OUT/r15_pytest_gpu.txt && ! grep -q "failed" $OUT/r15_pytest_gpu.txt || exit 1
bash tools/final_round.sh r15
