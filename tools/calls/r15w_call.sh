#!/bin/bash
# round 6, final sources: randomised soaks (the few-party paths of round 6; the round-5 soaks again)
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
{
timeout -k 10 500 python3 tools/soak_small_parties.py 80 606 | tail -4
timeout -k 10 300 python3 tools/soak_gs.py 10 2027 | tail -2
timeout -k 10 300 python3 tools/soak_small_calls.py 100 2027 | tail -2
timeout -k 10 300 python3 tools/soak_tree_update.py 30 2027 | tail -2
} 2>&1 | tee $OUT/r15w_soaks.txt
