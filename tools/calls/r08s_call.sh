#!/bin/bash
# round 5: randomised soaks on the final sources + the verifier's other shapes + configs[4] with the PMC traffic of the running build
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
( python3 tools/soak_gs.py 40 505 | tail -1; python3 tools/soak_small_calls.py 300 506 | tail -1; python3 tools/soak_tree_update.py 80 507 | tail -1 ) > $OUT/r08s_soaks.txt 2>&1; cat $OUT/r08s_soaks.txt
python3 tools/bench_verify.py 1024 1024 > $OUT/r08s_verify_only.txt 2>&1; python3 tools/bench_verify.py 65536 32 >> $OUT/r08s_verify_only.txt 2>&1; python3 tools/bench_verify_entities.py 16 >> $OUT/r08s_verify_only.txt 2>&1; tail -12 $OUT/r08s_verify_only.txt
python3 bench.py --mode verify --steps 30 --warmup 3 2>/dev/null | tail -1 > $OUT/r08s_bench_mode_verify.json; python3 -c "import json; d=json.load(open('$OUT/r08s_bench_mode_verify.json')); print(d['ms_per_step'], d['value'], d['roofline']['traffic'], d['roofline']['traffic_over_algorithmic'], d['cpu_baseline'])"
