#!/bin/bash
# round 6, final sources: the whole GPU suite, smoke(), configs[4] with the PMC traffic of the running build
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/r12y_pytest_gpu.txt 2>&1; tail -3 $OUT/r12y_pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/r12y_smoke.txt 2>&1; tail -1 $OUT/r12y_smoke.txt
python3 bench.py --mode verify --steps 20 > $OUT/r12y_bench_mode_verify.json 2> /dev/null
python3 -c "import json; d=json.loads(open('$OUT/r12y_bench_mode_verify.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_over_algorithmic'], d['combined_check_fallbacks_in_timed_region'])"
