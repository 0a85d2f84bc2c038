#!/bin/bash
# round 5, final sources: the whole GPU suite, smoke(), randomised soaks, the verifier's other shapes, configs[4] with the PMC traffic of the running build
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 -m pytest tests/ -x -q -m gpu > $OUT/r10y_pytest_gpu.txt 2>&1 || { tail -30 $OUT/r10y_pytest_gpu.txt; exit 1; }
tail -2 $OUT/r10y_pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r10y_smoke.txt 2>&1 || { tail -10 $OUT/r10y_smoke.txt; exit 1; }
tail -1 $OUT/r10y_smoke.txt
( python3 tools/soak_gs.py 40 605 | tail -1; python3 tools/soak_small_calls.py 300 606 | tail -1; python3 tools/soak_tree_update.py 80 607 | tail -1 ) > $OUT/r10y_soaks.txt 2>&1; cat $OUT/r10y_soaks.txt
python3 tools/bench_verify.py 1024 1024 > $OUT/r10y_verify_only.jsonl 2>&1 && python3 tools/bench_verify.py 65536 32 >> $OUT/r10y_verify_only.jsonl 2>&1 && python3 tools/bench_verify_entities.py 16 >> $OUT/r10y_verify_only.jsonl 2>&1; tail -6 $OUT/r10y_verify_only.jsonl | cut -c1-300
python3 bench.py --mode verify --steps 30 --warmup 3 2>/dev/null | tail -1 > $OUT/r10y_bench_mode_verify.json; python3 -c "import json; d=json.load(open('$OUT/r10y_bench_mode_verify.json')); print(d['ms_per_step'], d['value'], d['ms_per_step_pinned_host_buffers'], d['roofline']['traffic'], d['roofline']['traffic_over_algorithmic'], d['cpu_baseline'])"
