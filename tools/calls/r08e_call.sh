#!/bin/bash
# round 5, GPU call E: the whole GPU suite on the new build; bench vs host profile, interleaved; 2^20 beside the 2^19 share (projected scaling)
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 -m pytest tests -m gpu -q -x > $OUT/r08e_pytest_gpu.txt 2>&1; tail -6 $OUT/r08e_pytest_gpu.txt
for i in 1 2; do
  for prof in bench host; do
    python3 bench.py --profile $prof --log2-entities 19 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$prof', d['value'], d['ms_per_step'], c['window_bits'], c['high_half_rows'], round(c['device_memory_in_use_gb'],1), d['checksum'])" | tee -a $OUT/r08e_profile_ab.txt
  done
done
python3 bench.py --log2-entities 20 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r08e_bench_2e20.json
python3 -c "import json; d=json.load(open('$OUT/r08e_bench_2e20.json')); print('2^20', d['value'], d['ms_per_step'], d['config']['device_memory_in_use_gb'])"
