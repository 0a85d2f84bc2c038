#!/bin/bash
# round 5, GPU call B: preflight (1 rank over RCCL, 2 ranks over gloo), the sweep levers in the micro-benchmark, the per-GPU shares of the
# strong-scaling series on one GPU (2^19 / 2^18 / 2^17 entities), the multi-rank bench tests
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
python3 bench.py --preflight > $OUT/r08b_preflight_n1.json 2> $OUT/r08b_preflight_n1.err || { echo "preflight n1 failed"; tail -20 $OUT/r08b_preflight_n1.err; }
DAPOL_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --preflight > $OUT/r08b_preflight_n2_gloo.json 2> $OUT/r08b_preflight_n2_gloo.err || { echo "preflight n2 failed"; tail -20 $OUT/r08b_preflight_n2_gloo.err; }
echo "preflight done"; cat $OUT/r08b_preflight_n1.json
python3 tools/ubench_levers.py > $OUT/r08b_levers.txt 2> $OUT/r08b_levers.err; cat $OUT/r08b_levers.txt
for lg in 19 18 17; do
  python3 bench.py --log2-entities $lg --steps 3 --warmup 1 --no-secondary --cpu-budget-s 4 > $OUT/r08b_bench_2e$lg.json 2> $OUT/r08b_bench_2e$lg.err || echo "bench 2^$lg failed"
  tail -1 $OUT/r08b_bench_2e$lg.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['entities_total'], d['value'], d['ms_per_step'], d['complete'])"
done
python3 -m pytest tests/test_bench_multirank_gpu.py -x -q > $OUT/r08b_pytest_multirank.txt 2>&1; tail -3 $OUT/r08b_pytest_multirank.txt
