#!/bin/bash
# round 5, end-of-round evidence part B (after tools/install_pmc_profile.py r08): configs[4] profile, the driver's command, the one-GPU
# shares of the strong-scaling series + preflight (-> PROJECTED table), the N > 1 flow with 4 ranks over gloo, a power / clock trace
set -o pipefail
OUT=gpurun_out; mkdir -p $OUT
bash tools/profile_verify.sh r08 > $OUT/r08z_profile_verify.log 2>&1; tail -24 $OUT/r08z_profile_verify.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r08_bench_n1_driver_args.json 2> $OUT/r08_bench_n1_driver_args.err; tail -3 $OUT/r08_bench_n1_driver_args.err
tail -1 $OUT/r08_bench_n1_driver_args.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver', d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['wall_s_since_process_start'], d['complete'], d['roofline']['traffic'], d['roofline']['frac'])"
for lg in 20 19 18 17; do
  python3 bench.py --log2-entities $lg --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r08z_share_2e$lg.json
done
python3 bench.py --preflight > $OUT/r08z_preflight_n1.json 2> $OUT/r08z_preflight_n1.err
python3 tools/projected_scaling.py $OUT/r08z_share_2e20.json $OUT/r08z_share_2e19.json $OUT/r08z_share_2e18.json $OUT/r08z_share_2e17.json $OUT/r08z_preflight_n1.json > $OUT/r08_projected_scaling.md; cat $OUT/r08_projected_scaling.md
DAPOL_BENCH_BACKEND=gloo DAPOL_TABLE_GB=3 python3 bench.py --gpus 4 --log2-entities 12 --height 20 --steps 2 --warmup 1 --cpu-budget-s 3 2>/dev/null | tail -1 > $OUT/r08z_gloo4_rehearsal.json
python3 -c "import json; d=json.load(open('$OUT/r08z_gloo4_rehearsal.json')); print('gloo4', d['n_gpus'], d['value'], d['multi_gpu']['step_ms'], d['parity'])"
DAPOL_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --preflight 2>/dev/null | tail -1 > $OUT/r08z_preflight_n4_gloo.json; python3 -c "import json; d=json.load(open('$OUT/r08z_preflight_n4_gloo.json')); print('preflight4', d['ok'], d['checks'])"
bash tools/smi_watch.sh $OUT/r08z_power_clock_raw.txt python3 bench.py --log2-entities 18 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('smi-run', d['value'])"
grep -c sclk $OUT/r08z_power_clock_raw.txt
