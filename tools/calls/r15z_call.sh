#!/bin/bash
# round 6, end-of-round evidence part B1 (after tools/install_pmc_profile.py r15): the driver's command, configs[4] profile
# (part B2: r15y_call.sh)
set -o pipefail
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r15_bench_n1_driver_args.json 2> $OUT/r15_bench_n1_driver_args.err; tail -3 $OUT/r15_bench_n1_driver_args.err
tail -1 $OUT/r15_bench_n1_driver_args.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver', d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['wall_s_since_process_start'], d['complete'], d['roofline']['traffic'], d['roofline']['frac'], d['phases_ms'])"
bash tools/profile_verify.sh r15 > $OUT/r15z_profile_verify.log 2>&1; tail -24 $OUT/r15z_profile_verify.log
