#!/bin/bash
# End-of-round evidence, part A (one gpurun call, ~15 min):  bash tools/final_round.sh <tag>
#   profile_round.sh (bench + kernel-trace stats + PMC passes of the dominant kernel) and bench.py's other modes.
# Part B is tools/profile_verify.sh <tag> + the driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5) in a second call,
# after tools/install_pmc_profile.py <tag> has put this call's PMC summary under profiles/ (bench.py quotes it by source hash).
set -o pipefail
tag=${1:-r07}
cd "$(dirname "$0")/.."
bash tools/profile_round.sh $tag || exit 1
python3 bench.py --mode criterion > gpurun_out/${tag}_bench_mode_criterion.json 2> gpurun_out/${tag}_criterion.err || { tail -5 gpurun_out/${tag}_criterion.err; exit 1; }
python3 bench.py --mode build > gpurun_out/${tag}_bench_mode_build.json 2> gpurun_out/${tag}_build.err || { tail -5 gpurun_out/${tag}_build.err; exit 1; }
tail -c 300 gpurun_out/${tag}_bench_mode_criterion.json
