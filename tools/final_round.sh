set -o pipefail
cd /root/repo
bash tools/profile_round.sh r02bo || exit 1
python bench.py --mode criterion > gpurun_out/r02bo_bench_mode_criterion.json 2> gpurun_out/r02bo_criterion.err || exit 1
python bench.py --mode verify > gpurun_out/r02bo_bench_mode_verify.json 2> gpurun_out/r02bo_verify.err || exit 1
python bench.py --mode build > gpurun_out/r02bo_bench_mode_build.json 2> gpurun_out/r02bo_build.err || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02bo_bench_n1_driver_args.json 2> gpurun_out/r02bo_driver.err || exit 1
tail -c 400 gpurun_out/r02bo_bench_n1_driver_args.json
