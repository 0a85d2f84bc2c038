set -o pipefail
cd /root/repo
bash tools/profile_round.sh r02bk || exit 1
python bench.py --mode criterion > gpurun_out/r02bk_bench_mode_criterion.json 2> gpurun_out/r02bk_criterion.err || exit 1
python bench.py --mode verify > gpurun_out/r02bk_bench_mode_verify.json 2> gpurun_out/r02bk_verify.err || exit 1
python bench.py --mode build > gpurun_out/r02bk_bench_mode_build.json 2> gpurun_out/r02bk_build.err || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02bk_bench_n1_driver_args.json 2> gpurun_out/r02bk_driver.err || exit 1
tail -c 400 gpurun_out/r02bk_bench_n1_driver_args.json
