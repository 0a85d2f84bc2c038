#!/bin/bash
export DAPOL_ENV_KNOBS=1
# Evidence for BASELINE configs[4] (verification only, 1,024 proofs x 1,024 parties; benches/dapol.rs:93-141 of the reference):
#   tools/profile_verify.sh <tag>      (from the repo root, on the GPU box; one gpurun call)
#   gpurun_out/<tag>_verify_bench.json        python bench.py --mode verify --steps 20
#   gpurun_out/<tag>_verify_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command (the k_rv* rows are the verifier; the k_rp* rows
#                                             are the untimed setup that makes the proofs)
#   gpurun_out/<tag>_verify_pmc.txt / .json   separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ / GRBM) over the verifier's kernels
set -o pipefail
tag=${1:-r07}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 bench.py --mode verify --steps 20 > $OUT/${tag}_verify_bench.json 2> $OUT/${tag}_verify_bench.err || { tail -5 $OUT/${tag}_verify_bench.err; exit 1; }
tail -c 700 $OUT/${tag}_verify_bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_vstats -o stats -- python3 $R/bench.py --mode verify --no-cpu-baseline --no-bad-proof-leg --no-pinned-leg --warmup 1 --steps 20 > $OUT/${tag}_vstats.log 2>&1 || { tail -5 $OUT/${tag}_vstats.log; exit 1; }
cp $(find $OUT/${tag}_vstats -name "*kernel_stats.csv" | head -1) $OUT/${tag}_verify_kernel_stats.csv
rm -rf $OUT/${tag}_vstats
echo "stats done"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${tag}_vpmc_$n -o pmc -- python3 $R/bench.py --mode verify --no-cpu-baseline --no-bad-proof-leg --no-pinned-leg --warmup 2 --steps 5 > $OUT/${tag}_vpmc_$n.log 2>&1 || { tail -5 $OUT/${tag}_vpmc_$n.log; exit 1; }
  echo "pmc $n done"
done
cd $R
# 7 verification passes per run: 2 warm-ups (one per alternating batch) + 5 timed (the untimed pass with a bad proof, whose per-proof re-check is another path, is skipped)
python3 tools/pmc_all_summary.py --only k_rv --passes 7 --json $OUT/${tag}_verify_pmc.json $(ls -d $OUT/${tag}_vpmc_*/ ) > $OUT/${tag}_verify_pmc.txt
for d in $(ls -d $OUT/${tag}_vpmc_*/); do rm -rf $d; done
cat $OUT/${tag}_verify_pmc.txt
