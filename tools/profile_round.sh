#!/bin/bash
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
# One GPU call that refreshes the evidence under profiles/: tools/profile_round.sh <tag>   (run from the repo root on the GPU box)
#   gpurun_out/<tag>_ubench_madd.txt      VALU-only cost of the point operations (the MSM's issue roof)
#   gpurun_out/<tag>_bench.json           python bench.py (default workload, with the CPU baseline)
#   gpurun_out/<tag>_stats/               rocprofv3 --kernel-trace --stats of the same command without the CPU leg
#   gpurun_out/<tag>_pmc_*/               separate --pmc passes on 2^18 entities (two full chunks of 131,072 proofs)
set -o pipefail
tag=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
[ -x build/ubench_madd ] && build/ubench_madd > $OUT/${tag}_ubench_madd.txt 2>&1
python3 bench.py --steps 3 > $OUT/${tag}_bench.json 2> $OUT/${tag}_bench.err || { tail -5 $OUT/${tag}_bench.err; exit 1; }
tail -c 600 $OUT/${tag}_bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $OUT/${tag}_stats.log 2>&1 || { tail -5 $OUT/${tag}_stats.log; exit 1; }
find $OUT/${tag}_stats -name "*kernel_trace.csv" -delete          # (tens of MB; the merged gpurun_out/ is capped at 64 MiB)
echo "stats done"
# (one chunk in flight is the default since round 4: the trace above IS the serial split of the step; two chunks in flight for comparison)
DAPOL_STREAMS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_stats2 -o stats -- python3 $R/bench.py --no-cpu-baseline --no-secondary --log2-entities 18 --steps 1 --warmup 0 > $OUT/${tag}_stats2.log 2>&1 || { tail -5 $OUT/${tag}_stats2.log; exit 1; }
find $OUT/${tag}_stats2 -name "*kernel_trace.csv" -delete
echo "two-stream stats done"
# the dominant kernel: the generator-stationary sweep since round 3 (KERNEL="k_rp_msm<0" with DAPOL_GS=0 for the proof-stationary one)
KERNEL=${KERNEL:-k_rp_msm_gs}
PMC_LG=${PMC_LG:-18}
[ -n "$SKIP_BENCH" ] || true
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${tag}_pmc_$n -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-secondary --log2-entities $PMC_LG --warmup 0 --steps 1 > $OUT/${tag}_pmc_$n.log 2>&1 || { tail -5 $OUT/${tag}_pmc_$n.log; exit 1; }
  echo "pmc $n done"
  # keep only the rows of the dominant kernel (the merged gpurun_out/ is capped at 64 MiB)
  for f in $(find $OUT/${tag}_pmc_$n -name "*counter_collection.csv") $(find $OUT/${tag}_pmc_$n -name "*kernel_trace.csv"); do head -1 $f > $f.tmp; grep "k_rp_msm" $f >> $f.tmp; mv $f.tmp $f; done
done
cd $R
python3 tools/pmc_summary.py $(ls -d $OUT/${tag}_pmc_*/ ) --kernel "$KERNEL" --out $OUT/${tag}_msm_pmc.json
