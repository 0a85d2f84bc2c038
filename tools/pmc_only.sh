#!/bin/bash
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
# The --pmc passes of profile_round.sh alone: tools/pmc_only.sh <tag>   (KERNEL, PMC_LG and DAPOL_* from the environment)
set -o pipefail
tag=${1:-r03}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
KERNEL=${KERNEL:-k_rp_msm_gs}
PMC_LG=${PMC_LG:-17}
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${tag}_pmc_$n -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-secondary --log2-entities $PMC_LG --warmup 0 --steps 1 > $OUT/${tag}_pmc_$n.log 2>&1 || { tail -5 $OUT/${tag}_pmc_$n.log; exit 1; }
  echo "pmc $n done"
  for f in $(find $OUT/${tag}_pmc_$n -name "*counter_collection.csv"); do head -1 $f > $f.tmp; grep "k_rp_msm" $f >> $f.tmp; mv $f.tmp $f; done
  for f in $(find $OUT/${tag}_pmc_$n -name "*kernel_trace.csv"); do head -1 $f > $f.tmp; grep "k_rp_msm" $f >> $f.tmp; mv $f.tmp $f; done
done
cd $R
python3 tools/pmc_summary.py $(ls -d $OUT/${tag}_pmc_*/ ) --kernel "$KERNEL" --out $OUT/${tag}_msm_pmc.json
