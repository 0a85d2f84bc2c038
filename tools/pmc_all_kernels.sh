#!/bin/bash
export DAPOL_ENV_KNOBS=1
# PMC passes over EVERY prover kernel (not only the MSM): tools/pmc_all_kernels.sh <tag>.  One table: kernel, calls, avg ms,
# HBM read / write bytes per call (FETCH_SIZE x 2 / WRITE_SIZE, the guide's gfx950 correction), VALU instructions, wait share.
set -o pipefail
tag=${1:-r05}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
PMC_LG=${PMC_LG:-17}
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${tag}_pmcall_$n -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-secondary --log2-entities $PMC_LG --warmup 0 --steps 1 > $OUT/${tag}_pmcall_$n.log 2>&1 || { tail -5 $OUT/${tag}_pmcall_$n.log; exit 1; }
  echo "pmc $n done"
done
cd $R
python3 tools/pmc_all_summary.py $(ls -d $OUT/${tag}_pmcall_*/ ) > $OUT/${tag}_pmc_all_kernels.txt
# the raw CSVs are large (one row per dispatch and counter): keep the table only
for d in $(ls -d $OUT/${tag}_pmcall_*/); do rm -rf $d; done
cat $OUT/${tag}_pmc_all_kernels.txt
