#!/bin/bash
# round 6, call b: per-kernel split of the small-party regime
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out/r6b; mkdir -p $OUT
export TMPDIR=/tmp DAPOL_ENV_KNOBS=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/m1 -o stats -- python3 $R/tools/bench_small_parties.py --only batch --ms 1 --reps 2 > $OUT/m1.log 2>&1 || { tail -5 $OUT/m1.log; exit 1; }
find $OUT/m1 -name "*kernel_trace.csv" -delete
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/agg24 -o stats -- python3 $R/tools/bench_small_parties.py --only policy --aggs 24 --reps 2 > $OUT/agg24.log 2>&1 || { tail -5 $OUT/agg24.log; exit 1; }
find $OUT/agg24 -name "*kernel_trace.csv" -delete
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/m8 -o stats -- python3 $R/tools/bench_small_parties.py --only batch --ms 8 --reps 2 > $OUT/m8.log 2>&1 || { tail -5 $OUT/m8.log; exit 1; }
find $OUT/m8 -name "*kernel_trace.csv" -delete
find $OUT -name "*kernel_stats.csv" | head
