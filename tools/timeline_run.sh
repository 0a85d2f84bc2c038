#!/bin/bash
export DAPOL_ENV_KNOBS=1
# tools/timeline_run.sh <tag> [log2 entities]: kernel traces of one proving pass with one and with two chunks in flight -> gap summaries
set -o pipefail
tag=${1:-tl}; lg=${2:-18}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for s in 2 1; do
  DAPOL_STREAMS=$s rocprofv3 --kernel-trace --output-format csv -d $OUT/${tag}_tl$s -o tl -- python3 $R/bench.py --no-cpu-baseline --no-secondary --log2-entities $lg --steps 1 --warmup 0 > $OUT/${tag}_tl$s.log 2>&1 || { tail -5 $OUT/${tag}_tl$s.log; exit 1; }
  f=$(find $OUT/${tag}_tl$s -name "*kernel_trace.csv" | head -1)
  { echo "## DAPOL_STREAMS=$s, 2^$lg entities"; tail -1 $OUT/${tag}_tl$s.log | grep -o '"value": [0-9.]*'; python3 $R/tools/timeline_gaps.py $f; } >> $OUT/${tag}_timeline_gaps.txt
  rm -rf $OUT/${tag}_tl$s
done
cat $OUT/${tag}_timeline_gaps.txt
