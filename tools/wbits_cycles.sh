#!/bin/bash
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
# Cycles per table addition at different window widths (is the gap to the VALU-only cost caused by the HBM gathers?):
# rocprofv3 kernel stats of one chunk at a time + rocm-smi clock samples.   tools/wbits_cycles.sh <W> [<W> ...]
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp; export DAPOL_STREAMS=1
for W in "$@"; do
  export DAPOL_WBITS=$W
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 1; done ) > $OUT/smi_w$W.txt &
  wp=$!
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wb_$W -o s -- python3 $R/bench.py --no-cpu-baseline --log2-entities 18 --warmup 0 > $OUT/wb_$W.log 2>&1 )
  kill $wp
  echo "== W=$W"; grep "k_rp_msm" $OUT/wb_$W/s_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-160
  rm -f $OUT/wb_$W/s_kernel_trace.csv
done
