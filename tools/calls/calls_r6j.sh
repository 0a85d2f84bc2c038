#!/bin/bash
# round 6, call j: do the two lanes of a small call overlap on the device?  kernel trace of one height-24 splitting proof
set -o pipefail
export DAPOL_ENV_KNOBS=1 TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/r6j; mkdir -p $OUT
python3 tools/lanes_timeline.py | tail -4
DAPOL_NO_LANES=1 python3 tools/lanes_timeline.py | tail -2
GPU_MAX_HW_QUEUES=8 python3 tools/lanes_timeline.py | tail -2
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/tools/lanes_timeline.py > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r6j/trace/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# last call: from the last k_tree_find_leaves / path walk on
starts=[i for i,r in enumerate(rows) if "k_gather_parties" in r["Kernel_Name"]]
seg=rows[starts[-2]:]           # the two gathers of the last call
t0=int(seg[0]["Start_Timestamp"])
qs=sorted(set(r["Queue_Id"] for r in seg))
print("queues in the last call:", qs, " columns:", list(rows[0].keys())[:12])
for r in seg[:70]:
    print("%8.3f +%7.3f ms q%-3s %s"%((int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,qs.index(r["Queue_Id"]),r["Kernel_Name"][:50]))
PY
rm -rf $OUT/trace
