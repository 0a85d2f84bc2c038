#!/bin/bash
# round 5, GPU call: one verification pass as a kernel timeline (who overlaps whom after the fork)
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/r08f_vtrace -o t -- python3 $R/bench.py --mode verify --no-cpu-baseline --no-bad-proof-leg --no-pinned-leg --warmup 1 --steps 5 > $OUT/r08f_vtrace.log 2>&1 || { tail -5 $OUT/r08f_vtrace.log; exit 1; }
cd $R
f=$(find $OUT/r08f_vtrace -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_timeline.py $f k_rv_transcript 1 40 > $OUT/r08f_verify_timeline.txt
python3 - "$f" >> $OUT/r08f_verify_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tr = [i for i, r in enumerate(rows) if "k_rv_transcript" in r["Kernel_Name"]]          # one per pass (k_rv_absorb_V runs once per column block)
def first_absorb(before):
    i = before
    while i > 0 and not any(k in rows[i - 1]["Kernel_Name"] for k in ("k_rvb_verdicts", "copyBuffer")): i -= 1
    return next(j for j in range(i, before + 1) if "k_rv_absorb_V" in rows[j]["Kernel_Name"])
a, b = first_absorb(tr[-2]), first_absorb(tr[-1])
print("pass-to-pass (first absorb_V start to the next pass's): %.3f ms" % ((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6))
fin = [i for i in range(a, b) if "k_rvb_verdicts" in rows[i]["Kernel_Name"]][-1]
print("absorb_V start -> verdicts end: %.3f ms" % ((int(rows[fin]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6))
PY
rm -rf $OUT/r08f_vtrace
cat $OUT/r08f_verify_timeline.txt
