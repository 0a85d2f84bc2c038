#!/bin/bash
# round 5: kernel-trace stats of the 16-bit digit matrix against the 4-byte one (two builds, 2^18 entities, one pass each)
set -o pipefail
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for which in new base; do
  if [ $which = base ]; then export DAPOL_HIP_LIB=$R/build/libdapol_base.so; else unset DAPOL_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r12b_$which -o s -- python3 $R/bench.py --no-cpu-baseline --no-secondary --log2-entities 18 --steps 1 --warmup 1 > $OUT/r12b_$which.log 2>&1 || { tail -5 $OUT/r12b_$which.log; exit 1; }
  cp $(find $OUT/r12b_$which -name "*kernel_stats.csv" | head -1) $OUT/r12b_kernel_stats_$which.csv
  rm -rf $OUT/r12b_$which
done
cd $R
python3 - <<'PY'
import csv
def load(f):
    return {r["Name"].split("(")[0]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("gpurun_out/r12b_kernel_stats_new.csv"), load("gpurun_out/r12b_kernel_stats_base.csv")
tot_a, tot_b = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print("total kernel ms: new %.1f base %.1f" % (tot_a, tot_b))
for k in sorted(b, key=lambda k: -b[k][1])[:16]:
    if k in a:
        print("%-60s calls %6d  new %9.1f ms  base %9.1f ms  %+6.1f %%" % (k[-60:], b[k][0], a[k][1], b[k][1], 100 * (a[k][1] / b[k][1] - 1)))
PY
