#!/bin/bash
# Extra counter passes on the dominant kernel (stall attribution).  tools/pmc_extra.sh <tag>
R=$(pwd); OUT=$R/gpurun_out; export TMPDIR=/tmp
tag=${1:-x}
i=0
for c in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${tag}_px$i -o pmc -- python3 $R/bench.py --no-cpu-baseline --log2-entities 18 --warmup 0 > $OUT/${tag}_px$i.log 2>&1 ) || { tail -5 $OUT/${tag}_px$i.log; continue; }
  f=$OUT/${tag}_px$i/pmc_counter_collection.csv
  python3 - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "k_rp_msm<0, 4>" in n and int(r["Grid_Size"]) == 589824:
        acc[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-40s %.6g  (n=%d)" % (k, sum(v["v"]) / len(v["v"]), len(v["v"])))
PY
  rm -rf $OUT/${tag}_px$i
done
