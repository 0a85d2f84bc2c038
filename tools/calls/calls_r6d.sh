#!/bin/bash
# round 6, call d: PMC passes of the individual-proof batch (k_rp_msm_gs_hi) and of the tree build, a timeline of the tree build,
# the new bench legs, the two-rank bench flow
set -o pipefail
export DAPOL_ENV_KNOBS=1 TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out/r6d; mkdir -p $OUT
cd /tmp
# --- all-kernel PMC table of the m = 1 batch (2^17 proofs, one rep)
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/m1_pmc_$n -o pmc -- python3 $R/tools/bench_small_parties.py --only batch --ms 1 --reps 1 > $OUT/m1_pmc_$n.log 2>&1 || { tail -5 $OUT/m1_pmc_$n.log; exit 1; }
done
cd $R
python3 tools/pmc_all_summary.py $OUT/m1_pmc_FETCH_SIZE $OUT/m1_pmc_WRITE_SIZE $OUT/m1_pmc_SQ_INSTS_VALU > $OUT/m1_pmc_all_kernels.txt 2>&1 || tail -3 $OUT/m1_pmc_all_kernels.txt
python3 tools/pmc_summary.py $OUT/m1_pmc_*/ --kernel k_rp_msm_gs_hi --out $OUT/m1_msm_gs_hi_pmc.json > /dev/null 2>&1 || echo "pmc_summary failed"
for d in $OUT/m1_pmc_*/; do rm -rf $d; done
echo "m1 pmc done"
# --- the tree build: kernel trace with timestamps (timeline) and the all-kernel PMC table
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tree_trace -o t -- python3 $R/tools/bench_tree_only.py 20 3 > $OUT/tree_trace.log 2>&1 || { tail -5 $OUT/tree_trace.log; exit 1; }
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/tree_pmc_$n -o pmc -- python3 $R/tools/bench_tree_only.py 20 1 > $OUT/tree_pmc_$n.log 2>&1 || { tail -5 $OUT/tree_pmc_$n.log; exit 1; }
done
cd $R
python3 tools/pmc_all_summary.py $OUT/tree_pmc_*/ > $OUT/tree_pmc_all_kernels.txt 2>&1 || tail -3 $OUT/tree_pmc_all_kernels.txt
for d in $OUT/tree_pmc_*/; do rm -rf $d; done
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get("OUT","gpurun_out/r6d")+"/tree_trace/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last build: from the last k_tree_check_leaves on
starts=[i for i,r in enumerate(rows) if "k_tree_check_leaves" in r["Kernel_Name"]]
seg=rows[starts[-1]:]
t0=int(seg[0]["Start_Timestamp"]); busy=0; last_end=t0; gaps=0
out=[]
for r in seg:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if s>last_end: gaps+=s-last_end
    busy+=e-s; last_end=max(last_end,e)
    out.append("%9.3f ms  +%8.3f ms  %s"%((s-t0)/1e6,(e-s)/1e6,r["Kernel_Name"][:60]))
open(os.path.join(os.path.dirname(os.path.dirname(f)),"tree_timeline.txt"),"w").write("span %.3f ms, kernel time %.3f ms, idle gaps %.3f ms, %d launches\n"%((last_end-t0)/1e6,busy/1e6,gaps/1e6,len(seg))+"\n".join(out[:40])+"\n...\n"+"\n".join(out[-12:])+"\n")
PY
find $OUT/tree_trace -name "*kernel_trace.csv" -delete
head -3 $OUT/tree_timeline.txt
echo "tree done"
# --- the bench line with the new legs (small workload), verify mode, the two-rank flow
timeout -k 10 400 python3 bench.py --log2-entities 16 --steps 2 --warmup 1 --cpu-budget-s 3 > $OUT/bench_lg16.json 2> $OUT/bench_lg16.err || { tail -5 $OUT/bench_lg16.err; exit 1; }
tail -1 $OUT/bench_lg16.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['secondary'].get('small_parties'))[:1500]); print(d['value'], d['complete'])"
timeout -k 10 300 python3 bench.py --mode verify --steps 10 --warmup 2 > $OUT/bench_verify.json 2> $OUT/bench_verify.err || { tail -5 $OUT/bench_verify.err; exit 1; }
python3 -c "import json; d=json.loads(open('$OUT/bench_verify.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['combined_check_fallbacks_in_timed_region'], d['all_verified'], d['ms_per_step_pinned_host_buffers'])"
timeout -k 10 600 python3 -m pytest tests/test_bench_multirank_gpu.py -x -q > $OUT/multirank.log 2>&1 || { tail -30 $OUT/multirank.log; exit 1; }
tail -2 $OUT/multirank.log
