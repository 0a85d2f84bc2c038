#!/bin/bash
# round 6, end-of-round evidence part B2 (second half of part B: a gpurun lease is at most 20 minutes): the few-party regime (kernel
# stats + PMC on the final sources), the tree after the split (PMC + runs), --mode build, preflight with one RCCL rank, the N > 1
# flow with 4 ranks over gloo (inline preflight included)
set -o pipefail
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export DAPOL_ENV_KNOBS=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r15z_m1 -o stats -- python3 $R/tools/bench_small_parties.py --only batch --ms 1 --reps 2 > $OUT/r15z_m1.log 2>&1 || { tail -5 $OUT/r15z_m1.log; exit 1; }
cp $(find $OUT/r15z_m1 -name "*kernel_stats.csv" | head -1) $OUT/r15_m1_kernel_stats.csv; rm -rf $OUT/r15z_m1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r15z_agg24 -o stats -- python3 $R/tools/bench_small_parties.py --only policy --aggs 24 --reps 2 > $OUT/r15z_agg24.log 2>&1 || { tail -5 $OUT/r15z_agg24.log; exit 1; }
cp $(find $OUT/r15z_agg24 -name "*kernel_stats.csv" | head -1) $OUT/r15_agg24_kernel_stats.csv; rm -rf $OUT/r15z_agg24
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/r15z_m1pmc_$n -o pmc -- python3 $R/tools/bench_small_parties.py --only batch --ms 1 --reps 1 > $OUT/r15z_m1pmc_$n.log 2>&1 || { tail -5 $OUT/r15z_m1pmc_$n.log; exit 1; }
done
cd $R
python3 tools/pmc_all_summary.py $OUT/r15z_m1pmc_FETCH_SIZE $OUT/r15z_m1pmc_WRITE_SIZE $OUT/r15z_m1pmc_SQ_INSTS_VALU > $OUT/r15_m1_pmc_all_kernels.txt 2>&1
python3 tools/pmc_summary.py $OUT/r15z_m1pmc_*/ --kernel k_rp_msm_gs_hi --out $OUT/r15_m1_msm_gs_hi_pmc.json > /dev/null 2>&1 || echo "pmc_summary failed"
for d in $OUT/r15z_m1pmc_*/; do rm -rf $d; done
echo "few-party evidence done"
python3 tools/bench_small_parties.py --reps 2 > $OUT/r15_small_parties.txt 2>&1; grep -v "^{" $OUT/r15_small_parties.txt | cut -c1-200
python3 tools/bench_small_parties.py --only verify --aggs 32,24,8,0 --reps 3 2>&1 | grep "^verify" > $OUT/r15_small_parties_verify.txt; cat $OUT/r15_small_parties_verify.txt
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/r15z_treepmc_$n -o pmc -- python3 $R/tools/bench_tree_only.py 20 1 > $OUT/r15z_treepmc_$n.log 2>&1 || { tail -5 $OUT/r15z_treepmc_$n.log; exit 1; }
done
cd $R
python3 tools/pmc_all_summary.py $OUT/r15z_treepmc_*/ > $OUT/r15_tree_pmc_all_kernels.txt 2>&1
for d in $OUT/r15z_treepmc_*/; do rm -rf $d; done
python3 tools/bench_tree_only.py 20 4 > $OUT/r15_tree_only.txt 2>&1; tail -3 $OUT/r15_tree_only.txt
python3 bench.py --mode build > $OUT/r15z_bench_mode_build.json 2> /dev/null; tail -c 400 $OUT/r15z_bench_mode_build.json; echo
python3 bench.py --preflight > $OUT/r15z_preflight_n1.json 2> $OUT/r15z_preflight_n1.err
DAPOL_BENCH_BACKEND=gloo DAPOL_TABLE_GB=3 python3 bench.py --gpus 4 --log2-entities 12 --height 20 --steps 2 --warmup 1 --cpu-budget-s 3 2>/dev/null | tail -1 > $OUT/r15z_gloo4_rehearsal.json
python3 -c "import json; d=json.load(open('$OUT/r15z_gloo4_rehearsal.json')); print('gloo4', d['n_gpus'], d['value'], d['multi_gpu']['step_ms'], d['parity'], d['multi_gpu']['preflight']['ok'], d['multi_gpu']['preflight']['seconds'])"
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/r15y_smoke.txt 2>&1; tail -1 $OUT/r15y_smoke.txt
python3 bench.py --mode verify --steps 20 > $OUT/r15y_bench_mode_verify.json 2> /dev/null
python3 -c "import json; d=json.loads(open('$OUT/r15y_bench_mode_verify.json').read().strip().split('\n')[-1]); print('verify', d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_over_algorithmic'], d['combined_check_fallbacks_in_timed_region'])"
