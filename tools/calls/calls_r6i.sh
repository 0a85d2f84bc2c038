#!/bin/bash
# round 6, call i: the groups of a small call's plan on lanes of their own: tests, the reference's criterion cases before / after
set -o pipefail
export DAPOL_ENV_KNOBS=1
OUT=gpurun_out/r6i; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_small_parties.py tests/test_gpu_fault_paths.py tests/test_gpu_blake2b.py -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for i in 1 2; do
echo "== lanes (default)"; python3 bench.py --mode criterion 2>/dev/null | tail -1 | python3 -c "import sys,json; c=json.loads(sys.stdin.read()); print([(r['height'],r['policy'][:3],round(r['prove_ms'],2),round(r['verify_ms'],2)) for r in c['prove_verify']])"
echo "== one after the other (DAPOL_NO_LANES=1)"; DAPOL_NO_LANES=1 python3 bench.py --mode criterion 2>/dev/null | tail -1 | python3 -c "import sys,json; c=json.loads(sys.stdin.read()); print([(r['height'],r['policy'][:3],round(r['prove_ms'],2),round(r['verify_ms'],2)) for r in c['prove_verify']])"
done 2>&1 | tee $OUT/criterion_lanes_ab.txt
