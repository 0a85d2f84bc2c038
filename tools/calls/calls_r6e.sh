#!/bin/bash
# round 6, call e: the tree merge with the padding children in a launch of their own (DAPOL_TREE_SPLIT=1) against the fused kernel
set -o pipefail
export DAPOL_ENV_KNOBS=1
OUT=gpurun_out/r6e; mkdir -p $OUT
for i in 1 2; do
  echo "fused:"; python3 tools/bench_tree_only.py 20 4 | tail -3
  echo "split:"; DAPOL_TREE_SPLIT=1 python3 tools/bench_tree_only.py 20 4 | tail -3
done 2>&1 | tee $OUT/tree_split_ab.txt
DAPOL_TREE_SPLIT=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tree or build or merge or padding or shard" > $OUT/tree_split_tests.log 2>&1 || { tail -20 $OUT/tree_split_tests.log; exit 1; }
tail -2 $OUT/tree_split_tests.log
