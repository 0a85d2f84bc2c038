#!/bin/bash
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
# verification latency of b proofs per call: defaults, always batched, never batched
cd "$(dirname "$0")/.."
for cfg in "-" "DAPOL_VERIFY_RLC_MIN=2" "DAPOL_VERIFY_RLC_MIN=100000"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  env $e python tools/verify_small_sweep.py ${BS:-1 2 8 16 32 64 128 256 512} 2>&1 | grep verify
done
