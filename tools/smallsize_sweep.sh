cd /root/repo
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
for b in 2 4 8 16 32 64; do
  for cfg in "-" "DAPOL_NO_QUAD=1" "DAPOL_SMALL_TAIL=1" "DAPOL_SMALL_TAIL=1 DAPOL_NO_QUAD=1" "DAPOL_FS_SHAPE=1" "DAPOL_SMALL_TAIL=1 DAPOL_NO_QUAD=1 DAPOL_FS_SHAPE=1"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    echo "$b [$cfg] $(env $e python tools/bench_midsize_one.py $b 2>&1 | tail -1)"
  done
done
