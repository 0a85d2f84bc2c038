cd /root/repo
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
for i in 1 2 3; do
  for b in 2048 8192; do
    for cfg in "-" "DAPOL_LPL=16" "DAPOL_LPL=8"; do
      if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
      echo "$b [$cfg] $(env $e python tools/bench_midsize_one.py $b 2>&1 | cut -d' ' -f2 | tr '\n' ' ')"
    done
  done
done
