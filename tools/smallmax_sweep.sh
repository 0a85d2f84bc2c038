cd /root/repo
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
for b in 512 1024 1025 1536 2048 3072 4096; do
  for cfg in "-" "DAPOL_SMALL_MAX=4096"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    echo "$b [$cfg] $(env $e python tools/bench_midsize_one.py $b 2>&1 | cut -d' ' -f2 | tr '\n' ' ')"
  done
done
