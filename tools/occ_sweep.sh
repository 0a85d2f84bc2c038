cd /root/repo
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
for cfg in "DAPOL_LPL=32" "DAPOL_LPL=32 DAPOL_MSM_OCC_CAP=3" "DAPOL_LPL=32 DAPOL_MSM_OCC_CAP=2" "DAPOL_LPL=32 DAPOL_MSM_OCC_CAP=1" "DAPOL_LPL=16" "DAPOL_LPL=16 DAPOL_MSM_OCC_CAP=1"; do
  echo "4096 [$cfg] $(env $cfg python tools/bench_midsize_one.py 4096 2>&1 | cut -d' ' -f2 | tr '\n' ' ')"
done
