set -e
export DAPOL_ENV_KNOBS=1     # the DAPOL_* knobs below are read only by a process that opts in
cd /root/repo
out=gpurun_out/r02x_latency_sweep.txt
: > $out
python tools/bench_latency_one.py 12 >> $out 2>&1
for s in 32 64; do DAPOL_SMALL_SPLIT=$s python tools/bench_latency_one.py 12 >> $out 2>&1; done
for s in 16 32 64; do DAPOL_NO_TAIL=1 DAPOL_SMALL_SPLIT=$s python tools/bench_latency_one.py 12 >> $out 2>&1; done
DAPOL_NO_SMALL_HI=1 python tools/bench_latency_one.py 12 >> $out 2>&1
DAPOL_PATHS_LANE=1 python tools/bench_latency_one.py 12 >> $out 2>&1
cat $out
