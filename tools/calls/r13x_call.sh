#!/bin/bash
# round 6, last sanity of the bench flows after the bench.py-only changes (two-rank flow over gloo; the default line: complete, full host leg, traffic from the running build)
set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_bench_multirank_gpu.py -q 2>&1 | tail -2
timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['complete'], d['secondary']['host_buffers']['full_workload'], d['wall_s_since_process_start'], d['roofline']['traffic'])"
