#!/usr/bin/env python3
"""Round-4 verdict item 2: the two untried levers on the generator-stationary sweep, in the micro-benchmark with real-sized tables
(tools/ubench_msm_order.hip; lanes and tile as the product runs them: 131,072 proofs x 15 windows, 16 rows per launch):
  v   the product's kernel shape: 123 VGPRs, four wavefronts per SIMD, 128-byte entries
  w   RESIDENCY: the same loop held to 96 VGPRs = five wavefronts per SIMD
  p   BYTES: 96-byte packed entries (3 x 255 bits; six 16-byte loads + ~51 shift / mask operations to unpack)
Per variant: SIMD time per wavefront-addition and shader clock; L2 hit rate, fabric bytes per addition, SQ_WAIT_INST_ANY share of
the wave-cycles (separate rocprofv3 --pmc passes).  Run on the GPU box from the repo root:
  python3 tools/ubench_levers.py > gpurun_out/<tag>_levers.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ubench_rowsync import run_plain, run_pmc  # noqa: E402


def main():
    W, nwin = 17, 15
    lanes = 131072 * nwin
    print("# W = 17, 4,096 rows of 65,537 entries; %d lanes (131,072 proofs x 15 windows), tiles of 16 rows" % lanes)
    print("%-34s %8s %7s %8s %13s %10s" % ("variant", "ns/add", "GHz", "L2 hit", "fabric B/add", "wait-inst"))
    for label, v in (("v  4 waves/SIMD, 128-B entries", "v"), ("w  5 waves/SIMD (96 VGPRs)", "w"), ("p  96-B packed entries", "p"),
                     ("v  (again: spread of the box)", "v")):
        spec = "gst:%d:16:%s" % (lanes, v)
        rows = run_plain(W, spec, 3.0)
        if not rows:
            print("%-34s (no result)" % label)
            continue
        r = rows[0]
        hm = run_pmc(W, spec, ["TCC_HIT_sum", "TCC_MISS_sum"])
        fs = run_pmc(W, spec, ["FETCH_SIZE"])
        sq = run_pmc(W, spec, ["SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"])
        hit = hm.get("TCC_HIT_sum", 0) / max(1.0, hm.get("TCC_HIT_sum", 0) + hm.get("TCC_MISS_sum", 0)) if hm else float("nan")
        fabric = fs.get("FETCH_SIZE", 0) * 1024 * 2 / (r["lanes"] * r["tile"]) if fs else float("nan")
        wait = sq.get("SQ_WAIT_INST_ANY", 0) / max(1.0, sq.get("SQ_WAVE_CYCLES", 0)) if sq else float("nan")
        print("%-34s %8.1f %7.3f %7.1f%% %13.1f %9.1f%%" % (label, r["ns"], r["ghz"], 100 * hit, fabric, 100 * wait))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
