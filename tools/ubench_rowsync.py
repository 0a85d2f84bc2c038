#!/usr/bin/env python3
"""Round-3 verdict item 3: one more point on the table-width curve.  Windows of 15 / 16 / 17 bits under the generator-stationary
order, with the schedules that keep the wavefronts of a launch on the same one or two table rows ("row-synchronous": a row of
2.1 MB at 15 bits fits the 4 MB L2 of an XCD, 4.2 MB at 16 bits about does, 8.4 MB at 17 bits does not):
  gs        a chip's worth of lanes, accumulators in registers for the whole sweep: every wavefront walks the rows in step
  gst:T     the product's schedule: 65,536 x nwin lanes, tiles of T rows per launch, accumulators carried in HBM (T = 16 is the product)
Per (W, schedule): SIMD time per wavefront-addition and shader clock (tools/ubench_msm_order.hip), L2 hit rate and fabric bytes per
addition (separate rocprofv3 --pmc passes), and the end-to-end cost per 253-bit scalar = ns per addition x windows.
Run on the GPU box from the repo root:  python3 tools/ubench_rowsync.py > gpurun_out/<tag>_rowsync.txt"""
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "ubench_msm_order")
OUT = os.path.join(ROOT, "gpurun_out", "rowsync_tmp")


def run_plain(W, spec, secs):
    out = subprocess.run([BIN, str(W), "4096", spec, str(secs)], capture_output=True, text=True, timeout=600).stdout
    rows = []
    for line in out.splitlines():
        m = re.search(r"^(\S+)\s+lanes=(\d+) x (\d+) acc, tile=(\d+)\s+([\d.]+) ms per list sweep.*?([\d.]+) ns SIMD time per wave-addition\s+clock ([\d.]+) GHz", line)
        if m:
            rows.append(dict(name=m.group(1), lanes=int(m.group(2)), tile=int(m.group(4)), ms=float(m.group(5)), ns=float(m.group(6)), ghz=float(m.group(7))))
    return rows


def run_pmc(W, spec, counters):
    shutil.rmtree(OUT, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", OUT, "-o", "p", "--", BIN, str(W), "4096", spec, "0.15"],
                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    acc = {}
    for f in glob.glob(os.path.join(OUT, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gs" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    shutil.rmtree(OUT, ignore_errors=True)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    print("# windows of W bits: nwin = 253 // W + 1 windows; row = (2^(W-1) + 1) x 128 B; 4,096 rows")
    print("%-3s %-5s %-10s %5s %9s %8s %7s %9s %12s %14s" % ("W", "nwin", "schedule", "tile", "row MB", "ns/add", "GHz", "L2 hit", "fabric B/add", "ns x nwin"))
    for W in (15, 16, 17):
        nwin = 253 // W + 1
        lanes = 65536 * nwin
        row_mb = ((1 << (W - 1)) + 1) * 128 / 1e6
        specs = [("gs", "gs")] + [("gst:%d" % t, "gst:%d:%d:v" % (lanes, t)) for t in (16, 8, 4, 2)]
        for label, spec in specs:
            rows = run_plain(W, spec, 1.5)
            if not rows:
                print("%-3d %-5d %-10s  (no result)" % (W, nwin, label))
                continue
            r = rows[0]
            hm = run_pmc(W, spec, ["TCC_HIT_sum", "TCC_MISS_sum"])
            fs = run_pmc(W, spec, ["FETCH_SIZE"])
            hit = hm.get("TCC_HIT_sum", 0) / max(1.0, hm.get("TCC_HIT_sum", 0) + hm.get("TCC_MISS_sum", 0)) if hm else float("nan")
            rows_per_launch = r["tile"] if r["tile"] else 2048
            adds = r["lanes"] * rows_per_launch                       # additions per launch
            fabric = fs.get("FETCH_SIZE", 0) * 1024 * 2 / adds if fs else float("nan")
            print("%-3d %-5d %-10s %5d %9.1f %8.1f %7.3f %8.1f%% %12.1f %14.0f" % (W, nwin, label, r["tile"], row_mb, r["ns"], r["ghz"], 100 * hit, fabric, r["ns"] * nwin))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
