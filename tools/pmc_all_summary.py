#!/usr/bin/env python3
"""Per-kernel table from rocprofv3 --pmc passes (tools/pmc_all_kernels.sh): every kernel of the run, largest grid of each."""
import collections
import csv
import glob
import os
import re
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))      # (kernel, grid) -> counter -> values
    for d in sys.argv[1:]:
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not cc or not kt:
            continue
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
        seen = set()
        for r in csv.DictReader(open(cc[0])):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("dapol::", "").replace("void ", "")
            key = (k, int(r["Grid_Size"]))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                agg[key]["_ns"].append(dur[r["Dispatch_Id"]])
    rows = []
    for (k, g), c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        calls = len(c["_ns"]) / max(1, len(sys.argv) - 1)
        rows.append((m["_ns"] * calls, k, g, calls, m))
    rows.sort(reverse=True)
    print("%-34s %10s %7s %9s %9s %9s %10s %6s %6s" % ("kernel", "grid", "calls", "avg ms", "rd MB", "wr MB", "VALU Minst", "GB/s", "wait%"))
    for tot, k, g, calls, m in rows[:40]:
        ms = m["_ns"] / 1e6
        rd = m.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6
        wr = m.get("WRITE_SIZE", 0) * 1024 / 1e6
        wait = 100.0 * m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0
        print("%-34s %10d %7.1f %9.3f %9.1f %9.1f %10.2f %6.0f %6.1f" % (k[:34], g, calls, ms, rd, wr, m.get("SQ_INSTS_VALU", 0) / 1e6, (rd + wr) / ms if ms else 0, wait))


if __name__ == "__main__":
    main()
