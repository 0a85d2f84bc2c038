#!/usr/bin/env python3
"""Prints the kernels of the LAST verification call in a rocprofv3 kernel trace (CSV): start offset and duration.
usage: tools/kernel_timeline.py <kernel_trace.csv> [first-kernel-substring] [occurrence from the end, default 1] [kernels to print]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "k_rv_transcript"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
count = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]][-occ]
start = idx - 1 if idx > 0 and "k_rv_absorb_V" in rows[idx - 1]["Kernel_Name"] else idx
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:start + count]:
    n = r["Kernel_Name"].split("(")[0].replace("void dapol::", "").replace("dapol::", "")
    print("%-44s start %9.3f ms  dur %9.3f ms" % (n[:44], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
