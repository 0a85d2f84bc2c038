#!/usr/bin/env python3
"""Idle time of the GPU inside a proving pass, from a rocprofv3 --kernel-trace CSV: the union of the kernel intervals against the
span from the first proving kernel to the last, the gaps by the kernel that precedes them, and how much of the busy time has one /
two / more kernels in flight.   python tools/timeline_gaps.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("dapol::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    # the proving pass: from the first k_rp_ kernel to the last
    rp = [x for x in rows if x[2].startswith("k_rp_")]
    t0, t1 = rp[0][0], max(x[1] for x in rp)
    ev = [x for x in rows if x[0] >= t0 and x[1] <= t1]
    span = (t1 - t0) / 1e6
    busy, cur_end, gaps = 0, t0, collections.Counter()
    gapn = collections.Counter()
    last = None
    for s, e, n in ev:
        if s > cur_end:
            gaps[last] += s - cur_end
            gapn[last] += 1
            busy += 0
            cur_s = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            last = n
    print("span %.1f ms, busy (union) %.1f ms = %.1f %%, sum of kernel durations %.1f ms" % (span, busy / 1e6, 100 * busy / 1e6 / span, sum(e - s for s, e, _ in ev) / 1e6))
    # depth histogram
    pts = []
    for s, e, n in ev:
        pts.append((s, 1)); pts.append((e, -1))
    pts.sort()
    depth, prev, hist = 0, t0, collections.Counter()
    for t, d in pts:
        hist[depth] += t - prev
        prev = t
        depth += d
    print("kernels in flight: " + ", ".join("%d: %.1f %%" % (k, 100 * v / (t1 - t0)) for k, v in sorted(hist.items())))
    print("idle after kernel (ms total, count, us each):")
    for n, v in gaps.most_common(12):
        print("  %-32s %9.2f %6d %8.1f" % (n, v / 1e6, gapn[n], v / 1e3 / gapn[n]))
    by = collections.Counter()
    for s, e, n in ev:
        by[n] += e - s
    print("kernel time by name (ms):")
    for n, v in by.most_common(14):
        print("  %-32s %9.1f" % (n, v / 1e6))


if __name__ == "__main__":
    main()
