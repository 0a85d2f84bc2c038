#!/usr/bin/env python3
"""Per-kernel table from rocprofv3 --pmc passes (tools/pmc_all_kernels.sh, tools/profile_verify.sh): every kernel of the run, by grid.
  pmc_all_summary.py [--only PREFIX] [--passes N] [--json OUT] <pass dir> ...
--only keeps the kernels whose name starts with PREFIX (k_rv: the verifier); --passes N divides the call counts by N (passes of
the hot path per run) and adds per-pass totals; --json writes the rows plus the totals and the hash of the kernel sources."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys


def kernel_src_sha():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "dapol_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith(".") or not os.path.isfile(os.path.join(d, f)):
            continue                                  # (as bench.kernel_src_sha: sources only)
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    args = sys.argv[1:]
    only, passes, jout = None, 1, None
    while args and args[0].startswith("--"):
        if args[0] == "--only":
            only = args[1]
        elif args[0] == "--passes":
            passes = int(args[1])
        elif args[0] == "--json":
            jout = args[1]
        args = args[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))      # (kernel, grid) -> counter -> values
    ndirs = 0
    for d in args:
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not cc or not kt:
            continue
        ndirs += 1
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
        seen = set()
        for r in csv.DictReader(open(cc[0])):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("dapol::", "").replace("void ", "")
            if only and not k.startswith(only):
                continue
            key = (k, int(r["Grid_Size"]))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                agg[key]["_ns"].append(dur[r["Dispatch_Id"]])
    rows = []
    for (k, g), c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        calls = len(c["_ns"]) / max(1, ndirs) / passes
        rows.append((m["_ns"] * calls, k, g, calls, m))
    rows.sort(reverse=True)
    print("%-34s %10s %7s %9s %9s %9s %10s %6s %6s" % ("kernel", "grid", "calls", "avg ms", "rd MB", "wr MB", "VALU Minst", "GB/s", "wait%"))
    out, tot_ms, tot_rd, tot_wr, tot_valu = [], 0.0, 0.0, 0.0, 0.0
    for tot, k, g, calls, m in rows[:40]:
        ms = m["_ns"] / 1e6
        rd = m.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6          # KiB; the guide's gfx950 correction: FETCH_SIZE reports half of the bytes of wide coalesced reads
        wr = m.get("WRITE_SIZE", 0) * 1024 / 1e6
        wait = 100.0 * m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0
        print("%-34s %10d %7.1f %9.3f %9.1f %9.1f %10.2f %6.0f %6.1f" % (k[:34], g, calls, ms, rd, wr, m.get("SQ_INSTS_VALU", 0) / 1e6, (rd + wr) / ms if ms else 0, wait))
        out.append({"kernel": k, "grid": g, "calls_per_pass": calls, "avg_ms": ms, "read_MB": rd, "write_MB": wr, "valu_Minst": m.get("SQ_INSTS_VALU", 0) / 1e6,
                    "GBps": (rd + wr) / ms if ms else 0, "wait_share": wait / 100.0, "grbm_gui_active": m.get("GRBM_GUI_ACTIVE")})
        tot_ms += ms * calls
        tot_rd += rd * calls
        tot_wr += wr * calls
        tot_valu += m.get("SQ_INSTS_VALU", 0) / 1e6 * calls
    if passes > 1 or only:
        print("per pass: %.3f ms of kernel time (serialised by the counter collection), %.1f MB read + %.1f MB written, %.1f M VALU instructions" % (tot_ms, tot_rd, tot_wr, tot_valu))
    if jout:
        dom = out[0] if out else None
        json.dump({"kernel_src_sha": kernel_src_sha(), "only": only, "passes_per_run": passes, "kernels": out,
                   "per_pass": {"kernel_ms_serialised": tot_ms, "hbm_read_MB": tot_rd, "hbm_write_MB": tot_wr, "valu_Minst": tot_valu},
                   "dominant_kernel": dom and dom["kernel"], "dominant_share_of_kernel_time": (dom["avg_ms"] * dom["calls_per_pass"] / tot_ms) if dom and tot_ms else None,
                   "traffic_note": "FETCH_SIZE x 2 + WRITE_SIZE (KiB), the guide's gfx950 correction; L2<->fabric boundary (Infinity-Cache hits included)"},
                  open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
