#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (one directory per pass, each with *_counter_collection.csv and
*_kernel_trace.csv) for the dominant kernel and writes profiles/msm_pmc.json, which bench.py reads for
`roofline.traffic`.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE [gpurun_out/pmc_SQ_INSTS_VALU ...] \
        --kernel k_rp_msm<false --out profiles/msm_pmc.json

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": bytes = FETCH_SIZE*1024 * 2 (gfx950 reports exactly half
of a 16-byte-per-lane read stream; the table lookups are eight dwordx4 loads per lane) + WRITE_SIZE*1024 (exact for
16-byte-per-lane stores).  The factor is cross-checked here against the kernel's known gather bytes.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def load(d):
    cc = glob.glob(os.path.join(d, "*counter_collection.csv"))
    kt = glob.glob(os.path.join(d, "*kernel_trace.csv"))
    if not cc or not kt:
        raise SystemExit("no rocprofv3 CSVs in " + d)
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return list(csv.DictReader(open(cc[0]))), dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", default="k_rp_msm<false")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    per_grid = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args.dirs:
        rows, dur = load(d)
        seen = set()
        for r in rows:
            if args.kernel not in r["Kernel_Name"]:
                continue
            g = int(r["Grid_Size"])
            per_grid[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                per_grid[g]["_dur_ns"].append(dur[r["Dispatch_Id"]])
    if not per_grid:
        raise SystemExit("kernel not found")
    g = max(per_grid)                      # the full-chunk launches
    c = {k: sum(v) / len(v) for k, v in per_grid[g].items()}
    out = {"kernel": args.kernel, "grid_size": g, "launches_sampled": len(per_grid[g]["_dur_ns"]), "avg_launch_ms": c["_dur_ns"] / 1e6}
    if "FETCH_SIZE" in c:
        out["FETCH_SIZE_KB"] = c["FETCH_SIZE"]
        out["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        out["WRITE_SIZE_KB"] = c["WRITE_SIZE"]
        out["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c:
        out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out.get("hbm_write_bytes_per_launch", 0)
        out["hbm_GBps"] = out["hbm_bytes_per_launch"] / (c["_dur_ns"])
    for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE",
              "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum"):
        if k in c:
            out[k] = c[k]
    if "GRBM_GUI_ACTIVE" in c:
        out["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / c["_dur_ns"]
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        out["cycles_per_valu_inst_per_simd"] = (c["GRBM_GUI_ACTIVE"] / 8) / (c["SQ_INSTS_VALU"] / 1024)
    if "TCC_EA0_RDREQ_sum" in c and c["TCC_EA0_RDREQ_sum"]:
        out["l2_read_miss_latency_cycles"] = c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"]      # fabric round trip of an L2 read miss (Infinity Cache or HBM)
    if "TCC_HIT_sum" in c:
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    try:                                     # identity of the kernel sources these counters were collected on (bench.py quotes the
        from bench import kernel_src_sha     # file under roofline.traffic only when it matches the build that is running)
        out["kernel_src_sha"] = kernel_src_sha()
    except Exception:
        pass
    print(json.dumps(out, indent=1))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
