#!/usr/bin/env python3
"""After `tools/pmc_only.sh <tag>` (+ build/ubench_madd > gpurun_out/<tag>_ubench_madd.txt in the same GPU call): copies the summaries into
profiles/ (msm_pmc.json, <tag>_msm_pmc.json, <tag>_ubench_madd.txt) and rewrites profiles/valu_roof.json from them.
usage: python tools/install_pmc_profile.py <tag>"""
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
g, p = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
pm = json.load(open(os.path.join(g, tag + "_msm_pmc.json")))
ub = open(os.path.join(g, tag + "_ubench_madd.txt")).read().splitlines()[0]
m = re.search(r"=\s+([0-9.]+) cycles at the measured ([0-9.]+) GHz", ub)
cyc, ghz = float(m.group(1)), float(m.group(2))
shutil.copy(os.path.join(g, tag + "_msm_pmc.json"), os.path.join(p, "msm_pmc.json"))
shutil.copy(os.path.join(g, tag + "_msm_pmc.json"), os.path.join(p, tag + "_msm_pmc.json"))
shutil.copy(os.path.join(g, tag + "_ubench_madd.txt"), os.path.join(p, tag + "_ubench_madd.txt"))
lanes_rows = pm["grid_size"] * 16 / 64                      # wavefront-additions per tile launch (16 rows)
roof = {"source": "tools/ubench_madd.hip (profiles/%s_ubench_madd.txt), tools/pmc_summary.py (profiles/%s_msm_pmc.json), same box, same gpurun call" % (tag, tag),
        "register_only_cycles_per_wave_add": cyc, "register_only_clock_GHz": ghz, "kernel": pm["kernel"], "kernel_clock_GHz": round(pm["effective_clock_GHz"], 3),
        "valu_instructions_per_add": 979, "valu_instructions_per_add_in_kernel": round(pm["SQ_INSTS_VALU"] / lanes_rows, 1),
        "register_only_cycles_per_valu_inst": round(cyc / 979, 3), "kernel_cycles_per_valu_inst_per_simd": round(pm["cycles_per_valu_inst_per_simd"], 3),
        "valu_issue_frac": round((cyc / 979) / pm["cycles_per_valu_inst_per_simd"], 3), "kernel_src_sha": pm["kernel_src_sha"]}
json.dump(roof, open(os.path.join(p, "valu_roof.json"), "w"), indent=1)
print(json.dumps(roof))
