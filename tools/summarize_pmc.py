#!/usr/bin/env python3
"""Mean per-dispatch value of every counter in the rocprofv3 --pmc passes under a directory, per kernel (vgan:: kernels
only).  Writes <dir>/summary.json and prints a table.  usage: summarize_pmc.py <dir>"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "vgan::" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1, sort_keys=True)
for k, cs in out.items():
    print("==", k)
    for c in sorted(cs):
        print("   %-32s %.4g" % (c, cs[c]))
    if "SQ_ACTIVE_INST_VALU" in cs and "SQ_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_* count quad-cycles summed over the chip's SIMDs (MI355X_MICROARCH.md "Per-instruction cycle constants")
        simd_cycles = cs["GRBM_GUI_ACTIVE"] * 256 * 4
        print("   VALU issue fraction (ACTIVE_INST_VALU*4 / (GUI_ACTIVE*1024 SIMDs)) = %.3f" % (cs["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles))
