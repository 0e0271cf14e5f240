#!/usr/bin/env python3
"""Mean per-dispatch value of every counter in the rocprofv3 --pmc passes under a directory, per kernel (vgan:: kernels
only).  Writes <dir>/summary.json and prints a table; with a profile name also profiles/<name>.json and the kernels' issue
figures into profiles/valu_issue.json.  usage: summarize_pmc.py <dir> [<profile name>]"""
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
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "vgan::" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
if not out and os.path.exists(os.path.join(d, "summary.json")):  # (the passes' CSVs were summarised on the GPU box and removed)
    out = json.load(open(os.path.join(d, "summary.json")))
else:
    json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1, sort_keys=True)
issue = {}
for k, cs in out.items():
    print("==", k)
    for c in sorted(cs):
        print("   %-32s %.4g" % (c, cs[c]))
    if "SQ_ACTIVE_INST_VALU" in cs and "SQ_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_* count quad-cycles summed over the chip's SIMDs (MI355X_MICROARCH.md "Per-instruction cycle constants")
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the launch = GUI_ACTIVE / 8, SIMD-cycles = that x 1024 SIMDs
        simd_cycles = cs["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        frac = cs["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles
        print("   VALU issue fraction (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)) = %.3f" % frac)
        issue[k] = {"valu_issue_frac": frac, "lds_busy_frac": cs.get("SQ_LDS_IDX_ACTIVE", 0) / (cs["GRBM_GUI_ACTIVE"] / 8.0 * 256.0),
                    "lds_bank_conflict_share": cs.get("SQ_LDS_BANK_CONFLICT", 0) / max(cs.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                    "valu_insts": cs.get("SQ_INSTS_VALU"), "salu_insts": cs.get("SQ_INSTS_SALU"), "lds_insts": cs.get("SQ_INSTS_LDS")}
if issue and len(sys.argv) > 2:  # summarize_pmc.py <dir> <profile name>: the committed figure bench.py quotes as *_from_profile
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = os.environ.get("VGAN_COMMIT")  # (the GPU box holds a snapshot without .git: the caller names the tree)
    if not commit:
        try:
            commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            commit = None
    # the counters themselves under profiles/<name>.json; valu_issue.json keeps the kernels of earlier profiles of the SAME tree
    json.dump(out, open(os.path.join(root, "profiles", sys.argv[2] + ".json"), "w"), indent=1, sort_keys=True)
    vi = os.path.join(root, "profiles", "valu_issue.json")
    names, kernels = [sys.argv[2]], dict(issue)
    try:
        old = json.load(open(vi))
        if old.get("commit") == commit and old.get("profile") != sys.argv[2]:
            names = [n for n in old["profile"].split(" + ") if n != sys.argv[2]] + names
            kernels = {**old.get("kernels", {}), **issue}
    except Exception:
        pass
    json.dump({"profile": " + ".join(names), "commit": commit, "kernels": kernels}, open(vi, "w"), indent=1)
