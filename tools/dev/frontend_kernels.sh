#!/bin/bash
# GPU box: bash tools/dev/frontend_kernels.sh [n_reads] -> per-kernel standalone averages of the device front end on one piece
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
n=${1:-500000}
rm -rf /tmp/fk; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fk -- python3 $R/tools/dev/frontend_kernels.py $n 2>&1 | grep "reads:"
python3 - <<'P'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/fk/*/*kernel_stats.csv")[0])))
for r in rows:
    if float(r["TotalDurationNs"]) / int(r["Calls"]) > 30000 or "vgan" in r["Name"]:
        print("%-48s calls %4s  avg %9.1f us" % (r["Name"][:48], r["Calls"], float(r["AverageNs"]) / 1e3))
P
