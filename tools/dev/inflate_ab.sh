#!/bin/bash
# developer A/B of the inflate kernels: per-kernel averages on a 1 M-read file with the product library and tagged builds (bash tools/dev/inflate_ab.sh _ta _tb ...)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
for t in "" "$@"; do
  export VGAN_LIB=$R/vgan_amd/lib/libvgan_gpu$t.so
  rm -rf /tmp/pi; (cd /tmp && VGAN_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- python3 $R/tools/dev/inflate_time.py 1000000 2>&1 | grep "older kernel" | head -1)
  python3 - "$t" <<'P'
import csv, glob, sys
rows = {r["Name"].split("(")[0].split("::")[-1]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(glob.glob("/tmp/pi/*/*kernel_stats.csv")[0]))}
print("lib%-4s tokens %.2f ms, lzw %.2f ms, crc %.2f ms" % (sys.argv[1], rows.get("gd_tokens_kernel", 0), rows.get("gd_lzw_kernel", 0), rows.get("gd_crc_kernel", 0)))
P
done
