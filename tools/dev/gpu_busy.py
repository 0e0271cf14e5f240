#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv): for the last `window` seconds of kernel activity, the fraction of time at least one kernel ran and
the sum of kernel durations by name (python3 tools/dev/gpu_busy.py <kernel_trace.csv> [window_s])."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 0.27
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t1 = max(e for _, e, _ in iv)
t0 = t1 - int(win * 1e9)
iv = [(max(s, t0), e, n) for s, e, n in iv if e > t0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for s, e, n in iv:
    tot[n[:40]] = tot.get(n[:40], 0) + (e - s)
print("window %.0f ms: some kernel running %.0f ms (%.0f %%), kernel durations summed %.0f ms" % (win * 1e3, busy / 1e6, 100.0 * busy / (win * 1e9), sum(tot.values()) / 1e6))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print("  %-40s %.1f ms" % (n, v / 1e6))
