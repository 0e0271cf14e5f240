#!/bin/bash
# developer A/B: the segment kernel's time with the product library and with tagged developer builds (vgan_amd/build.py VGAN_BUILD_TAG)
for t in "" "$@"; do
  lib=$PWD/vgan_amd/lib/libvgan_gpu$t.so
  for i in 1 2; do
    VGAN_LIB=$lib python3 bench.py --mode node_weights --steps 30 --warmup 5 --cpu-seconds 0 --no-pmc --no-frontend 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('lib%-6s avg_launch_ms %.4f  frac %.4f  ms_per_step %.4f' % ('$t', r['avg_launch_ms'], r['frac'], d['ms_per_step']))"
  done
done
