#!/bin/bash
# GPU box: the euka bench line for several variant builds (tools/dev/ek_ab.sh "" _bg16), three runs each
for t in "$@"; do
  for i in 1 2 3; do
    VGAN_LIB=$PWD/vgan_amd/lib/libvgan_gpu$t.so timeout 600 python3 bench.py --path euka --steps 30 --warmup 5 --cpu-seconds 0 --no-pmc 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('variant [$t]', round(d['value']/1e9,3), 'G reads/s', round(d['ms_per_step'],4), 'ms/step, kernel', round(d['roofline']['avg_launch_ms'],4))"
  done
done
