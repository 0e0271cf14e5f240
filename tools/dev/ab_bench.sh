#!/bin/bash
# GPU box: the default bench line's kernel time for several variant builds (tools/dev/ab_bench.sh "" _aux2 ...), three runs each
for t in "$@"; do
  for i in 1 2 3; do
    VGAN_LIB=$PWD/vgan_amd/lib/libvgan_gpu$t.so timeout 600 python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-pmc --no-frontend --no-ingest --no-extra 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('variant [$t]', round(d['value']/1e9,3), 'G reads/s', round(d['ms_per_step'],4), 'ms/step, kernel', round(d['roofline']['avg_launch_ms'],4))"
  done
done
