#!/bin/bash
# GPU box: the soibean bench line for several variant builds (tools/dev/sb_ab.sh "" _warm), at 1M and 2M reads, three runs each
for n in 1000000 2000000; do
for t in "$@"; do
  for i in 1 2 3; do
    VGAN_LIB=$PWD/vgan_amd/lib/libvgan_gpu$t.so timeout 600 python3 bench.py --path soibean --reads $n --steps 50 --warmup 10 --cpu-seconds 0 --no-pmc 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('reads $n variant [$t]', round(d['ms_per_step'],4), 'ms/step, kernel', round(d['roofline']['avg_launch_ms'],4))"
  done
done
done
