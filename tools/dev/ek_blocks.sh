#!/bin/bash
# GPU box: the euka bench line over the number of workgroups (VGAN_EUKA_BLOCKS, a developer aid of launch_euka_reads)
for nblk in "$@"; do
  for i in 1 2; do
    VGAN_EUKA_BLOCKS=$nblk timeout 600 python3 bench.py --path euka --steps 30 --warmup 5 --cpu-seconds 0 --no-pmc 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('blocks $nblk', round(d['value']/1e9,3), 'G reads/s', round(d['ms_per_step'],4), 'ms/step, kernel', round(d['roofline']['avg_launch_ms'],4))"
  done
done
