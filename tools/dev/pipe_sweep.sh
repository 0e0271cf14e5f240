#!/bin/bash
# GPU box: `vgan haplocart` on a 10 M-read GAM with the device front end at several (slots, piece size) settings
d=$(mktemp -d /tmp/vgan_ps_XXXX)
python3 - "$d" ${1:-10000000} <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
from vgan_amd import haplocart as hc
d, n = sys.argv[1], int(sys.argv[2])
g = hc.synth_graph()
g.write(d)
CH = 1000000
with open(d + "/r.gam", "wb") as f:
    for c0 in range(0, n, CH):
        a = hc.synth_reads(g, min(CH, n - c0), first_read=c0)
        a.write_gam(d + "/part.gam")
        blob = open(d + "/part.gam", "rb").read()
        f.write(blob[:-28] if c0 + CH < n else blob)
        del a
P
for cfg in "3 201326592" "4 150994944" "5 120795955" "6 100663296" "4 201326592" "2 268435456"; do
  set -- $cfg
  for rep in 1 2; do
    s=$(date +%s%N)
    VGAN_GAMPIPE_SLOTS=$1 VGAN_GAMPIPE_PIECE=$2 VGAN_TIMING=1 VGAN_HC_DEVICE_GAM=1 vgan_amd/bin/vgan haplocart -g $d/r.gam --hc-files $d -q -t -1 --keep-duplicates -o $d/o.tsv -pf $d/p.txt 2> $d/err.log > /dev/null
    e=$(date +%s%N)
    echo "slots $1 piece $2: wall $(( (e - s) / 1000000 )) ms; $(grep 'front end' $d/err.log | sed 's/.*pieces on/pieces on/' | cut -c1-60) ... $(grep 'front end' $d/err.log | grep -o '[0-9]* ms from start to finish ([0-9]* after' ) $(grep 'front end' $d/err.log | grep -o '[0-9.]* GB of device memory')"
  done
done
rm -rf $d
