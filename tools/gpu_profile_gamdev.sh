#!/bin/bash
# GPU box: rocprofv3 kernel trace of `vgan haplocart` with the GAM front end on the device, on a synthetic GAM of N reads
#   bash tools/gpu_profile_gamdev.sh <tag> [n_reads]     -> gpurun_out/prof_<tag>/ (kernel stats), gpurun_out/<tag>.log (the CLI's timeline)
export TMPDIR=/tmp
tag=${1:-round5_gamdev}; n=${2:-10000000}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
d=$(mktemp -d /tmp/vgan_gd_XXXX)
python3 - "$d" "$n" <<'P'
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
print("GAM of %d reads: %.1f MB" % (n, os.path.getsize(d + "/r.gam") / 1e6))
P
cd /tmp
VGAN_KEEP_TEARDOWN=1 VGAN_TIMING=1 VGAN_HC_DEVICE_GAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- $GRAFT_REPO_ROOT/vgan_amd/bin/vgan haplocart -g $d/r.gam --hc-files $d -q -t -1 --keep-duplicates -o $d/o.tsv > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
cd $GRAFT_REPO_ROOT
grep "front end\|haplocart @" gpurun_out/$tag.log | tail -12
rm -rf $d
