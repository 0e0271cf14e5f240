#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counter list>" <bench args...>   -- one rocprofv3 --pmc pass, CSV under gpurun_out/pmc_<tag>
tag=$1; shift
ctrs=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp
rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py "$@" --cpu-seconds 0 --no-parity --steps 2 --warmup 1 > $R/gpurun_out/pmc_$tag/bench.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc_$tag/*/*counter_collection.csv'):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row['Kernel_Name'].startswith('vgan::'):
            agg[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
    for k, v in sorted(agg.items()):
        print('%-40s %-28s n=%d mean=%.4g' % (k[0], k[1], len(v), sum(v)/len(v)))
PY
