#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command, and separate PMC passes for HBM traffic.
# usage: tools/gpu_profile.sh <tag> <bench args...>
set -x
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$tag
cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag/trace -- python3 $R/bench.py "$@" --cpu-seconds 0 --no-pmc --no-frontend --no-ingest > $R/gpurun_out/prof_$tag/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/pmc_fetch -- python3 $R/bench.py "$@" --cpu-seconds 0 --no-pmc --no-frontend --no-ingest --steps 3 --warmup 1 > $R/gpurun_out/prof_$tag/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$tag/pmc_write -- python3 $R/bench.py "$@" --cpu-seconds 0 --no-pmc --no-frontend --no-ingest --steps 3 --warmup 1 > $R/gpurun_out/prof_$tag/bench_pmc_write.log 2>&1
cd $R
find gpurun_out/prof_$tag -name "*.csv" | head -30
for f in $(find gpurun_out/prof_$tag/trace -name "*kernel_stats.csv"); do echo "== $f"; head -20 $f; done
