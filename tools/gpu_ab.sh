#!/bin/bash
# developer aid (GPU box): the segment kernel's time under the product library and under tagged variant builds, interleaved
#   tools/gpu_ab.sh <read_len> <tag> [<tag> ...]      (variants built by tools/build_variant.sh)
mkdir -p gpurun_out
export TMPDIR=/tmp
rl=$1; shift
out=gpurun_out/ab_$rl.log
: > $out
for rep in 1 2; do
  echo "== default" >> $out; python3 tools/wave_time.py 1000000 $rl 20 2>&1 | tail -1 >> $out
  for t in "$@"; do
    echo "== $t" >> $out; VGAN_LIB=$PWD/vgan_amd/lib/libvgan_gpu_$t.so python3 tools/wave_time.py 1000000 $rl 20 2>&1 | tail -1 >> $out
  done
done
cat $out
