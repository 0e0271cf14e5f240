#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/t10_shapes.log
: > $out
for rl in 40 75 150; do
  echo "== default $rl" >> $out; python3 tools/wave_time.py 1000000 $rl 20 2>&1 | tail -1 >> $out
  echo "== nr12 $rl" >> $out; VGAN_LIB=$PWD/vgan_amd/lib/libvgan_gpu_nr12.so python3 tools/wave_time.py 1000000 $rl 20 2>&1 | tail -1 >> $out
done
cat $out
