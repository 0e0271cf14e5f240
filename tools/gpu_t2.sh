#!/bin/bash
# kernel A/B session: every tagged library on the same resident packed batch (segment kernel ms + checksum)
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/t2_ab.log
: > $out
echo "== default (DIRECT)" >> $out; python3 tools/wave_time.py 1000000 150 30 2>&1 | tail -1 >> $out
echo "== default NO_DIRECT" >> $out; VGAN_WV_NO_DIRECT=1 python3 tools/wave_time.py 1000000 150 30 2>&1 | tail -1 >> $out
for lib in vgan_amd/lib/libvgan_gpu_*.so; do
  echo "== $lib" >> $out
  VGAN_LIB=$PWD/$lib python3 tools/wave_time.py 1000000 150 30 2>&1 | tail -1 >> $out
  echo "== $lib NO_DIRECT" >> $out
  VGAN_WV_NO_DIRECT=1 VGAN_LIB=$PWD/$lib python3 tools/wave_time.py 1000000 150 30 2>&1 | tail -1 >> $out
done
cat $out
