#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/t12.log
: > $out
timeout 900 python3 -m pytest tests/test_packed_gpu.py tests/test_hc_gpu.py tests/test_devflat_gpu.py tests/test_pyref_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $out
for rl in 150; do for rep in 1 2 3; do
python3 tools/wave_time.py 1000000 $rl 20 2>&1 | tail -1 | sed 's/sum(final).*//' >> $out
done; done
VGAN_LIB=$PWD/vgan_amd/lib/libvgan_gpu_sp.so python3 tools/wave_spans.py 1000000 150 2>&1 | tail -10 >> $out
cat $out
