#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/dev/census.bin 28000 30000 31000 31744 32000 32256 32512 32768 2>&1 | tee gpurun_out/t4_census.log
timeout 600 python -m pytest tests/test_packed_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/t4_pytest.log
