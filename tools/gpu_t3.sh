#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/dev/census.bin 2>&1 | tee gpurun_out/t3_census.log
timeout 1500 python -m pytest tests/test_sb_gpu.py tests/test_pyref_gpu.py tests/test_packed_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/t3_pytest.log
timeout 300 python bench.py --path soibean --steps 20 --warmup 3 --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/t3_bench_sb.log
