#!/bin/bash
# round 4, first GPU session: the packed route's tests, the wave kernel's, one default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_packed_gpu.py tests/test_wave_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/t1_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -2 | tee gpurun_out/t1_bench.log
