#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_sb_gpu.py tests/test_wave_gpu.py tests/test_packed_gpu.py tests/test_hc_gpu.py tests/test_distributed_gpu.py tests/test_pyref_gpu.py -x -q 2>&1 | tail -12 | tee gpurun_out/t5_pytest.log
