#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_devflat_gpu.py -x -q 2>&1 | tail -25 | tee gpurun_out/t7_pytest.log
