#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hc_gpu.py -x -q -k "cli_chunk_loop" 2>&1 | tail -4 | tee gpurun_out/t8_pytest.log
timeout 900 python3 tools/e2e_timeline.py 1000000 2 > gpurun_out/t8_e2e_1m.log 2>&1
timeout 1500 python3 tools/e2e_timeline.py 10000000 3 VGAN_DF_PAGEABLE=1 VGAN_HC_HOST_FLATTEN=1 > gpurun_out/t8_e2e_10m.log 2>&1
