#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_devflat_gpu.py tests/test_hc_gpu.py -x -q -k "devflat or device_flatten or cli_chunk_loop" 2>&1 | tail -4 | tee gpurun_out/t8_pytest.log
timeout 1500 python3 tools/e2e_timeline.py 10000000 3 > gpurun_out/t8_e2e_10m.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-pmc --no-extra 2>&1 | tail -1 > gpurun_out/t8_bench.json
