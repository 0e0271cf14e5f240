#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/t9_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/t9_smoke.log
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/t9_bench_default.json
timeout 600 python bench.py --path euka 2>&1 | tail -1 > gpurun_out/t9_bench_euka.json
timeout 600 python bench.py --path soibean 2>&1 | tail -1 > gpurun_out/t9_bench_soibean.json
