#!/bin/bash
# One GPU-box session: parity tests, smoke, bench in each mode. Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -5
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_default.log
for mode in per_read per_read_dense; do
  timeout 600 python bench.py --mode $mode --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | tail -3 | tee gpurun_out/bench_$mode.log
done
