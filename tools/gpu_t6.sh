#!/bin/bash
# round 4 profile session: the default bench line under rocprofv3 (kernel trace + FETCH/WRITE passes) and the SQ counter passes
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_profile.sh round4_v1_node --steps 20 --warmup 5 > gpurun_out/t6_profile.log 2>&1
bash tools/gpu_pmc_wave.sh round4_v1_node_sq 1000000 150 5 > gpurun_out/t6_pmc.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_round4_v1_node_sq round4_v1_node_sq > /dev/null 2>&1
cp profiles/valu_issue.json gpurun_out/valu_issue_round4_v1.json
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/t6_bench.json
tail -5 gpurun_out/t6_pmc.log
