#!/bin/bash
# round 5 closing session (GPU box): full GPU suite, smoke, the node-weights trace + FETCH/WRITE passes, SQ passes of the segment
# kernel, the length sweep, the three bench lines (soibean also at 2 M reads) and a trace + counter passes of the soibean refresh.
export TMPDIR=/tmp
tag=${1:-round5_v1}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/${tag}_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/${tag}_smoke.log
bash tools/gpu_profile.sh ${tag}_node --steps 20 --warmup 5 > gpurun_out/${tag}_profile.log 2>&1
bash tools/gpu_pmc_wave.sh ${tag}_node_sq 1000000 150 5 > gpurun_out/${tag}_pmc.log 2>&1
bash tools/gpu_profile.sh ${tag}_soibean --path soibean --reads 2000000 --steps 20 --warmup 5 > gpurun_out/${tag}_profile_soibean.log 2>&1
python3 tools/len_sweep.py 2>&1 | grep read_len > gpurun_out/${tag}_len_sweep.jsonl
timeout 900 python3 bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_default.json
timeout 900 python3 bench.py --path euka --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_euka.json
timeout 900 python3 bench.py --path soibean --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_soibean.json
timeout 900 python3 bench.py --path soibean --reads 2000000 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_soibean2m.json
cat gpurun_out/${tag}_pytest_gpu.log gpurun_out/${tag}_smoke.log gpurun_out/${tag}_len_sweep.jsonl
for f in default euka soibean soibean2m; do head -c 700 gpurun_out/${tag}_bench_$f.json; echo; done
