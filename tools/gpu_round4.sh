#!/bin/bash
# round 4 closing session (GPU box): full GPU suite, smoke, the node-weights trace + FETCH/WRITE passes, the SQ passes of the
# segment kernel and of the euka kernel, the three bench lines (soibean also at 2 M reads).   VGAN_COMMIT names the tree.
export TMPDIR=/tmp
tag=${1:-round4_v2}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/${tag}_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/${tag}_smoke.log
bash tools/gpu_profile.sh ${tag}_node --steps 20 --warmup 5 > gpurun_out/${tag}_profile.log 2>&1
bash tools/gpu_pmc_wave.sh ${tag}_node_sq 1000000 150 5 > gpurun_out/${tag}_pmc.log 2>&1
bash tools/gpu_pmc.sh ${tag}_euka_sq --path euka > gpurun_out/${tag}_pmc_euka.log 2>&1
bash tools/gpu_profile.sh ${tag}_euka --path euka --steps 20 --warmup 5 > gpurun_out/${tag}_profile_euka.log 2>&1
timeout 900 python3 bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_default.json
timeout 900 python3 bench.py --path euka --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_euka.json
timeout 900 python3 bench.py --path soibean --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_soibean.json
timeout 900 python3 bench.py --path soibean --reads 2000000 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_soibean2m.json
cat gpurun_out/${tag}_pytest_gpu.log gpurun_out/${tag}_smoke.log
for f in default euka soibean soibean2m; do head -c 600 gpurun_out/${tag}_bench_$f.json; echo; done
