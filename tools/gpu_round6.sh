#!/bin/bash
# round 6 closing session (GPU box): full GPU suite, smoke, the node-weights trace + FETCH/WRITE passes, SQ passes of the segment
# kernel, the length and class sweeps, the bench lines (the default one with its end_to_end record; euka; soibean at 1 M and 2 M reads),
# traces + counter passes of euka's and soibean's kernels, the inflate kernels alone (trace + SQ passes), `vgan haplocart` on a 10 M-read
# GAM under the kernel trace, `vgan euka` on a 5 M-read GAM and `vgan soibean` on a 2 M-read one.
export TMPDIR=/tmp
tag=${1:-round6_v1}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/${tag}_pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/${tag}_smoke.log
bash tools/gpu_profile.sh ${tag}_node --steps 20 --warmup 5 > gpurun_out/${tag}_profile.log 2>&1
bash tools/gpu_pmc_wave.sh ${tag}_node_sq 1000000 150 5 > gpurun_out/${tag}_pmc.log 2>&1
bash tools/gpu_profile.sh ${tag}_soibean --path soibean --reads 2000000 --steps 20 --warmup 5 > gpurun_out/${tag}_profile_soibean.log 2>&1
bash tools/gpu_profile.sh ${tag}_euka --path euka --steps 20 --warmup 5 > gpurun_out/${tag}_profile_euka.log 2>&1
python3 tools/len_sweep.py 2>&1 | grep read_len > gpurun_out/${tag}_len_sweep.jsonl
python3 tools/class_sweep.py 2>&1 | grep mappability_values > gpurun_out/${tag}_class_sweep.jsonl
t0=$(date +%s); timeout 1200 python3 bench.py 2>&1 | tail -1 > gpurun_out/${tag}_bench_default.json; echo "python3 bench.py (no flags): $(( $(date +%s) - t0 )) s of wall clock" > gpurun_out/${tag}_bench_default_wall.log
timeout 900 python3 bench.py --path euka --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_euka.json
timeout 900 python3 bench.py --path soibean --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_soibean.json
timeout 900 python3 bench.py --path soibean --reads 2000000 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/${tag}_bench_soibean2m.json
# the inflate kernels alone on a 1 M-read file: trace, then SQ counters
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_inflate -- python3 $R/tools/dev/inflate_time.py 1000000 > $R/gpurun_out/${tag}_inflate.log 2>&1)
bash tools/dev/pmc_cmd.sh ${tag}_inflate $R/tools/dev/inflate_time.py 1000000 > gpurun_out/${tag}_inflate_pmc.log 2>&1
bash tools/dev/frontend_kernels.sh 500000 > gpurun_out/${tag}_frontend_kernels.log 2>&1
bash tools/gpu_profile_gamdev.sh ${tag}_gamdev 10000000 > gpurun_out/${tag}_gamdev_run.log 2>&1
python3 tools/e2e_device_gam.py 10000000 2>&1 | grep -v "gampipe piece\|hc consume\|hc_devflat" | cut -c1-1500 > gpurun_out/${tag}_e2e_haplocart.log
python3 tools/e2e_device_euka.py 5000000 2>&1 | cut -c1-900 > gpurun_out/${tag}_e2e_euka.log
python3 tools/e2e_device_soibean.py 2000000 2>&1 | cut -c1-900 > gpurun_out/${tag}_e2e_soibean.log
cat gpurun_out/${tag}_pytest_gpu.log gpurun_out/${tag}_smoke.log gpurun_out/${tag}_len_sweep.jsonl gpurun_out/${tag}_bench_default_wall.log
for f in default euka soibean soibean2m; do head -c 600 gpurun_out/${tag}_bench_$f.json; echo; done
tail -5 gpurun_out/${tag}_e2e_haplocart.log gpurun_out/${tag}_e2e_euka.log gpurun_out/${tag}_e2e_soibean.log
