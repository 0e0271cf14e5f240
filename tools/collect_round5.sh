#!/bin/bash
# after tools/gpu_round5.sh <tag> on the GPU box: the summaries the judge reads, from gpurun_out/ into profiles/
tag=${1:-round5_v2}
python3 tools/summarize_profile.py ${tag}_node ${tag}_node
python3 tools/summarize_profile.py ${tag}_soibean ${tag}_soibean
cp gpurun_out/pmc_${tag}_node_sq/summary.json profiles/${tag}_node_sq.json
cp gpurun_out/pmc_${tag}_node_sq/kernel_stats.csv profiles/${tag}_node_sq_kernel_stats.csv
cp gpurun_out/${tag}_bench_default.json gpurun_out/${tag}_bench_euka.json gpurun_out/${tag}_bench_soibean.json gpurun_out/${tag}_bench_soibean2m.json profiles/
cp gpurun_out/${tag}_len_sweep.jsonl profiles/
ls -la profiles | grep ${tag}
