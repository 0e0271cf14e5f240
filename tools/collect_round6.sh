#!/bin/bash
# after tools/gpu_round6.sh <tag> on the GPU box: the summaries the judge reads, from gpurun_out/ into profiles/
tag=${1:-round6_v1}
for k in node soibean euka; do python3 tools/summarize_profile.py ${tag}_$k ${tag}_$k; done
cp gpurun_out/pmc_${tag}_node_sq/summary.json profiles/${tag}_node_sq.json
cp gpurun_out/pmc_${tag}_node_sq/kernel_stats.csv profiles/${tag}_node_sq_kernel_stats.csv
cp gpurun_out/pmc_${tag}_inflate/summary.json profiles/${tag}_inflate_sq.json 2>/dev/null
f=$(ls gpurun_out/prof_${tag}_inflate/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f profiles/${tag}_inflate_kernel_stats.csv
f=$(ls gpurun_out/prof_${tag}_gamdev/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f profiles/${tag}_gamdev_kernel_stats.csv
for f in default euka soibean soibean2m; do cp gpurun_out/${tag}_bench_$f.json profiles/; done
cp gpurun_out/${tag}_len_sweep.jsonl gpurun_out/${tag}_class_sweep.jsonl profiles/
for f in inflate gamdev e2e_haplocart e2e_euka e2e_soibean pytest_gpu smoke frontend_kernels bench_default_wall; do [ -f gpurun_out/${tag}_$f.log ] && cp gpurun_out/${tag}_$f.log profiles/${tag}_$f.log; done
ls -la profiles | grep ${tag}
