#!/bin/bash
# SQ / TCC counters of tools/wave_prof.py in separate rocprofv3 --pmc passes (8 SQ slots per pass) + a kernel trace,
# summarised per kernel.   usage: tools/gpu_pmc_wave.sh <tag> [wave_prof args]   (GPU box; writes gpurun_out/pmc_<tag>/)
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/wave_prof.py "$@" > $out/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $R/tools/wave_prof.py "$@" > $out/p$i.log 2>&1
done
cd $R
python3 tools/summarize_pmc.py $out > $out/summary.txt
for f in $(find $out/trace -name "*kernel_stats.csv"); do cp $f $out/kernel_stats.csv; done
rm -rf $out/p? $out/trace
cat $out/summary.txt | grep -A40 "segment"
