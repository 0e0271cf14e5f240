#!/bin/bash
# SQ / TCC counters of the bench command in separate rocprofv3 --pmc passes (8 SQ slots per pass), summarised per kernel.
# usage: tools/gpu_pmc.sh <tag> <bench args...>     (run on the GPU box; writes gpurun_out/pmc_<tag>/)
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $R/bench.py "$@" --cpu-seconds 0 --no-extra --no-pmc --no-frontend --steps 3 --warmup 1 > $out/p$i.log 2>&1
done
cd $R
python3 tools/summarize_pmc.py $out
