cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
VGAN_LIB=$R/vgan_amd/lib/libvgan_gpu_st.so python3 tools/col8_stats.py 1000000 150 > gpurun_out/r5_stats.log 2>&1
python3 tools/wave_time.py 1000000 150 20 >> gpurun_out/r5_stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_a -- python3 tools/wave_time.py 1000000 150 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_b -- python3 tools/wave_time.py 1000000 150 3 > /dev/null 2>&1
python3 - <<'P' >> gpurun_out/r5_stats.log
import csv,glob,collections
for d in ('pmc_a','pmc_b'):
    v=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv'%d, recursive=True):
        for row in csv.DictReader(open(f)):
            if 'hc_segment' in row['Kernel_Name']:
                v[row['Counter_Name']].append(float(row['Counter_Value']))
    for k,x in sorted(v.items()): print(d,k,sum(x)/len(x),len(x))
P
cat gpurun_out/r5_stats.log
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
