cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
run() { # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$n -- python3 tools/wave_time.py 1000000 150 3 > /dev/null 2>&1
}
run a FETCH_SIZE
run b WRITE_SIZE
run c TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run d TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run e SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run f SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
python3 - <<'P' > gpurun_out/r5_counters2.log
import csv,glob,collections
for d in 'abcdef':
    v=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv'%d, recursive=True):
        for row in csv.DictReader(open(f)):
            if 'hc_segment' in row['Kernel_Name']:
                v[row['Counter_Name']].append(float(row['Counter_Value']))
    for k,x in sorted(v.items()): print(d,k,sum(x)/len(x),len(x))
P
cat gpurun_out/r5_counters2.log
rm -rf gpurun_out/pmc_?
