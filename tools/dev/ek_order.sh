#!/bin/bash
# developer A/B: euka's read kernel by the span its blocks' reads are length-ordered within (tagged builds -DEK_ORDER_SPAN=N: _oN):
# time per launch and FETCH_SIZE per launch
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
for t in "" "$@"; do
  export VGAN_LIB=$R/vgan_amd/lib/libvgan_gpu$t.so
  python3 $R/bench.py --path euka --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline', {})
print('lib%-5s value %.3e  ms_per_step %.4f  kernel %.4f ms' % ('$t', d['value'], d['ms_per_step'], r.get('avg_launch_ms', 0)))"
  rm -rf /tmp/ekpmc; (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ekpmc -- python3 $R/bench.py --path euka --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1)
  python3 - <<PY
import csv, glob
for f in glob.glob('/tmp/ekpmc/**/*counter_collection.csv', recursive=True):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'euka_read_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
    if v: print('   FETCH_SIZE per launch: %.1f MB raw (x2 = %.1f MB) over %d launches' % (sum(v) / len(v) / 1024, sum(v) / len(v) / 512, len(v)))
PY
done
