for L in "$@"; do
echo "== $L"
VGAN_LIB=vgan_amd/lib/libvgan_gpu$L.so timeout 300 python bench.py --no-pmc --no-frontend --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('step ms', r['ms_per_step'], 'kernel ms', r['roofline']['avg_launch_ms'])"
done
timeout 500 python -m pytest tests/test_hc_gpu.py -q -x 2>&1 | tail -2
