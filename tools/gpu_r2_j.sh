# round-2 tenth pass: norm kernels (two-row), tiny-map production-width convs, accumulate dgrad, bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
timeout 300 python tools/norm_kernels_bench.py 2>/dev/null
SECONDS=0
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_prod_shapes_gpu.py -k "cfg1 or accumulates" -m gpu -q -s -p no:cacheprovider > gpurun_out/r2_j.log 2>&1
echo "pytest rc=$? elapsed $SECONDS s"
grep -E "passed|failed|error" gpurun_out/r2_j.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2_j.log | head -40
grep -aoE "[A-Za-z0-9 >@_-]+ n[0-9]+ (float32|bfloat16): [a-z0-9=(). e+-]+" gpurun_out/r2_j.log | head -60
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value',d['value'],'ms',d['ms_per_step'],'conv ms',r['conv_ms_per_step'],'frac',r['frac'], {k:round(v['tflops']) for k,v in r['by_kind'].items()})
print(d.get('warp',{}).get('roofline'))"
