# round 3: fused batch-norm backward statistics -- parity tests, A/B bench, which norms still take their own pass
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1200 python -m pytest tests/test_blocks_gpu.py tests/test_prod_shapes_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 | cut -c1-260
echo "blocks+prod elapsed $SECONDS s"
for f in 0 1 0 1; do
  SE3DS_FUSED_BN_BWD=$f python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fused=$f', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['conv_ms_per_step'],2), {k:round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})
"
done
SE3DS_NORM_DEBUG=1 python tools/step_times.py 2>&1 | tail -70 | cut -c1-200
SECONDS=0
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -5 | cut -c1-200
echo "nets elapsed $SECONDS s"
