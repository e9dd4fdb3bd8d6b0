# round 3: pre-scaled dx + bias gradient from the norm backward apply (partial convs) -- parity + A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1200 python -m pytest tests/test_blocks_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 | cut -c1-260
echo "blocks elapsed $SECONDS s"
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['conv_ms_per_step'],2), {k:round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})
"; }
for rep in 1 2; do
  SE3DS_FUSED_ROW_SCALE=0 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "rows=0"
  SE3DS_FUSED_ROW_SCALE=1 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "rows=1"
done
SE3DS_NORM_DEBUG=1 python tools/step_times.py 2>&1 | grep -i "fused-rows\|wall\|host" | head
SECONDS=0
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_prod_shapes_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-200
echo "nets+prod+dist elapsed $SECONDS s"
