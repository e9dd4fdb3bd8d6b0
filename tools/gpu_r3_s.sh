# round 3: fused batch-norm backward statistics with prefetched operands -- parity + A/B bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
mkdir -p gpurun_out
SECONDS=0
timeout 1200 python -m pytest tests/test_blocks_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 | cut -c1-260
echo "blocks elapsed $SECONDS s"
for f in 0 1 0 1; do
  SE3DS_FUSED_BN_BWD=$f python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fused=$f', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['conv_ms_per_step'],2), {k:round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})
"
done
