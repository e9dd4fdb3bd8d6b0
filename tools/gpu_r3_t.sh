# round 3: fused bias gradient (runtime switch) and addend prefetch (library variant) -- parity + A/B
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
  SE3DS_FUSED_BIAS_GRAD=0 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "bias=0"
  SE3DS_FUSED_BIAS_GRAD=1 python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "bias=1"
done
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/base.so
cp se3ds_amd/csrc/libse3ds_hip_pf.so se3ds_amd/csrc/libse3ds_hip.so
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "bias=1 addend-prefetch"
done
timeout 600 python -m pytest tests/test_prod_shapes_gpu.py -m gpu -x -q -k "accumulates" 2>&1 | tail -3
cp /tmp/base.so se3ds_amd/csrc/libse3ds_hip.so
SECONDS=0
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_prod_shapes_gpu.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-200
echo "nets+prod elapsed $SECONDS s"
