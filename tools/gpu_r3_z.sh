# round 3: conv epilogue without the activation select when the conv has none -- A/B + parity subset
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['conv_ms_per_step'],2), {k:round(v['tflops']) for k,v in d['roofline']['by_kind'].items()})
"; }
echo "== conv_bench new"; N=8 python tools/conv_bench.py 2>&1 | grep "128->128 @\|1024->1024"
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/new.so
for rep in 1 2; do
  cp se3ds_amd/csrc/_exp/lib_base.so se3ds_amd/csrc/libse3ds_hip.so
  python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "old"
  cp /tmp/new.so se3ds_amd/csrc/libse3ds_hip.so
  python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "new"
done
SECONDS=0
timeout 1500 python -m pytest tests/test_prod_shapes_gpu.py tests/test_blocks_gpu.py tests/test_golden_kernels.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-250
echo "subset elapsed $SECONDS s"
