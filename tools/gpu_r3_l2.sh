# round 3: thin head kernels with the chunk geometry hoisted out of the tile loop -- parity + A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 900 python -m pytest tests/test_prod_shapes_gpu.py tests/test_blocks_gpu.py -m gpu -x -q -k "head or 128" 2>&1 | tail -3
cat > /tmp/thin_bench.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from se3ds_amd.hipops import nn
DEV='cuda:0'
for cout in (3, 1):
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', 128, cout, 3, 1, 'VALID', True, 'spectral')
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  sg = nn.SpectralGroup([layer], torch.device(DEV))
  ctx = nn.Ctx(DEV, torch.bfloat16, training=True, record=True)
  x = nn.Var(torch.randn((8, 512, 1024, 128), device=DEV).to(torch.bfloat16))
  for it in range(3):
    ctx.tape = []
    sg.power_iteration(True)
    prof = nn.ConvProfiler(); nn.set_conv_profiler(prof if it == 2 else None)
    y = nn.conv2d(ctx, x, layer, pad=1)
    y.grad = torch.randn(y.data.shape, device=DEV).to(torch.bfloat16)
    x.grad = None
    ctx.backward(); torch.cuda.synchronize()
    nn.set_conv_profiler(None)
  print('128->%d @512x1024 n8: ' % cout + '  '.join('%s %.3f ms' % (k, v['ms']) for k, v in prof.summary()['by_kind'].items()))
PY
cp se3ds_amd/csrc/libse3ds_hip.so /tmp/new.so
for v in base new base new; do
  if [ $v = base ]; then cp se3ds_amd/csrc/_exp/lib_base.so se3ds_amd/csrc/libse3ds_hip.so; else cp /tmp/new.so se3ds_amd/csrc/libse3ds_hip.so; fi
  echo "== $v"; python /tmp/thin_bench.py 2>&1 | grep "@512"
done
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['value'],3), round(d['ms_per_step'],2), round(d['roofline']['frac'],4))
"; }
for rep in 1 2; do
  cp se3ds_amd/csrc/_exp/lib_base.so se3ds_amd/csrc/libse3ds_hip.so
  python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "old"
  cp /tmp/new.so se3ds_amd/csrc/libse3ds_hip.so
  python bench.py --no-cpu-baseline --no-batch-max 2>/dev/null | line "new"
done
