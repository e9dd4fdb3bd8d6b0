"""The 1x1 family of the step through the C entry points (no Python op overhead), forward / data
gradient / data gradient into an existing gradient, under the tile-selection switches:
  python tools/conv1x1_bench.py            # default cost model, SE3DS_BIG_TILE=0, =1 side by side
Times in us and TFLOP/s; HBM floor = (input + output [+ addend]) bytes at 5 TB/s."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
from se3ds_amd import _lib
DEV = 'cuda:0'
N = int(os.environ.get('N', '8'))
shapes = [  # cin, cout, h, w, launches per step (fwd)
    (512, 2048, 32, 64, 23), (2048, 512, 32, 64, 24), (256, 1024, 64, 128, 4), (1024, 256, 64, 128, 5),
    (128, 512, 128, 256, 4), (512, 128, 128, 256, 4), (128, 128, 256, 512, 2), (1024, 4096, 16, 32, 3),
    (4096, 1024, 16, 32, 2), (512, 256, 128, 256, 1), (1024, 512, 64, 128, 1), (2048, 1024, 32, 64, 1),
]
dtype = torch.bfloat16
L = _lib.lib()


def timed(fn, reps=30):
  for _ in range(3):
    fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


variants = [('default', None), ('BIG_TILE=0', '0'), ('BIG_TILE=1', '1')]
tot = {v[0]: 0.0 for v in variants}
for cin, cout, h, w, cnt in shapes:
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, 1, 1, 'VALID', False, 'plain')
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  x = torch.randn((N, h, w, cin), device=DEV).to(dtype)
  wt, wn = layer.operands(ctx)
  y = torch.empty((N, h, w, cout), device=DEV, dtype=dtype)
  dx = torch.empty_like(x)
  dy = torch.randn((N, h, w, cout), device=DEV).to(dtype)
  rows = int(L.se3ds_conv2d_fwd_stats_rows(3, N, cin, h, w, cout, 1, 1, 1, 0, 1))
  stats = torch.empty((max(rows, 1), 2, cout), device=DEV)
  s_ = _lib.stream()
  def fwd():
    if rows > 0:
      L.se3ds_conv2d_fwd_stats(x.data_ptr(), wt.data_ptr(), y.data_ptr(), 3, N, h, w, cin, h, w, cout, 1, 1, 1,
                               0, 0, 0, None, 1, None, None, None, None, 0, 0.0, stats.data_ptr(), s_)
    else:
      L.se3ds_conv2d_fwd(x.data_ptr(), wt.data_ptr(), y.data_ptr(), 3, N, h, w, cin, h, w, cout, 1, 1, 1, 0, 0,
                         0, None, 1, None, None, None, None, 0, 0.0, s_)
  def dgrad():
    L.se3ds_conv2d_dgrad(dy.data_ptr(), wn.data_ptr(), dx.data_ptr(), 3, N, h, w, cin, h, w, cout, 1, 1, 1, 0,
                         0, 0, None, None, None, None, 0, 0.0, s_)
  def dgrad_acc():
    L.se3ds_conv2d_dgrad_acc(dy.data_ptr(), wn.data_ptr(), dx.data_ptr(), 3, N, h, w, cin, h, w, cout, 1, 1,
                             1, 0, 0, 0, None, None, None, None, 0, 0.0, dx.data_ptr(), s_)
  flops = 2.0 * N * h * w * cin * cout
  px = N * h * w
  floor = {'fwd': (px * (cin + cout) * 2) / 5e6, 'dgrad': (px * (cin + cout) * 2) / 5e6,
           'dgrad_acc': (px * (2 * cin + cout) * 2) / 5e6}
  line = '1x1 %4d->%4d @%dx%d n%d (x%d): ' % (cin, cout, h, w, N, cnt)
  for what, fn in (('fwd', fwd), ('dgrad', dgrad), ('dgrad_acc', dgrad_acc)):
    ts = []
    for name, v in variants:
      if v is None:
        os.environ.pop('SE3DS_BIG_TILE', None)
      else:
        os.environ['SE3DS_BIG_TILE'] = v
      t = timed(fn)
      ts.append(t)
      if what != 'dgrad':
        tot[name] += t * cnt
    os.environ.pop('SE3DS_BIG_TILE', None)
    line += '%s %s us (floor %.0f, %4.0f TF/s) | ' % (what, ' / '.join('%5.1f' % t for t in ts), floor[what],
                                                      flops / ts[0] / 1e6)
  print(line)
print('per step (fwd + dgrad_acc, launches weighted):', {k: '%.2f ms' % (v / 1e3) for k, v in tot.items()})
