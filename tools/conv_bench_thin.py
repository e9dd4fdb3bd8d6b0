import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
DEV = 'cuda:0'
N = int(os.environ.get('N', '8'))
shapes = [  # name, cin, cout, k, stride, h, w, pad, xgrad
    ('enc conv1 7x7s2 5->128 @512x1024', 5, 128, 7, 2, 512, 1024, 3, False),
    ('D g0 4x4s2 4->128 @512x1024 (2N)', 4, 128, 4, 2, 512, 1024, 2, True),
    ('head last 3x3 128->3 @512x1024', 128, 3, 3, 1, 512, 1024, 1, True),
    ('head last 3x3 128->1 @512x1024', 128, 1, 3, 1, 512, 1024, 1, True),
    ('D final 4x4 512->1 @18x34 (2N)', 512, 1, 4, 1, 18, 34, 0, True),
]
dtype = torch.bfloat16
for name, cin, cout, k, s, h, w, pad, xg in shapes:
  n = 2 * N if '(2N)' in name else N
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, k, s, 'VALID' if pad else 'SAME', True, 'plain')
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  x = nn.Var(torch.randn((n, h, w, cin), device=DEV).to(dtype), requires_grad=xg)
  res = {}
  for it in range(3):
    prof = nn.ConvProfiler(); nn.set_conv_profiler(prof if it == 2 else None)
    ctx.tape = []
    y = nn.conv2d(ctx, x, layer, pad=pad)
    y.grad = torch.randn(y.data.shape, device=DEV).to(dtype)
    x.grad = None
    ctx.backward()
    torch.cuda.synchronize()
    nn.set_conv_profiler(None)
    if it == 2:
      res = prof.summary()['by_kind']
  print('%-36s n=%d  ' % (name, n) + '  '.join('%s %.3f ms' % (kk, v['ms']) for kk, v in res.items()))
