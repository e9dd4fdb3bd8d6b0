"""The thin layers through the Python ops, HIP events per kernel class (nn.ConvProfiler): heads 3x3
128 -> 3 / 1 @512x1024, stems 7x7 s2 5 -> 128 and 4x4 s2 4 -> 128.  ms per call, batch N (default 8)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
DEV = 'cuda:0'
dtype = torch.bfloat16
N = int(os.environ.get('N', '8'))
cases = [  # kind, cin, cout, k, stride, pad, bias, mask, n, h, w
    ('spectral', 128, 3, 3, 1, 1, True, False, N, 512, 1024),
    ('spectral', 128, 1, 3, 1, 1, True, False, N, 512, 1024),
    ('partial', 5, 128, 7, 2, 3, True, True, N, 512, 1024),
    ('plain', 4, 128, 4, 2, 2, True, False, 2 * N, 512, 1024),
]
for kind, cin, cout, k, s, pad, bias, masked, n, h, w in cases:
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, k, s, 'VALID', bias, kind)
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  sg = nn.SpectralGroup([layer], torch.device(DEV))
  sg.power_iteration(True)
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  x = nn.Var(torch.randn((n, h, w, cin), device=DEV).to(dtype))
  mask = (torch.rand((n, h, w), device=DEV) < 0.9).float() if masked else None
  acc = {}
  for it in range(8):
    prof = nn.ConvProfiler()
    nn.set_conv_profiler(prof if it >= 3 else None)
    ctx.tape = []
    y = nn.conv2d(ctx, x, layer, pad=pad, mask=mask) if masked else nn.conv2d(ctx, x, layer, pad=pad)
    y = y[0] if isinstance(y, tuple) else y
    y.grad = torch.randn(y.data.shape, device=DEV).to(dtype)
    x.grad = None
    ctx.backward()
    torch.cuda.synchronize()
    nn.set_conv_profiler(None)
    if it >= 3:
      for kk, v in prof.summary()['by_kind'].items():
        acc.setdefault(kk, []).append(v['ms'])
  print('THIN %dx%ds%d %d->%d @%dx%d n%d: ' % (k, k, s, cin, cout, h, w, n) +
        '  '.join('%s %.3f ms' % (kk, min(v)) for kk, v in acc.items()))
  del x, y, ctx, store
  torch.cuda.empty_cache()
