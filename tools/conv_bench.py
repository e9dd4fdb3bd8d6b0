"""Per-shape timing of the conv kernels (fwd / dgrad / wgrad), bf16, HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
from se3ds_amd import _lib
DEV = 'cuda:0'
N = int(os.environ.get('N', '4'))
shapes = [  # name, cin, cout, k, stride, h, w, pad
    ('dec1 3x3 1024->1024 @32x64', 1024, 1024, 3, 1, 32, 64, 1),
    ('head 3x3 128->128 @512x1024', 128, 128, 3, 1, 512, 1024, 1),
    ('dec4 3x3 128->128 @256x512', 128, 128, 3, 1, 256, 512, 1),
    ('dec2 3x3 512->512 @32x64', 512, 512, 3, 1, 32, 64, 1),
    ('dec3 3x3 256->256 @64x128', 256, 256, 3, 1, 64, 128, 1),
    ('enc 1x1 2048->512 @32x64', 2048, 512, 1, 1, 32, 64, 0),
    ('enc 1x1 512->2048 @32x64', 512, 2048, 1, 1, 32, 64, 0),
    ('enc 3x3 512->512 @32x64', 512, 512, 3, 1, 32, 64, 1),
    ('D 4x4s2 128->256 @257x513', 128, 256, 4, 2, 257, 513, 2),
]
dtype = torch.bfloat16
for name, cin, cout, k, s, h, w, pad in shapes:
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, k, s, 'VALID', False, 'plain')
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  ctx = nn.Ctx(DEV, dtype, training=True, record=True)
  x = nn.Var(torch.randn((N, h, w, cin), device=DEV).to(dtype))
  if os.environ.get('ZERO'):
    x.data.zero_(); store.theta.zero_()
  flops = None
  res = {}
  for it in range(3):
    ctx.tape = []
    y = nn.conv2d(ctx, x, layer, pad=pad)
    y.grad = torch.randn(y.data.shape, device=DEV).to(dtype)
    prof = nn.ConvProfiler(); nn.set_conv_profiler(prof if it == 2 else None)
    if it == 2:
      ctx.tape = []
      y = nn.conv2d(ctx, x, layer, pad=pad)
      y.grad = torch.randn(y.data.shape, device=DEV).to(dtype)
    x.grad = None
    ctx.backward()
    torch.cuda.synchronize()
    nn.set_conv_profiler(None)
    if it == 2:
      res = prof.summary()['by_kind']
  print('%-30s N=%d  ' % (name, N) + '  '.join('%s %.3f ms %.0f TF/s' % (kk, v['ms'], v['tflops']) for kk, v in res.items()))
