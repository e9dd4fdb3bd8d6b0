"""fwd / dgrad / wgrad time of partial (masked) convs vs the same shapes without a mask (bf16).
  python tools/masked_conv_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
DEV = 'cuda:0'
N = int(os.environ.get('N', '8'))
shapes = [('1x1 2048->512 @32x64', 2048, 512, 1, 32, 64, 0), ('1x1 512->2048 @32x64', 512, 2048, 1, 32, 64, 0),
          ('3x3 512->512 @32x64', 512, 512, 3, 32, 64, 1), ('1x1 1024->256 @64x128', 1024, 256, 1, 64, 128, 0),
          ('3x3 256->256 @64x128', 256, 256, 3, 64, 128, 1)]
for name, cin, cout, k, h, w, pad in shapes:
  for masked in (False, True):
    store = nn.ParamStore()
    layer = nn.ConvLayer(store, 'c', cin, cout, k, 1, 'VALID' if k == 3 else 'SAME', True, 'partial')
    store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
    ctx = nn.Ctx(DEV, torch.bfloat16, training=True, record=True)
    x = nn.Var(torch.randn((N, h, w, cin), device=DEV).bfloat16(), requires_grad=True)
    mask = (torch.rand((N, h, w), device=DEV) > 0.3).float() if masked else None
    res = {}
    for it in range(3):
      ctx.tape = []
      prof = nn.ConvProfiler(); nn.set_conv_profiler(prof if it == 2 else None)
      y, um = nn.conv2d(ctx, x, layer, pad=pad, mask=mask)
      y.grad = torch.randn(y.data.shape, device=DEV).bfloat16()
      x.grad = None
      ctx.backward()
      torch.cuda.synchronize()
      nn.set_conv_profiler(None)
      if it == 2:
        res = prof.summary()['by_kind']
    print('%-24s mask=%d  ' % (name, masked) + '  '.join('%s %.3f ms' % (kk, v['ms']) for kk, v in res.items()))
