"""One conv shape, fwd + bwd a few times (for PMC runs): python tools/one_conv.py cin cout k s h w pad n"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
cin, cout, k, s, h, w, pad, n = map(int, sys.argv[1:9])
DEV = 'cuda:0'
dtype = torch.bfloat16
store = nn.ParamStore()
layer = nn.ConvLayer(store, 'c', cin, cout, k, s, 'VALID', False, 'plain')
store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
ctx = nn.Ctx(DEV, dtype, training=True, record=True)
x = nn.Var(torch.randn((n, h, w, cin), device=DEV).to(dtype))
for it in range(4):
  ctx.tape = []
  y = nn.conv2d(ctx, x, layer, pad=pad)
  y.grad = torch.randn(y.data.shape, device=DEV).to(dtype)
  x.grad = None
  ctx.backward()
torch.cuda.synchronize()
