"""Bandwidth of the norm kernels (bf16): fwd (stats + apply) and bwd (stats + apply) per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from se3ds_amd.hipops import nn
DEV = 'cuda:0'
shapes = [('batch', 8, 256, 512, 128, False), ('batch', 8, 128, 256, 256, True), ('batch', 8, 64, 128, 512, True),
          ('batch', 8, 32, 64, 1024, True), ('batch', 8, 32, 64, 2048, True), ('batch', 8, 512, 1024, 128, False),
          ('instance', 16, 257, 513, 128, False), ('instance', 16, 129, 257, 256, False)]
for kind, n, h, w, c, with_res in shapes:
  store = nn.ParamStore()
  layer = nn.NormLayer(store, 'n', c, kind)
  store.finalize(DEV, torch.Generator(device=DEV).manual_seed(1))
  ctx = nn.Ctx(DEV, torch.bfloat16, training=True, record=True)
  ctx.param_grads = True
  x = nn.Var(torch.randn((n, h, w, c), device=DEV).bfloat16(), requires_grad=True)
  res = nn.Var(torch.randn((n, h, w, c), device=DEV).bfloat16(), requires_grad=True) if with_res else None
  nb = x.data.numel() * 2 / 1e9
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
  for it in range(3):
    ctx.tape = []
    x.grad = None
    if res is not None: res.grad = None
    ev[0].record()
    y = nn.norm_act(ctx, x, layer, act=nn.ACT_LEAKY if hasattr(nn, 'ACT_LEAKY') else 2, alpha=0.2, res=res)
    ev[1].record()
    y.grad = torch.randn_like(y.data)
    ev[2].record() if False else None
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    s0.record()
    ctx.backward()
    s1.record()
    torch.cuda.synchronize()
  tf, tb = ev[0].elapsed_time(ev[1]), s0.elapsed_time(s1)
  pf = 3 + (1 if with_res else 0)            # fwd passes: x, x, y (+res)
  pb = 7 + (1 if with_res else 0)            # bwd passes: dy,y,x, dy,y,x, dx (+dres)
  print('%-8s %s  tensor %.3f GB  fwd %.3f ms (%.2f TB/s for %d passes)  bwd %.3f ms (%.2f TB/s for %d passes)' %
        (kind, (n, h, w, c), nb, tf, pf * nb / tf, pf, tb, pb * nb / tb, pb))
