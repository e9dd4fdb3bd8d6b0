import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets_torch as O
from se3ds_amd.models import image_models
from se3ds_amd.hipops import nn
from tests.test_nets_gpu import rel_err
DEV='cuda:0'
D = image_models.SNMultiScaleDiscriminator(n_dis=2, dis_dims=4, n_layers=3, device=DEV, seed=5)
x = torch.rand((4, 64, 128, 4), generator=torch.Generator().manual_seed(1))
p = {k: v.detach().cpu().clone().requires_grad_(not k.endswith("/u")) for k, v in D.store.views.items()}
res_o, _ = O.discriminator_forward(p, x, True, n_dis=2, n_layers=3)
for sub in res_o:
  for t in sub: t.retain_grad()
gen = torch.Generator().manual_seed(2)
seeds = [torch.randn(sub[-1].shape, generator=gen) for sub in res_o]
loss = sum((s * sub[-1]).sum() for s, sub in zip(seeds, res_o))
loss.backward()
ctx = D.make_ctx(True, record=True)
xv = nn.to_var(ctx, x.to(DEV))
res = D.forward(ctx, xv)
grads = {}
# wrap: capture grads of every feature map right before its producer's bwd consumes them
tape = ctx.tape; ctx.tape = None
for sub, s in zip(res, seeds):
  sub[-1].grad = s.to(DEV)
for i, fn in enumerate(reversed(tape)):
  for si, sub in enumerate(res):
    for li, v in enumerate(sub):
      if v.grad is not None and (si, li) not in grads:
        grads[(si, li)] = v.grad.float().cpu().numpy().copy()
  fn()
for (si, li), g in sorted(grads.items()):
  go = res_o[si][li].grad.numpy()
  print('scale', si, 'map', li, g.shape, 'err %.3e' % rel_err(g, go))
print('--- param grads (random seeds)')
D.spectral.backward_fixup()
for k in D.store.trainable_names:
  print('%-24s err %.3e scale %.2e' % (k, rel_err(D.store.grad_views[k].cpu().numpy(), p[k].grad.numpy()), p[k].grad.abs().max()))
# constant (hinge-like) seeds, fp32 oracle vs fp64 oracle vs hip
for k in p: p[k].grad = None
def run_oracle(dt):
  torch.set_default_dtype(dt)
  pp = {k: v.detach().to(dt).requires_grad_(not k.endswith('/u')) for k, v in D0.items()}
  r, _ = O.discriminator_forward(pp, x.to(dt), True, n_dis=2, n_layers=3)
  loss = sum(torch.cat([torch.relu(1 + s[-1][:2]), torch.relu(1 - s[-1][2:])]).mean() for s in r)
  loss.backward()
  torch.set_default_dtype(torch.float32)
  return {k: v.grad.double().numpy() for k, v in pp.items() if v.grad is not None}
D2 = image_models.SNMultiScaleDiscriminator(n_dis=2, dis_dims=4, n_layers=3, device=DEV, seed=5)
D0 = {k: v.detach().cpu().clone() for k, v in D2.store.views.items()}
g32, g64 = run_oracle(torch.float32), run_oracle(torch.float64)
ctx = D2.make_ctx(True, record=True)
res = D2.forward(ctx, nn.to_var(ctx, x.to(DEV)))
import se3ds_amd._lib as _lib
for sub in res:
  last = sub[-1].data; half = last.numel() // 2
  s = torch.empty(2, device=DEV); gd = torch.empty_like(last)
  _lib.check(_lib.lib().se3ds_hinge(last.data_ptr(), ctx.code, half, 0.5 / half, 0.0, s.data_ptr(), gd.data_ptr(), None, _lib.stream()), 'h')
  sub[-1].grad = gd
ctx.backward(); D2.spectral.backward_fixup()
print('--- param grads (hinge seeds): hip-vs-f64, f32oracle-vs-f64')
for k in D2.store.trainable_names:
  print('%-24s %.3e %.3e scale %.2e' % (k, rel_err(D2.store.grad_views[k].cpu().numpy(), g64[k]), rel_err(g32[k], g64[k]), np.abs(g64[k]).max()))
