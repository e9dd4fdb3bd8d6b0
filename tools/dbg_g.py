import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets_torch as O
from se3ds_amd.models import image_models
from tests.test_nets_gpu import synth_batch, rel_err
DEV='cuda:0'
G = image_models.ResNetGenerator(image_size=64, gen_dims=4, z_dim=4, resnet_version='50', device=DEV, seed=3)
gen = torch.Generator().manual_seed(4)
# non-trivial BN state so that moving-stat mode is not the identity
upd = {}
for n in G.store.state_names:
  if n.endswith('moving_mean'): upd[n] = (torch.randn(G.store[n].shape, generator=gen) * 0.1).numpy()
  if n.endswith('moving_variance'): upd[n] = (torch.rand(G.store[n].shape, generator=gen) + 0.5).numpy()
for n in G.store.trainable_names:
  if n.endswith('gamma'): upd[n] = (torch.rand(G.store[n].shape, generator=gen) + 0.5).numpy()
  if n.endswith('beta') or n.endswith('bias'): upd[n] = (torch.randn(G.store[n].shape, generator=gen) * 0.1).numpy()
G.store.load_dict(upd)
batch = synth_batch(2, 64)
p = {k: v.detach().cpu().clone().requires_grad_(k in G.store.trainable_names) for k, v in G.store.views.items()}
outs_o, _ = O.generator_forward(p, batch, True, gen_dims=4, resnet_version='50', z_dim=4, bn_training=False)
w_rgb = torch.randn(outs_o[6].shape, generator=gen); w_d = torch.randn(outs_o[3].shape, generator=gen)
((outs_o[6] * w_rgb).sum() + (outs_o[3] * w_d).sum()).backward()
ctx = G.make_ctx(True, record=True); ctx.bn_use_moving = True
outs, (push_rgb, push_depth) = G.forward(ctx, {k: v.to(DEV) for k, v in batch.items()})
print('fwd rgb %.2e depth %.2e' % (rel_err(outs[6].cpu().numpy(), outs_o[6].detach().numpy()), rel_err(outs[3].cpu().numpy(), outs_o[3].detach().numpy())))
push_rgb(w_rgb.to(DEV)); push_depth(w_d.to(DEV)); ctx.backward(); G.spectral.backward_fixup()
errs = []
for k in G.store.trainable_names:
  go = p[k].grad
  if go is None: print('oracle grad None', k); continue
  sc = float(go.abs().max())
  e = rel_err(G.store.grad_views[k].cpu().numpy(), go.numpy()) if sc > 1e-9 else float(G.store.grad_views[k].abs().max())
  errs.append((e, sc, k))
errs.sort(reverse=True)
for e in errs[:12]: print('%.3e scale %.2e %s' % e)
print('n tensors', len(errs), 'median err %.2e' % np.median([e[0] for e in errs]))
