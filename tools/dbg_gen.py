import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets_torch as O
from se3ds_amd.models import image_models
from tests.test_nets_gpu import synth_batch, rel_err
DEV='cuda:0'
training=True
G = image_models.ResNetGenerator(image_size=64, gen_dims=4, z_dim=4, resnet_version='50', device=DEV, seed=3)
batch = synth_batch(2, 64)
p = {k: v.detach().cpu().clone() for k, v in G.store.views.items()}
to, to64 = {}, {}
outs_o, upd = O.generator_forward(p, batch, training, gen_dims=4, resnet_version='50', z_dim=4, taps=to)
torch.set_default_dtype(torch.float64)
outs_64, _ = O.generator_forward({k: v.double() for k, v in p.items()}, {k: v.double() for k, v in batch.items()}, training, gen_dims=4, resnet_version='50', z_dim=4, taps=to64)
torch.set_default_dtype(torch.float32)
ctx = G.make_ctx(training)
ctx.taps = {}
outs, _ = G.forward(ctx, {k: v.to(DEV) for k, v in batch.items()})
for k in ('b1','s1','s2','s3','enc','ctx','dec','ddec'):
  a = ctx.taps[k].data.float().cpu().numpy(); b = to[k].detach().numpy(); c = to64[k].detach().numpy()
  print(k, 'hip-vs-f32oracle %.2e  hip-vs-f64 %.2e  f32oracle-vs-f64 %.2e' % (rel_err(a,b), rel_err(a,c), rel_err(b,c)))
for i in (3,6):
  a=outs[i].cpu().numpy(); b=outs_o[i].detach().numpy(); c=outs_64[i].detach().numpy()
  print('out', i, 'hip-vs-f32oracle %.2e  hip-vs-f64 %.2e  f32oracle-vs-f64 %.2e' % (rel_err(a,b), rel_err(a,c), rel_err(b,c)))
