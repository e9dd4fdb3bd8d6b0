import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets_torch as O
from tests.test_nets_gpu import synth_batch, rel_err, _make_gan, _oracle_params, _cfg, _f64, fp64_oracle
DEV='cuda:0'
gan = _make_gan(64, 4, '50', 3)
batch = synth_batch(2, 64, seed=78)
gp, dp = _oracle_params(gan.generator), _oracle_params(gan.discriminator)
cfg = _cfg(4, '50', 3)
G = gan.generator
ctx = G.make_ctx(True)
outs, _ = G.forward(ctx, {k: v.to(DEV) for k, v in batch.items()})
gen_h, dep_h = outs[6].cpu(), outs[3].cpu()
with torch.no_grad():
  o32, _ = O.generator_forward(gp, batch, True, **cfg['gen'])
with fp64_oracle():
  with torch.no_grad():
    o64, _ = O.generator_forward(_f64(gp), _f64(batch), True, **cfg['gen'])
print('gen   hip-vs-64 %.2e  o32-vs-64 %.2e' % (rel_err(gen_h.numpy(), o64[6].numpy()), rel_err(o32[6].numpy(), o64[6].numpy())))
print('depth hip-vs-64 %.2e  o32-vs-64 %.2e' % (rel_err(dep_h.numpy(), o64[3].numpy()), rel_err(o32[3].numpy(), o64[3].numpy())))
d = (dep_h.double() - o64[3]).abs(); print('depth abs err max', float(d.max()), 'mean', float(d.mean()), 'num>1e-3', int((d>1e-3).sum()), 'of', d.numel())
d2 = (o32[3].double() - o64[3]).abs(); print('o32 depth abs err max', float(d2.max()), 'mean', float(d2.mean()), 'num>1e-3', int((d2>1e-3).sum()))
def dgrads(gen, dep):
  with fp64_oracle():
    pp = {k: v.double().requires_grad_(not k.endswith('/u')) for k, v in dp.items()}
    fake = torch.cat([gen.double(), dep.double()], -1); real = torch.cat([batch['image'].double(), batch['depth'].double()], -1)
    logits, _ = O.discriminator_forward(pp, torch.cat([fake, real], 0), True, **cfg['dis'])
    _, dl = O.d_losses(logits)
    dl.backward()
    return {k: v.grad.numpy() for k, v in pp.items() if v.grad is not None}
g_ref = dgrads(o64[6], o64[3]); g_h = dgrads(gen_h, dep_h); g_o = dgrads(o32[6], o32[3])
for k in list(g_ref)[:3] + ['dis1/g0/conv/kernel']:
  print(k, 'D-grad sensitivity: with hip G out %.2e, with o32 G out %.2e' % (rel_err(g_h[k], g_ref[k]), rel_err(g_o[k], g_ref[k])))
