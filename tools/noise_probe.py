"""fp32 rounding-noise probe: for the ops that carry the cfg1 step, how far is the HIP fp32 path from
an fp64 evaluation, next to how far the fp32 PyTorch-CPU oracle is from the same fp64 evaluation?

Deep batch-normalised nets at random initialisation amplify rounding noise by ~1e4, so
a constant factor between the two noise levels shows up in the full-step parity tests.  Inputs here
are general fp32 values (NOT bf16-representable: products round too).  Metric: relative L2 error.

  python tools/noise_probe.py            (on a GPU box)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import nets_torch as O  # noqa: E402   (a measurement tool, not the product path)
from se3ds_amd.hipops import nn  # noqa: E402

DEV = 'cuda:0'


def l2(a, b):
  a = np.asarray(a, np.float64).ravel()
  b = np.asarray(b, np.float64).ravel()
  return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def probe_conv(name, cin, cout, k, stride, pad, n, h, w, spectral=False):
  gen = torch.Generator().manual_seed(cin + cout + k + h)
  x = torch.randn((n, h, w, cin), generator=gen)
  kern = nn.glorot_uniform((k, k, cin, cout), gen)
  u = nn.truncated_normal_init((1, cout), gen)
  ho = nn.conv_out_size(h, k, stride, 'VALID', pad)[0]
  wo = nn.conv_out_size(w, k, stride, 'VALID', pad)[0]
  gy = torch.randn((n, ho, wo, cout), generator=gen)
  ref = {}
  for dt in (torch.float32, torch.float64):
    ko = kern.detach().clone().to(dt).requires_grad_(True)
    xo = x.detach().clone().to(dt).requires_grad_(True)
    net = O.Net({'c/kernel': ko, 'c/u': u.to(dt)}, training=True)
    xin = O.pad_layer(xo, pad, circular_pad=False, training=True) if pad else xo
    yo = net.spectral_conv(xin, 'c', stride, 'VALID') if spectral else net.conv2d(xin, 'c', stride, 'VALID')
    yo.backward(gy.to(dt))
    ref[dt] = dict(y=yo.detach().numpy(), dx=xo.grad.numpy(), dk=ko.grad.numpy())
  store = nn.ParamStore()
  layer = nn.ConvLayer(store, 'c', cin, cout, k, stride, 'VALID', False, 'spectral' if spectral else 'plain')
  store.finalize(DEV, None)
  d = {'c/kernel': kern.numpy()}
  if spectral:
    d['c/u'] = u.numpy()
  store.load_dict(d)
  sg = nn.SpectralGroup([layer], torch.device(DEV))
  ctx = nn.Ctx(DEV, torch.float32, training=True, record=True)
  sg.power_iteration(training=False)
  xv = nn.Var(x.to(DEV), requires_grad=True)
  yv = nn.conv2d(ctx, xv, layer, pad=pad, wrap=False)
  yv.grad = gy.to(DEV)
  ctx.backward()
  sg.backward_fixup()
  got = dict(y=yv.data.cpu().numpy(), dx=xv.grad.cpu().numpy(), dk=store.grad_views['c/kernel'].cpu().numpy())
  r64 = ref[torch.float64]
  line = f'{name:34s}'
  for key in ('y', 'dx', 'dk'):
    eh, eo = l2(got[key], r64[key]), l2(ref[torch.float32][key], r64[key])
    line += f'  {key}: hip {eh:.1e} orc {eo:.1e} ({eh / max(eo, 1e-30):4.1f}x)'
  print(line, flush=True)


def probe_norm(name, kind, n, h, w, c, mean_scale):
  gen = torch.Generator().manual_seed(c + h)
  x = torch.randn((n, h, w, c), generator=gen) * 0.7 + torch.randn(c, generator=gen) * mean_scale
  gy = torch.randn((n, h, w, c), generator=gen)
  gamma = torch.rand(c, generator=gen) + 0.5
  beta = torch.randn(c, generator=gen) * 0.2
  store = nn.ParamStore()
  layer = nn.NormLayer(store, 'n', c, kind)
  store.finalize(DEV, None)
  store.load_dict({'n/gamma': gamma.numpy(), 'n/beta': beta.numpy()})
  base = {k_: v.cpu().clone() for k_, v in store.views.items()}
  ref = {}
  for dt in (torch.float32, torch.float64):
    p = {k_: v.to(dt).clone() for k_, v in base.items()}
    p['n/gamma'].requires_grad_(True)
    p['n/beta'].requires_grad_(True)
    xo = x.detach().clone().to(dt).requires_grad_(True)
    net = O.Net(p, training=True)
    yo = net.sync_bn(xo, 'n') if kind == 'batch' else net.instance_norm(xo, 'n')
    yo.backward(gy.to(dt))
    ref[dt] = dict(y=yo.detach().numpy(), dx=xo.grad.numpy(), dg=p['n/gamma'].grad.numpy(),
                   db=p['n/beta'].grad.numpy())
  ctx = nn.Ctx(DEV, torch.float32, training=True, record=True)
  xv = nn.Var(x.to(DEV))
  yv = nn.norm_act(ctx, xv, layer, act=0)
  yv.grad = gy.to(DEV)
  ctx.backward()
  got = dict(y=yv.data.cpu().numpy(), dx=xv.grad.cpu().numpy(), dg=store.grad_views['n/gamma'].cpu().numpy(),
             db=store.grad_views['n/beta'].cpu().numpy())
  r64 = ref[torch.float64]
  line = f'{name:34s}'
  for key in ('y', 'dx', 'dg', 'db'):
    eh, eo = l2(got[key], r64[key]), l2(ref[torch.float32][key], r64[key])
    line += f'  {key}: hip {eh:.1e} orc {eo:.1e} ({eh / max(eo, 1e-30):4.1f}x)'
  print(line, flush=True)


def main():
  probe_conv('3x3 1024->1024 @8x16', 1024, 1024, 3, 1, 1, 2, 8, 16)
  probe_conv('3x3 1024->1024 @8x16 spectral', 1024, 1024, 3, 1, 1, 2, 8, 16, spectral=True)
  probe_conv('3x3 4096->512 @4x8', 4096, 512, 3, 1, 1, 2, 4, 8)
  probe_conv('1x1 2048->512 @8x16', 2048, 512, 1, 1, 0, 2, 8, 16)
  probe_conv('1x1 512->2048 @8x16 spectral', 512, 2048, 1, 1, 0, 2, 8, 16, spectral=True)
  probe_conv('3x3 s2 256 @32x64', 256, 256, 3, 2, 1, 2, 32, 64)
  probe_conv('3x3 128 @128x256', 128, 128, 3, 1, 1, 2, 128, 256)
  probe_conv('7x7 s2 5->128 @128x256', 5, 128, 7, 2, 3, 2, 128, 256)
  probe_conv('4x4 s2 128->256 @65x129', 128, 256, 4, 2, 2, 4, 65, 129)
  for ms in (0.0, 2.0):
    probe_norm(f'batch 2x4x8x4096 mean {ms}', 'batch', 2, 4, 8, 4096, ms)
    probe_norm(f'batch 2x8x16x1024 mean {ms}', 'batch', 2, 8, 16, 1024, ms)
    probe_norm(f'batch 2x32x64x256 mean {ms}', 'batch', 2, 32, 64, 256, ms)
    probe_norm(f'batch 2x128x256x128 mean {ms}', 'batch', 2, 128, 256, 128, ms)
    probe_norm(f'instance 4x33x65x256 mean {ms}', 'instance', 4, 33, 65, 256, ms)


if __name__ == '__main__':
  main()
