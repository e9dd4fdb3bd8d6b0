"""Generates tests/golden/kernel_fixtures.npz: seeded inputs and the CPU oracle's outputs
(oracle/nets_torch.py, PyTorch-CPU fp32) for a few small convolution and normalisation cases.

A fixture is data: the GPU tests compare the HIP path with these arrays without running the oracle
(independent of the GPU box's CPU torch build); a CPU test re-runs the oracle on the stored inputs
and must reproduce the stored outputs.  Inputs are bf16-representable, so the same arrays serve the
fp32 and the bf16 path.

  python tests/golden/make_kernel_fixtures.py      (from the repo root, CPU only)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nets_torch as O  # noqa: E402
from tests import test_prod_shapes_gpu as P  # noqa: E402

# name, kind, cin, cout, k, stride, padding, pad, bias, mask, h, w, batches, oracle chunk
CONV_CASES = [
    ('fx spectral 3x3 64->128', 'spectral', 64, 128, 3, 1, 'VALID', 1, True, False, 8, 16, (2,), 2),
    ('fx partial_spectral 1x1 128->64', 'partial_spectral', 128, 64, 1, 1, 'SAME', 0, True, True, 9, 17, (2,), 2),
    ('fx partial 3x3 s2 64->64', 'partial', 64, 64, 3, 2, 'VALID', 1, True, True, 12, 24, (2,), 2),
]
# kind, n, h, w, c, act, with_res
NORM_CASES = [
    ('batch', 2, 6, 10, 64, 1, True),
    ('instance', 2, 7, 9, 32, 2, False),
]
ALPHA = 0.2


def norm_case(case):
  kind, n, h, w, c, act, with_res = case
  gen = torch.Generator().manual_seed(100 + c + h)
  d = dict(gamma=torch.rand(c, generator=gen) + 0.5, beta=torch.randn(c, generator=gen) * 0.2,
           x=P._bf(torch.randn((n, h, w, c), generator=gen) * 0.7 + torch.randn(c, generator=gen)),
           r=P._bf(torch.randn((n, h, w, c), generator=gen)),
           gy=P._bf(torch.randn((n, h, w, c), generator=gen)))
  d.update(norm_oracle(case, d))
  return d


def norm_oracle(case, d):
  kind, n, h, w, c, act, with_res = case
  p = {'n/gamma': d['gamma'].clone().requires_grad_(True), 'n/beta': d['beta'].clone().requires_grad_(True),
       'n/moving_mean': torch.zeros(c), 'n/moving_variance': torch.ones(c)}
  xo, ro = d['x'].clone().requires_grad_(True), d['r'].clone().requires_grad_(True)
  net = O.Net(p, training=True)
  y = net.sync_bn(xo, 'n') if kind == 'batch' else net.instance_norm(xo, 'n')
  if with_res:
    y = y + ro
  y = torch.relu(y) if act == 1 else O.leaky_relu(y, ALPHA)
  y.backward(d['gy'])
  out = dict(y=y.detach(), dx=xo.grad, dgamma=p['n/gamma'].grad, dbeta=p['n/beta'].grad)
  if with_res:
    out['dres'] = ro.grad
  if kind == 'batch':
    out['moving_mean'] = net.updates['n/moving_mean']
    out['moving_variance'] = net.updates['n/moving_variance']
  return out


def main():
  arrays = {}
  for i, case in enumerate(CONV_CASES):
    n = case[12][0]
    x, kern, b, u, mask, gy = P._inputs(case, n)
    ref = P._oracle(case, n, inputs=(x, kern, b, u, mask, gy))
    for k_, v in dict(x=x, kern=kern, b=b, u=u, mask=mask, gy=gy).items():
      if v is not None:
        arrays[f'conv{i}/in/{k_}'] = v.numpy()
    for k_, v in ref.items():
      if v is not None:
        arrays[f'conv{i}/out/{k_}'] = np.asarray(v)
  for i, case in enumerate(NORM_CASES):
    for k_, v in norm_case(case).items():
      arrays[f'norm{i}/{k_}'] = v.detach().numpy()
  path = os.path.join(ROOT, 'tests', 'golden', 'kernel_fixtures.npz')
  np.savez_compressed(path, **arrays)
  print(path, os.path.getsize(path), 'bytes,', len(arrays), 'arrays')


if __name__ == '__main__':
  main()
