"""CPU tests: pin the network oracle (oracle/nets_torch.py) against the known-answer tests
the reference holds for models/layers.py and models/image_models.py."""
import os

import numpy as np
import pytest
import torch

from oracle import nets_torch as O


def test_pad_layer_golden(golden_dir):
  # models/layers_test.py:136-179 -- three full 8x8 expected outputs
  g = np.load(os.path.join(golden_dir, 'reference_literals.npz'))
  x = torch.from_numpy(g['pad_input']).reshape(1, 4, 4, 1)
  out = O.pad_layer(x, 2, circular_pad=True, training=None)
  np.testing.assert_allclose(out[0, :, :, 0].numpy(), g['pad_const_circ'])
  out = O.pad_layer(x, 2, circular_pad=False, training=None)
  np.testing.assert_allclose(out[0, :, :, 0].numpy(), g['pad_const_nocirc'])
  out = O.pad_layer(x, 2, circular_pad=True, training=None, mode='SYMMETRIC')
  np.testing.assert_allclose(out[0, :, :, 0].numpy(), g['pad_symm_circ'])
  # training=True disables the circular wrap (layers.py:68-72)
  out = O.pad_layer(x, 2, circular_pad=True, training=True)
  np.testing.assert_allclose(out[0, :, :, 0].numpy(), g['pad_const_nocirc'])


def _glorot(shape, gen):
  rf = int(np.prod(shape[:-2]))
  lim = (6.0 / ((shape[-2] + shape[-1]) * rf)) ** 0.5
  return (torch.rand(shape, generator=gen) * 2 - 1) * lim


@pytest.mark.parametrize('batch,k,stride', [(1, 3, 2), (4, 5, 1)])
def test_partial_conv_equals_conv_without_mask(batch, k, stride):
  # models/layers_test.py:106-134: PartialConv(no mask) == tf.nn.conv2d(x, kernel)
  gen = torch.Generator().manual_seed(0)
  x = torch.rand((batch, 32, 32, 32), generator=gen)
  p = {'c/kernel': _glorot((k, k, 32, 16), gen), 'c/bias': torch.zeros(16)}
  net = O.Net(p, training=None)
  out, um = net.partial_conv(x, None, 'c', stride, 'VALID')
  ref = O.tf_conv2d(x, p['c/kernel'], stride, 'VALID')
  assert out.shape == ref.shape and um.shape == ref.shape[:3] + (1,)
  np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
  # with a mask the shapes stay the same, spectral variant too
  mask = (torch.rand((batch, 32, 32, 1), generator=gen) > 0.5).float()
  p['c/u'] = torch.randn((1, 16), generator=gen) * 0.05
  o2, _ = net.partial_conv(x, mask, 'c', stride, 'VALID', spectral=True)
  assert o2.shape == ref.shape


def _res_stack_params(name, inplanes, planes, blocks, stride, gen, spectral=False, expansion=4):
  p = {}
  def conv(nm, cin, cout, k, bias=True):
    p[nm + '/kernel'] = _glorot((k, k, cin, cout), gen)
    if bias:
      p[nm + '/bias'] = torch.randn(cout, generator=gen) * 0.1
    if spectral:
      p[nm + '/u'] = torch.randn((1, cout), generator=gen) * 0.05
  def bn(nm, c):
    p[nm + '/gamma'] = torch.ones(c); p[nm + '/beta'] = torch.zeros(c)
    p[nm + '/moving_mean'] = torch.zeros(c); p[nm + '/moving_variance'] = torch.ones(c)
  cin = inplanes
  if stride != 1 or inplanes != planes * expansion:
    conv(name + '/downsample', cin, planes * expansion, 1, bias=False)
  for i in range(blocks):
    b = f'{name}/block{i}'
    conv(b + '/conv1', cin, planes, 1); bn(b + '/bn1', planes)
    conv(b + '/conv2', planes, planes, 3); bn(b + '/bn2', planes)
    conv(b + '/conv3', planes, planes * expansion, 1); bn(b + '/bn3', planes * expansion)
    if i == 0 and name + '/downsample/kernel' in p:
      bn(b + '/ds_norm', planes * expansion)
    cin = planes * expansion
  return p


@pytest.mark.parametrize('batch,size,stride', [(1, 32, 1), (2, 32, 2)])
def test_resstack_shapes(batch, size, stride):
  # models/layers_test.py:30-47
  gen = torch.Generator().manual_seed(1)
  p = _res_stack_params('s', 8, 8, 2, stride, gen)
  net = O.Net(p, training=None)
  out, mask = net.res_stack(torch.rand((batch, size, size, 8), generator=gen), None, 's', 8, 8, 2,
                            stride, False, False)
  assert out.shape == (batch, size // stride, size // stride, 32)
  assert mask.shape == (batch, size // stride, size // stride, 1)


def test_resstack_masking_invariance():
  # models/layers_test.py:64-86: a change under the mask leaves the output EXACTLY equal
  gen = torch.Generator().manual_seed(2)
  b, size, cin, cout = 2, 32, 8, 4
  p = _res_stack_params('s', cin, cout, 2, 1, gen)
  x = torch.rand((b, size, size, cin), generator=gen)
  m = (torch.arange(size, dtype=torch.float32) > size // 2).float()
  mask = m[None, :, None, None].repeat(b, 1, size, 1)
  net = O.Net(p, training=None)
  o1, _ = net.res_stack(x, mask, 's', cin, cout, 2, 1, False, False)
  x2 = x.clone()
  x2[:, 0, 0, :] = 1
  o2, _ = net.res_stack(x2, mask, 's', cin, cout, 2, 1, False, False)
  assert torch.equal(o1, o2)


def test_conv_transpose_matches_gradient_of_same_conv():
  """Keras Conv2DTranspose(k3, s2, SAME, output_padding=1) is the gradient of the SAME
  stride-2 forward conv; the oracle's crop convention must satisfy <convT(x), y> = <x, conv(y)>."""
  gen = torch.Generator().manual_seed(3)
  for k in (2, 3):
    kern = torch.randn((k, k, 5, 7), generator=gen)   # (kh,kw,cout_T,cin_T) = HWIO of the conv
    x = torch.randn((2, 6, 8, 7), generator=gen)
    y = torch.randn((2, 12, 16, 5), generator=gen)
    lhs = (O.keras_conv2d_transpose(x, kern, None, 2) * y).sum()
    rhs = (x * O.tf_conv2d(y, kern, 2, 'SAME')).sum()
    np.testing.assert_allclose(float(lhs), float(rhs), rtol=1e-4)


def test_avg_and_max_pool_same_semantics():
  x = torch.arange(2 * 5 * 7 * 1, dtype=torch.float32).reshape(2, 5, 7, 1)
  a = O.avg_pool3s2_same(x)
  assert a.shape == (2, 3, 4, 1)
  # top-left window covers rows 0..1/cols 0..1 only when padding sits at the top/left; TF puts
  # the extra padding at the bottom/right for even sizes and splits it evenly for odd sizes:
  # 5 rows -> pad (1,1): first window rows {-1,0,1} -> 2 valid rows
  np.testing.assert_allclose(float(a[0, 0, 0, 0]), float(x[0, 0:2, 0:2, 0].mean()))
  m = O.max_pool_same(x)
  assert m.shape == (2, 3, 4, 1)
  assert float(m[0, 2, 3, 0]) == float(x[0, 4, 6, 0])


def test_fp64_yardstick_fast_paths_equal_the_plain_formulation():
  """The binary64 runs of the oracle (yardstick of the GPU gradient tests) take a patch-matrix
  convolution and divide the spectral conv's OUTPUT by sigma instead of its kernel; both must be
  the plain F.conv2d with the normalised kernel up to binary64 rounding, values and gradients."""
  import torch.nn.functional as F
  g = torch.Generator().manual_seed(0)
  for (k, stride, padding, h, w) in ((3, 1, 'SAME', 6, 9), (4, 2, 'SAME', 9, 12), (1, 1, 'VALID', 5, 4),
                                     (7, 2, 'VALID', 15, 17), (3, 2, 'SAME', 8, 8), (2, 1, 'VALID', 4, 6)):
    x = torch.randn((2, h, w, 5), generator=g, dtype=torch.float64, requires_grad=True)
    kern = torch.randn((k, k, 5, 7), generator=g, dtype=torch.float64, requires_grad=True)
    y = O.tf_conv2d(x, kern, stride, padding)
    xn = x.permute(0, 3, 1, 2)
    if padding == 'SAME':
      pt, pb = O._same_pads(h, k, stride)
      pl, pr = O._same_pads(w, k, stride)
      xn = F.pad(xn, (pl, pr, pt, pb))
    want = F.conv2d(xn, kern.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)
    assert y.shape == want.shape
    cot = torch.randn(y.shape, generator=g, dtype=torch.float64)
    gx, gk = torch.autograd.grad((y * cot).sum(), (x, kern))
    wx, wk = torch.autograd.grad((want * cot).sum(), (x, kern))
    for a, b in ((y, want), (gx, wx), (gk, wk)):
      assert float((a - b).abs().max()) <= 1e-13 * float(b.abs().max()), (k, stride, padding)
  # spectral conv: conv(x, W / sigma) (fp32 path, plain) == conv(x, W) / sigma (binary64 path)
  p64 = {'c/kernel': torch.randn((3, 3, 4, 6), generator=g, dtype=torch.float64),
         'c/u': torch.randn((1, 6), generator=g, dtype=torch.float64),
         'c/bias': torch.randn((6,), generator=g, dtype=torch.float64)}
  x0 = torch.randn((2, 7, 5, 4), generator=g, dtype=torch.float64)
  cot = torch.randn((2, 7, 5, 6), generator=g, dtype=torch.float64)
  outs = []
  for plain in (False, True):
    p = {k: v.clone().requires_grad_(k != 'c/u') for k, v in p64.items()}
    x = x0.clone().requires_grad_(True)
    net = O.Net(p, training=True)
    if plain:
      sigma, _ = O.power_iteration(p['c/kernel'], p['c/u'])
      w_n = p['c/kernel'] / (sigma + 1e-10)
      xn = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1))
      y = F.conv2d(xn, w_n.permute(3, 2, 0, 1)).permute(0, 2, 3, 1) + p['c/bias']
    else:
      y = net.spectral_conv(x, 'c', 1, 'SAME')
    gx, gk, gb = torch.autograd.grad((y * cot).sum(), (x, p['c/kernel'], p['c/bias']))
    outs.append((y.detach(), gx, gk, gb))
  for a_, b_ in zip(*outs):
    assert float((a_ - b_).abs().max()) <= 1e-13 * float(b_.abs().max())
